"""Per-iteration schedules of the MoCo pretrain loop (host-side scalars; MAIN_MOCO:608-629)."""
import math


def adjust_learning_rate(optimizer, epoch, lr, epochs, warmup_epochs, cos=True, schedule=()):
    """Half-cycle cosine after linear warm-up (cos=True) or step-wise x0.1 at the milestones; writes the value into every
    param group and returns it (MAIN_MOCO:608-623).  ``epoch`` may be fractional (epoch + i / iters_per_epoch)."""
    if cos:
        if epoch < warmup_epochs:
            lr_ = lr * epoch / warmup_epochs
        else:
            lr_ = lr * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))
    else:
        lr_ = lr
        for milestone in schedule:
            lr_ *= 0.1 if epoch >= milestone else 1.0
    if optimizer is not None:
        for group in optimizer.param_groups:
            group["lr"] = lr_
    return lr_


def adjust_moco_momentum(epoch, epochs, moco_m):
    """m = 1 - 0.5 (1 + cos(pi e / E)) (1 - m0)   (MAIN_MOCO:626-629)."""
    return 1.0 - 0.5 * (1.0 + math.cos(math.pi * epoch / epochs)) * (1.0 - moco_m)
