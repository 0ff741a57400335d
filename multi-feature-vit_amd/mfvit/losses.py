"""nn.CrossEntropyLoss (mean) for a handful of classes on the HIP path (MAIN_CA:432,873; MAIN_SS:714)."""
import torch

from . import ops


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss, dlogits, preds = ops.cross_entropy(logits.contiguous().float(), target.contiguous().long(), want_grad=True)
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(preds)
        return loss.reshape(()), preds

    @staticmethod
    def backward(ctx, gloss, _gpreds):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gloss, None


def cross_entropy(logits, target):
    """Returns (loss scalar, preds) - preds = argmax(logits, 1) as torch.max would give (MAIN_CA:870)."""
    return _CEFn.apply(logits, target)
