"""Epoch metrics of the finetune / CA drivers on the device.

The reference copies predictions, logits and labels of every batch to the host (MAIN_CA:886-899) and calls scikit-learn at the end
of the epoch (label_binarize + roc_curve + auc per class, MAIN_CA:901-909).  `EpochMeter` keeps them in HBM; `compute()` runs two
small HIP kernels (argmax + confusion matrix, one-vs-rest AUC pair counts as exact integers) and does ONE device-to-host copy.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream


def eval_counts(scores, labels, num_classes=None):
    """scores f32 [n, C], labels int64 [n] on the GPU -> (confusion [C, C] int64, preds [n] int64, u2 [C] int64, npos [C] int64)."""
    require_cuda(scores, labels)
    if scores.dtype != torch.float32 or labels.dtype != torch.int64 or scores.dim() != 2:
        raise _lib.MfvitError("eval_counts wants float32 [n, C] scores and int64 labels")
    n, C = scores.shape
    if num_classes is not None and num_classes != C:
        raise _lib.MfvitError("scores have a different class count")
    if scores.stride(1) != 1:
        scores = scores.contiguous()
    conf = torch.zeros(C, C, device=scores.device, dtype=torch.int64)
    u2 = torch.zeros(C, device=scores.device, dtype=torch.int64)
    npos = torch.zeros(C, device=scores.device, dtype=torch.int64)
    preds = torch.empty(n, device=scores.device, dtype=torch.int64)
    check(lib().mfvit_eval_counts(ptr(scores), scores.stride(0), ptr(labels), n, C, ptr(conf), ptr(preds), ptr(u2), ptr(npos), stream()),
          "mfvit_eval_counts")
    return conf, preds, u2, npos


class EpochMeter:
    """Accumulates `output`, `target` and the summed loss of an epoch on the device; mirrors MAIN_CA:884-911.

        meter = EpochMeter(num_classes=3)
        for ...:  meter.update(output, target, loss)        # loss = the batch MEAN, as criterion() returns it (MAIN_CA:873,884)
        epoch_loss, epoch_auc, epoch_acc = meter.compute()  # no host sync before this call
    """

    def __init__(self, num_classes=3):
        self.num_classes = num_classes
        self.reset()

    def reset(self):
        self._scores, self._labels = [], []
        self._loss_sum = None
        self.confusion = None
        self.auc_per_class = None

    def update(self, output, target, loss):
        out = output.detach().float()
        self._scores.append(out)
        self._labels.append(target.detach().long().view(-1))
        w = loss.detach().float() * out.shape[0]                      # running_loss += loss.item() * images.size(0), MAIN_CA:884
        self._loss_sum = w if self._loss_sum is None else self._loss_sum + w

    def compute(self):
        scores = torch.cat(self._scores, 0).contiguous()
        labels = torch.cat(self._labels, 0).contiguous()
        n = scores.shape[0]
        conf, _, u2, npos = eval_counts(scores, labels, self.num_classes)
        host = torch.cat([conf.view(-1), u2, npos, self._loss_sum.view(1).double().view(torch.int64)]).cpu()   # one copy
        C = self.num_classes
        conf = host[:C * C].view(C, C)
        u2, npos = host[C * C:C * C + C].double(), host[C * C + C:C * C + 2 * C].double()
        loss_sum = host[-1:].view(torch.float64).item()
        nneg = n - npos
        auc = u2 / (2.0 * npos * nneg)                                # nan for a class without positives or negatives, as sklearn
        self.confusion = conf
        self.auc_per_class = auc
        return loss_sum / n, float(auc.mean()), float(conf.diag().sum()) / n
