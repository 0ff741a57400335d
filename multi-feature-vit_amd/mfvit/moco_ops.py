"""Row-wise MoCo ops on the HIP path: F.normalize, positive logits, negative logits against the queue, cross entropy
over the (1 + K)-wide logits, EMA over flat arenas."""
import torch

from . import _lib, ops
from ._lib import check, lib, ptr, stream


class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x)
        x = x.contiguous().float()
        n, C = x.shape
        y = torch.empty_like(x)
        inv = torch.empty(n, device=x.device, dtype=torch.float32)
        check(lib().mfvit_l2norm_fwd(ptr(x), ptr(y), ptr(inv), n, C, 1e-12, stream()), "mfvit_l2norm_fwd")
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = dy.contiguous().float()
        dx = torch.empty_like(y)
        check(lib().mfvit_l2norm_bwd(ptr(dy), ptr(y), ptr(inv), ptr(dx), y.shape[0], y.shape[1], stream()), "mfvit_l2norm_bwd")
        return dx


def l2_normalize(x):
    """F.normalize(x, dim=1)  (BLD:165,175)."""
    return _L2NormFn.apply(x)


class _PosLogitFn(torch.autograd.Function):
    """l_pos[i] = q[i] . k[i]   (BLD:183); k carries no gradient."""

    @staticmethod
    def forward(ctx, q, k):
        n, C = q.shape
        out = torch.empty(n, 1, device=q.device, dtype=torch.float32)
        check(lib().mfvit_rowdot(ptr(q), ptr(k), ptr(out), 1, 1.0, n, C, stream()), "mfvit_rowdot")
        ctx.save_for_backward(k)
        return out

    @staticmethod
    def backward(ctx, dl):
        (k,) = ctx.saved_tensors
        return dl.reshape(-1, 1) * k, None


class _NegLogitFn(torch.autograd.Function):
    """l_neg = q @ queue   (BLD:185; queue (C, K) held key-major as queue_t (K, C)); exact-f32 MFMA, no queue gradient."""

    @staticmethod
    def forward(ctx, q, queue_t):
        q = q.contiguous()
        out = ops.linear_fwd(q, queue_t)                        # (n, K)
        # the enqueue that follows overwrites rows of the queue in place: the backward needs the OLD keys, so keep a copy
        # (what BLD:185 `self.queue.clone().detach()` does; 64 MiB, ~20 us of HBM traffic per step)
        ctx.save_for_backward(queue_t.clone())
        return out

    @staticmethod
    def backward(ctx, dl):
        (queue_t,) = ctx.saved_tensors
        K, C = queue_t.shape
        n = dl.shape[0]
        # dq[n][c] = sum_j dl[n][j] queue_t[j][c]: reduction over the 65536 keys -> split-M wgrad kernel on the transposed dl
        _, dlt = ops.cast_transpose(dl.contiguous().float(), torch.float32, want_straight=False)     # (K, n)
        pad = (-n) % 128
        if pad:
            dlt = torch.nn.functional.pad(dlt, (0, pad))
        out = ops.linear_wgrad(queue_t, dlt, scratch=ops.wgrad_scratch(dlt.device))     # (C, n + pad) = queue_t^T @ dl^T; split partials + fixed-order reduce
        return out[:, :n].t().contiguous(), None


def pos_logits(q, k):
    return _PosLogitFn.apply(q.contiguous().float(), k.contiguous().float())


def neg_logits(q, queue_t):
    return _NegLogitFn.apply(q.float(), queue_t)


class _CERowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous().float()
        n, C = logits.shape
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        dlogits = torch.empty_like(logits)
        lse = torch.empty(n, device=logits.device, dtype=torch.float32)     # (with the row log-sum-exps kept the library adds the row losses in a fixed order)
        check(lib().mfvit_cross_entropy_rows(ptr(logits), C, ptr(target.contiguous().long()), ptr(loss), ptr(lse), ptr(dlogits), C, n, C,
                                             stream()), "mfvit_cross_entropy_rows")
        ctx.save_for_backward(dlogits)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


def cross_entropy_rows(logits, target):
    """nn.CrossEntropyLoss()(logits, target) for wide rows (MAIN_MOCO:330,535)."""
    return _CERowsFn.apply(logits, target)


@torch.no_grad()
def ema_update_(dst_flat, src_flat, m, dst_params=()):
    """dst = dst * m + src * (1 - m) over flat f32 arenas (BLD:83-89).  ``dst_params``: the parameters that are views of
    ``dst_flat`` - their version counters are bumped so that weight-shadow caches see the update."""
    _lib.require_cuda(dst_flat, src_flat)
    assert dst_flat.numel() == src_flat.numel() and dst_flat.dtype == src_flat.dtype == torch.float32
    check(lib().mfvit_ema_update(ptr(dst_flat), ptr(src_flat), float(m), dst_flat.numel(), stream()), "mfvit_ema_update")
    torch._C._increment_version([dst_flat] + list(dst_params))
    return dst_flat
