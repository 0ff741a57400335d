"""Projector / predictor MLP building blocks of MoCo on the HIP path: bias-free Linear (MFMA GEMMs) and BatchNorm1d
with SyncBatchNorm semantics (BLD:62-78; MAIN_MOCO:297), each one autograd node over the C ABI."""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, ops
from ._lib import BF16, BF16X3, F16, F32, check, lib, ptr, stream


def _tdtype(precision):
    """Compute dtype of the projector / predictor MLPs.  'bf16x3' (f32-grade split products in the encoder) runs these small GEMMs
    (n rows x 4096) on the exact-f32 MFMA kernels: the same accuracy class, and their cost is negligible beside the encoder."""
    code = _lib.dtype_code(precision)
    return {BF16: torch.bfloat16, F16: torch.float16, BF16X3: torch.float32, F32: torch.float32}[code]


class _LinearFn(torch.autograd.Function):
    """y = x @ W.T (no bias).  x: (n, K) any float dtype; W: (N, K) f32 master; y in the compute dtype."""

    @staticmethod
    def forward(ctx, x, weight, cdtype):
        _lib.require_cuda(x, weight)
        xT = x.to(cdtype).contiguous()
        w, wt = ops.cast_transpose(weight.detach(), cdtype, want_straight=(cdtype != torch.float32), want_transposed=True)
        if w is None:
            w = weight.detach()
        y = ops.linear_fwd(xT, w)
        ctx.save_for_backward(xT, wt)
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        xT, wt = ctx.saved_tensors
        dy = dy.to(xT.dtype).contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_fwd(dy, wt).to(ctx.in_dtype)     # dx = dy @ W  (wt = W^T, [K][N])
        if ctx.needs_input_grad[1]:
            dw = ops.linear_wgrad(dy, xT, scratch=ops.wgrad_scratch(dy.device))      # (N, K) f32; split partials + fixed-order reduce (no float atomics)
        return dx, dw, None


class HipLinear(nn.Module):
    """nn.Linear(in, out, bias=False) with the same parameter name (``weight``) and default init."""

    def __init__(self, in_features, out_features, precision="bf16"):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.precision = precision
        ref = nn.Linear(in_features, out_features, bias=False)      # same default (kaiming-uniform) initialisation
        self.weight = nn.Parameter(ref.weight.detach().clone())

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, _tdtype(self.precision))

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias=False, precision={self.precision}"


def _sync_group():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_world_size()
    return 1


class _BNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, out_f32, training):
        _lib.require_cuda(x)
        x = x.contiguous()
        n, C = x.shape
        code = ops._code_of(x)
        dev = x.device
        mean = torch.empty(C, device=dev, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        W = _sync_group()
        if training:
            st = torch.empty(2 * C + 1, device=dev, dtype=torch.float32)
            check(lib().mfvit_bn_stats(code, ptr(x), n, C, ptr(st), st.data_ptr() + 4 * C, stream()), "mfvit_bn_stats")
            st[2 * C] = float(n)
            if W > 1:
                flat = torch.empty(W * (2 * C + 1), device=dev, dtype=torch.float32)    # concatenated form: every backend takes it
                dist.all_gather_into_tensor(flat, st)
                allst = flat.view(W, 2 * C + 1)
                means = allst[:, :C].contiguous()
                m2s = allst[:, C:2 * C].contiguous()
                counts = allst[:, 2 * C].contiguous()
            else:
                means, m2s, counts = st[:C], st[C:2 * C], st[2 * C:]
            check(lib().mfvit_bn_combine(ptr(means), ptr(m2s), ptr(counts), W, C, eps, momentum, ptr(mean), ptr(invstd),
                                         ptr(running_mean), ptr(running_var), stream()), "mfvit_bn_combine")
            total = float(n * W)
        else:
            mean.copy_(running_mean)
            invstd.copy_(torch.rsqrt(running_var + eps))
            total = float(n)
        y = torch.empty(n, C, device=dev, dtype=x.dtype)
        check(lib().mfvit_bn_apply(code, ptr(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), int(relu), ptr(y), n, C, stream()),
              "mfvit_bn_apply")
        ctx.save_for_backward(x, y if relu else None, mean, invstd, gamma)
        ctx.relu, ctx.total, ctx.W, ctx.code, ctx.training = relu, total, W, code, training
        ctx.out_f32 = out_f32
        return y.float() if out_f32 else y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, gamma = ctx.saved_tensors
        n, C = x.shape
        dy = dy.to(x.dtype).contiguous()
        s = torch.empty(2, C, device=x.device, dtype=torch.float32)
        check(lib().mfvit_bn_bwd_sums(ctx.code, ptr(dy), ptr(x), ptr(y), ptr(mean), ptr(invstd), int(ctx.relu), n, C, ptr(s),
                                      s.data_ptr() + 4 * C, stream()), "mfvit_bn_bwd_sums")
        dgamma = s[1].clone() if gamma is not None and ctx.needs_input_grad[1] else None
        dbeta = s[0].clone() if gamma is not None and ctx.needs_input_grad[2] else None
        if ctx.training and ctx.W > 1:
            dist.all_reduce(s)                       # global sums of dy' and dy' * xhat (SyncBatchNorm backward)
        if not ctx.training:                         # eval mode: statistics are constants
            s.zero_()
        dx = torch.empty_like(x)
        check(lib().mfvit_bn_bwd_apply(ctx.code, ptr(dy), ptr(x), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), int(ctx.relu), ptr(s),
                                       s.data_ptr() + 4 * C, 1.0 / ctx.total, ptr(dx), n, C, stream()), "mfvit_bn_bwd_apply")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class HipBatchNorm1d(nn.Module):
    """nn.BatchNorm1d with the same parameter / buffer names.  In training mode the batch statistics are those of the
    GLOBAL batch whenever a process group with more than one rank is initialised (what SyncBatchNorm gives the reference,
    MAIN_MOCO:297).  Not a subclass of ``_BatchNorm`` on purpose: ``convert_sync_batchnorm`` leaves it in place.
    ``relu=True`` fuses the following nn.ReLU (its slot in the Sequential is kept so state-dict indices match BLD:62-78)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, relu=False, out_f32=False):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        self.relu, self.out_f32 = relu, out_f32
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        if self.training:
            self.num_batches_tracked.add_(1)
        return _BNFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum, self.relu,
                           self.out_f32, self.training)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, affine={self.affine}, fused_relu={self.relu}"


class FusedReLUSlot(nn.Module):
    """Occupies the nn.ReLU position of the reference's Sequential (indices 2 and 5); the ReLU itself runs fused inside
    the preceding HipBatchNorm1d."""

    def forward(self, x):
        return x
