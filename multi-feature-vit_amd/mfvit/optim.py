"""Multi-tensor optimizers on the HIP path: torch.optim.Optimizer subclasses with the reference's
constructor signatures whose ``step()`` is one or two kernel launches over a device-resident chunk table."""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

CHUNK = 1 << 16


def _bump_versions(params):
    """The kernels write parameters through raw pointers: tell autograd (saved-tensor checks) and the encoders' weight-shadow
    caches (keyed on the parameter versions) that the data changed."""
    torch._C._increment_version(list(params))


class _TableOptimizer(torch.optim.Optimizer):
    nstate = 1

    def _table(self, group):
        """(table tensor, ntensors) for the params of `group` that have gradients; rebuilt when any pointer changes."""
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None, 0
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32:
                raise _lib.MfvitError("the HIP optimizers need contiguous f32 parameters and gradients on the GPU")
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            st = self.state[p]
            if "s0" not in st:
                st["s0"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if self.nstate > 1:
                    st["s1"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        group["_mfvit_live"] = ps
        key = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in ps)
        # Gradients are re-created every step (zero_grad(set_to_none=True)); the caching allocator hands the same few addresses out
        # in a cycle, so keep a table per address pattern instead of rebuilding (a ~700-row Python loop + an upload) whenever it
        # changes.  The upload goes through pinned memory and does not block the host: a pageable cudaMemcpy here would wait for the
        # whole backward still queued on the stream and leave the GPU idle until the host has caught up again (measured: 1.9 ms of
        # idle GPU per step).
        cache = group.setdefault("_mfvit_tables", {})
        hit = cache.get(key)
        if hit is None:
            rows = []
            for tid, p in enumerate(ps):
                st = self.state[p]
                n = p.numel()
                for a in range(0, n, CHUNK):
                    c = min(CHUNK, n - a)
                    rows.append([tid, p.data_ptr() + 4 * a, p.grad.data_ptr() + 4 * a, st["s0"].data_ptr() + 4 * a,
                                 st["s1"].data_ptr() + 4 * a if self.nstate > 1 else 0, c, self._flag(p, group)])
            host = torch.tensor(rows, dtype=torch.int64).pin_memory()
            if len(cache) >= 8:
                cache.clear()
            hit = cache[key] = (host.to(ps[0].device, non_blocking=True), len(ps), host)   # keep the pinned source alive
        return hit[0], hit[1]

    def _flag(self, p, group):
        return 0


class LARS(_TableOptimizer):
    """LARS optimizer, no rate scaling or weight decay for parameters <= 1D (OPT:10-43)."""

    def __init__(self, params, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001):
        defaults = dict(lr=lr, weight_decay=weight_decay, momentum=momentum, trust_coefficient=trust_coefficient)
        super().__init__(params, defaults)

    def _flag(self, p, group):
        return 1 if p.ndim > 1 else 0

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            table, nt = self._table(g)
            if table is None:
                continue
            norms = torch.empty(2 * nt, device=table.device, dtype=torch.float32)
            check(lib().mfvit_lars_step(ptr(table), table.shape[0], nt, ptr(norms), float(g["lr"]), float(g["weight_decay"]),
                                        float(g["momentum"]), float(g["trust_coefficient"]), stream()), "mfvit_lars_step")
            _bump_versions(g["_mfvit_live"])


class Adam(_TableOptimizer):
    """torch.optim.Adam semantics (L2 weight decay folded into the gradient)."""
    nstate = 2
    decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _flag(self, p, group):
        return 1 if self.decoupled else 0

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            table, nt = self._table(g)
            if table is None:
                continue
            g["_step"] = g.get("_step", 0) + 1
            check(lib().mfvit_adam_step(ptr(table), table.shape[0], float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                        float(g["eps"]), float(g["weight_decay"]), g["_step"], stream()), "mfvit_adam_step")
            _bump_versions(g["_mfvit_live"])


class AdamW(Adam):
    """torch.optim.AdamW semantics (decoupled weight decay; default 1e-2 like torch)."""
    decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)


class SGD(_TableOptimizer):
    """torch.optim.SGD semantics (momentum, L2 weight decay; no dampening / nesterov)."""

    def __init__(self, params, lr=1e-3, momentum=0, weight_decay=0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            table, nt = self._table(g)
            if table is None:
                continue
            first = 0 if g.get("_started") else 1
            g["_started"] = True
            check(lib().mfvit_sgd_step(ptr(table), table.shape[0], float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), first,
                                       stream()), "mfvit_sgd_step")
            _bump_versions(g["_mfvit_live"])
