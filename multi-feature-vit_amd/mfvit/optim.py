"""Multi-tensor optimizers on the HIP path: torch.optim.Optimizer subclasses with the reference's
constructor signatures whose ``step()`` is one or two kernel launches over a device-resident chunk table.

State layout = the reference's / torch's, so optimizer state dicts move both ways between this package and the reference drivers
(MAIN_MOCO:461-467 saves ``optimizer.state_dict()``, its resume path loads it):
  LARS (OPT:10-43)            state[p] = {'mu'}
  Adam / AdamW (torch.optim)  state[p] = {'step' (f32 scalar tensor), 'exp_avg', 'exp_avg_sq'}
  SGD (torch.optim)           state[p] = {'momentum_buffer'}
The device chunk tables are runtime caches of raw addresses: they live on the optimizer object (never inside ``param_groups``,
which ``state_dict()`` pickles), are keyed by every address a row holds (parameter, gradient, both state tensors), and are dropped by
``load_state_dict``.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

CHUNK = 1 << 14          # elements per table row = per workgroup (45 M parameters: ~2,800 workgroups of 256 threads)


def _bump_versions(params):
    """The kernels write parameters through raw pointers: tell autograd (saved-tensor checks) and the encoders' weight-shadow
    caches (keyed on the parameter versions) that the data changed."""
    torch._C._increment_version(list(params))


class _TableOptimizer(torch.optim.Optimizer):
    state_names = ("s0",)          # names of the per-parameter state tensors, in table-column order

    # ---- runtime caches (never pickled: Optimizer.__getstate__ keeps defaults / state / param_groups only)
    def _cache(self):
        c = self.__dict__.get("_mfvit_cache")
        if c is None:
            c = self.__dict__["_mfvit_cache"] = {}
        return c

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache().clear()      # the loaded state tensors live at new addresses
        for g in self.param_groups:     # checkpoints written by round-1 builds carried these runtime keys
            g.pop("_mfvit_live", None)
            g.pop("_mfvit_tables", None)

    def _ensure_state(self, p):
        st = self.state[p]
        for name in self.state_names:
            t = st.get(name)
            if t is None:
                st[name] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            elif (not t.is_cuda or t.device != p.device or t.dtype != torch.float32 or not t.is_contiguous()
                  or t.shape != p.shape):
                st[name] = t.to(device=p.device, dtype=torch.float32).reshape(p.shape).contiguous()
        return st

    def _table(self, gi, group):
        """(table tensor, ntensors, live params) for the params of `group` that have gradients; rebuilt when any address changes."""
        cache = self._cache().setdefault(gi, {})
        # Fast path (the host sets the pace of a small-batch step: tools/host_profile.py measured 1.8 ms of Python per optimizer step here,
        # of a 6.8 ms step at 16 pairs): the same parameter objects with the same addresses as the last call - parameter, gradient, every state
        # tensor - mean the same table.  One flat list of integers is compared; nothing else is touched.
        params = group["params"]
        last = cache.get("last")
        if last is not None and len(last[0]) == len(params) and last[4] == len(self.state):     # (state.clear() / a new parameter: the slow path)
            names = self.state_names
            sig = []
            try:
                for p, st in zip(params, last[1]):
                    g = p.grad
                    sig.append(p.data_ptr())
                    sig.append(0 if g is None else g.data_ptr())
                    if st is not None:
                        for n in names:
                            sig.append(st[n].data_ptr())
            except (KeyError, AttributeError):
                sig = None
            if sig is not None and sig == last[2] and all(a is b for a, b in zip(params, last[0])):
                return last[3]
        ps = [p for p in params if p.grad is not None]
        if not ps:
            return None, 0, ps
        keys = []
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32:
                raise _lib.MfvitError("the HIP optimizers need contiguous f32 parameters and gradients on the GPU")
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            st = self._ensure_state(p)
            keys.append((p.data_ptr(), p.grad.data_ptr(), p.numel()) + tuple(st[n].data_ptr() for n in self.state_names))
        key = tuple(keys)
        # Gradients are re-created every step (zero_grad(set_to_none=True)); the caching allocator hands the same few addresses out
        # in a cycle, so keep a table per address pattern instead of rebuilding (a ~700-row Python loop + an upload) whenever it
        # changes.  The upload goes through pinned memory and does not block the host: a pageable cudaMemcpy here would wait for the
        # whole backward still queued on the stream and leave the GPU idle until the host has caught up again (measured: 1.9 ms of
        # idle GPU per step).
        tables = cache.setdefault("tables", {})
        hit = tables.get(key)
        if hit is None:
            rows = []
            for tid, p in enumerate(ps):
                st = self.state[p]
                n = p.numel()
                s0 = st[self.state_names[0]].data_ptr()
                s1 = st[self.state_names[1]].data_ptr() if len(self.state_names) > 1 else 0
                for a in range(0, n, CHUNK):
                    c = min(CHUNK, n - a)
                    rows.append([tid, p.data_ptr() + 4 * a, p.grad.data_ptr() + 4 * a, s0 + 4 * a, s1 + 4 * a if s1 else 0, c,
                                 self._flag(p, group)])
            host = torch.tensor(rows, dtype=torch.int64).pin_memory()
            if len(tables) >= 8:
                tables.clear()
            hit = tables[key] = (host.to(ps[0].device, non_blocking=True), len(ps), host)   # keep the pinned source alive
        out = (hit[0], hit[1], ps)
        # signature of this call for the fast path: the state dicts of the live parameters (None for a parameter without a gradient)
        sts = [self.state[p] if p.grad is not None else None for p in params]
        sig = []
        for p, st in zip(params, sts):
            sig.append(p.data_ptr())
            sig.append(0 if p.grad is None else p.grad.data_ptr())
            if st is not None:
                for n in self.state_names:
                    sig.append(st[n].data_ptr())
        cache["last"] = (list(params), sts, sig, out, len(self.state))
        return out

    def _flag(self, p, group):
        return 0

    # ``optimizer.before_group = fn(group_index)`` runs before the kernels of each param group are queued: the data-parallel driver
    # joins only that group's gradient exchange there (GradSync.finish(owner)), so the update of the arenas whose all-reduce is
    # complete overlaps the exchanges still on the links.
    before_group = None

    def _pre(self, gi):
        cb = self.__dict__.get("before_group")
        if cb is not None:
            cb(gi)

    @torch.no_grad()
    def unscale_(self, inv_scale, found_inf):
        """GradScaler.unscale_: g *= inv_scale for every gradient of this optimizer; found_inf (device f32[1]) is set to 1 when a
        gradient held an inf / nan (mfvit_amp_unscale)."""
        for gi, g in enumerate(self.param_groups):
            self._pre(gi)
            table, nt, _ = self._table(gi, g)
            if table is None:
                continue
            check(lib().mfvit_amp_unscale(ptr(table), table.shape[0], float(inv_scale), ptr(found_inf), stream()), "mfvit_amp_unscale")


class LARS(_TableOptimizer):
    """LARS optimizer, no rate scaling or weight decay for parameters <= 1D (OPT:10-43)."""
    state_names = ("mu",)

    def __init__(self, params, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001):
        defaults = dict(lr=lr, weight_decay=weight_decay, momentum=momentum, trust_coefficient=trust_coefficient)
        super().__init__(params, defaults)

    def _flag(self, p, group):
        return 1 if p.ndim > 1 else 0

    @torch.no_grad()
    def step(self):
        for gi, g in enumerate(self.param_groups):
            self._pre(gi)
            table, nt, live = self._table(gi, g)
            if table is None:
                continue
            norms = torch.empty(2 * nt, device=table.device, dtype=torch.float32)
            check(lib().mfvit_lars_step(ptr(table), table.shape[0], nt, ptr(norms), float(g["lr"]), float(g["weight_decay"]),
                                        float(g["momentum"]), float(g["trust_coefficient"]), stream()), "mfvit_lars_step")
            _bump_versions(live)


class Adam(_TableOptimizer):
    """torch.optim.Adam semantics (L2 weight decay folded into the gradient)."""
    state_names = ("exp_avg", "exp_avg_sq")
    decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _flag(self, p, group):
        return 1 if self.decoupled else 0

    def _bump_steps(self, gi, live):
        """state[p]['step'] += 1 for the live parameters; returns the values BEFORE the increment.  The per-parameter f32 scalars of torch.optim
        (host tensors) are 0-dim views of ONE host vector per param group, so the increment is one add on that vector and the values one
        tolist() - 324 scalar tensors bumped one by one (`_foreach_add_` has no fast path on the CPU) were 0.5 ms of host time per step.  A 'step'
        somebody replaced (a loaded state dict, a torch optimizer's state) is taken at its value and moved into the shared vector."""
        cache = self._cache().setdefault(("steps", gi), {})
        views = cache.get("views")
        if cache.get("live") is live:          # (the table's fast path hands the same list object out while nothing has changed)
            sts = cache["sts"]
        else:
            sts = [self.state[p] for p in live]
            cache["live"], cache["sts"] = live, sts
        if views is None or len(views) != len(sts) or any(st.get("step") is not v for st, v in zip(sts, views)):
            vals = []
            for st in sts:
                t = st.get("step")
                vals.append(float(t) if t is not None else 0.0)
            buf = torch.tensor(vals, dtype=torch.float32)
            views = [buf[i] for i in range(len(vals))]
            for st, v in zip(sts, views):
                st["step"] = v
            cache["buf"], cache["views"] = buf, views
        buf = cache["buf"]
        vals = [int(v) for v in buf.tolist()]
        buf += 1.0
        return vals

    @torch.no_grad()
    def step(self):
        for gi, g in enumerate(self.param_groups):
            self._pre(gi)
            table, nt, live = self._table(gi, g)
            if table is None:
                continue
            # per-parameter 'step' like torch.optim (f32 scalar tensors on the host).  One kernel launch serves one step number (bias
            # correction): normally every parameter of the group agrees and the whole table goes out in one launch; otherwise (a parameter
            # whose first gradient arrived later, an unfrozen layer, a loaded torch state dict with mixed steps) the table rows - laid out
            # parameter by parameter - are launched in runs of equal step.
            vals = self._bump_steps(gi, live)
            if vals and vals.count(vals[0]) == len(vals):          # the normal case: one step number, the whole table in one launch
                runs = [[0, table.shape[0], vals[0] + 1]]
            else:
                runs = []                                      # (first table row, rows, step after the increment)
                row = 0
                for p, v in zip(live, vals):
                    nrows = (p.numel() + CHUNK - 1) // CHUNK
                    if runs and runs[-1][2] == v + 1:
                        runs[-1][1] += nrows
                    else:
                        runs.append([row, nrows, v + 1])
                    row += nrows
                assert row == table.shape[0]
            for first, nrows, step in runs:
                check(lib().mfvit_adam_step(ptr(table) + first * table.shape[1] * 8, nrows, float(g["lr"]), float(g["betas"][0]),
                                            float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), step, stream()), "mfvit_adam_step")
            _bump_versions(live)


class AdamW(Adam):
    """torch.optim.AdamW semantics (decoupled weight decay; default 1e-2 like torch)."""
    decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)


class SGD(_TableOptimizer):
    """torch.optim.SGD semantics (momentum, L2 weight decay; no dampening / nesterov).  torch initialises 'momentum_buffer' with the
    first update d; a zero buffer gives the same first step (momentum * 0 + d), so the kernel never needs the first-step case."""
    state_names = ("momentum_buffer",)

    def __init__(self, params, lr=1e-3, momentum=0, weight_decay=0, dampening=0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError("the HIP SGD kernel implements the reference's configuration (MAIN_CA:445-448): no dampening, no nesterov")
        # dampening / nesterov are kept in the group so that torch.optim.SGD can load this optimizer's state dict (and vice versa)
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False))

    @torch.no_grad()
    def step(self):
        for gi, g in enumerate(self.param_groups):
            self._pre(gi)
            table, nt, live = self._table(gi, g)
            if table is None:
                continue
            check(lib().mfvit_sgd_step(ptr(table), table.shape[0], float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), 0,
                                       stream()), "mfvit_sgd_step")
            _bump_versions(live)
