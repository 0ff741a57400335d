"""Loss scaling for the fp16 path: ``GradScaler`` with torch.cuda.amp.GradScaler's interface and
semantics, as the reference's pretraining loop uses it (MAIN_MOCO:349 ``scaler = torch.cuda.amp.GradScaler()``; :546-548
``scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()``; :368 / :465 state_dict in the checkpoint).

The unscale + inf check is ONE multi-tensor kernel over the optimizer's device chunk table (mfvit_amp_unscale) instead of torch's
per-device / per-dtype foreach passes; like torch, ``step`` reads the found-inf flag on the host (one 4-byte copy) and skips the
optimizer step when a gradient overflowed, and ``update`` halves the scale after an overflow / doubles it after
``growth_interval`` clean steps.
"""
import torch

from . import _lib


class GradScaler:
    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._enabled = bool(enabled)
        self._scale = float(init_scale)
        self._growth_factor = float(growth_factor)
        self._backoff_factor = float(backoff_factor)
        self._growth_interval = int(growth_interval)
        self._growth_tracker = 0
        self._found_inf = None        # device f32[1], per step
        self._unscaled = set()        # ids of optimizers already unscaled this step
        self._last_found = 0.0

    # ---- torch.cuda.amp.GradScaler interface
    def is_enabled(self):
        return self._enabled

    def get_scale(self):
        return self._scale if self._enabled else 1.0

    def scale(self, outputs):
        """loss * scale (MAIN_MOCO:546).  The scale rides on the host as a Python float: no device tensor to keep in sync."""
        if not self._enabled:
            return outputs
        return outputs * self._scale

    def _flag(self, device):
        if self._found_inf is None or self._found_inf.device != device:
            self._found_inf = torch.zeros(1, device=device, dtype=torch.float32)
        return self._found_inf

    def unscale_(self, optimizer):
        if not self._enabled:
            return
        if id(optimizer) in self._unscaled:
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().")
        if not hasattr(optimizer, "unscale_"):
            raise _lib.MfvitError("mfvit.amp.GradScaler drives the HIP multi-tensor optimizers of mfvit.optim (they own the chunk table)")
        dev = next(p for g in optimizer.param_groups for p in g["params"]).device
        optimizer.unscale_(1.0 / self._scale, self._flag(dev))
        self._unscaled.add(id(optimizer))

    def step(self, optimizer, *args, **kwargs):
        """MAIN_MOCO:547.  Unscales if the caller has not, then steps unless a gradient overflowed (returns None in that case, like
        torch)."""
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if id(optimizer) not in self._unscaled:
            self.unscale_(optimizer)
        self._last_found = float(self._found_inf.item()) if self._found_inf is not None else 0.0
        if self._last_found == 0.0:
            return optimizer.step(*args, **kwargs)
        return None

    def update(self, new_scale=None):
        """MAIN_MOCO:548."""
        if not self._enabled:
            return
        if new_scale is not None:
            self._scale = float(new_scale)
        elif self._last_found != 0.0:
            self._scale *= self._backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self._growth_interval:
                self._scale *= self._growth_factor
                self._growth_tracker = 0
        if self._found_inf is not None:
            self._found_inf.zero_()
        self._unscaled.clear()
        self._last_found = 0.0

    def state_dict(self):
        """Same keys as torch's (the reference stores it under 'scaler', MAIN_MOCO:465)."""
        if not self._enabled:
            return {}
        return {"scale": self._scale, "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, state_dict):
        if not self._enabled:
            return
        if len(state_dict) == 0:
            raise RuntimeError("The source state dict is empty, possibly because it was saved from a disabled instance of GradScaler.")
        self._scale = float(state_dict["scale"])
        self._growth_factor = float(state_dict["growth_factor"])
        self._backoff_factor = float(state_dict["backoff_factor"])
        self._growth_interval = int(state_dict["growth_interval"])
        self._growth_tracker = int(state_dict["_growth_tracker"])
