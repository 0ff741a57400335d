"""Flat f32 parameter arenas: every parameter of a module group becomes a view of one contiguous buffer so that the
HIP kernels, fused optimizers, EMA and gradient all-reduce address them as slices (288 GB HBM: few, large buffers)."""
import torch


class ParamArena:
    def __init__(self, named_params):
        """named_params: list of (name, nn.Parameter) in the layout order the C ABI expects."""
        self.named = list(named_params)
        self.flat = None
        self.offsets = {}
        self.rebuild()

    def rebuild(self):
        total = sum(p.numel() for _, p in self.named)
        dev = self.named[0][1].device
        flat = torch.empty(total, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for name, p in self.named:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1).float())
                p.data = flat[off:off + n].view(p.shape)
                self.offsets[name] = (off, n)
                off += n
        self.flat = flat
        self.params = [p for _, p in self.named]

    def intact(self):
        base = self.flat.data_ptr()
        for name, p in self.named:
            off, _ = self.offsets[name]
            if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                return False
        return True

    def ensure(self):
        if not self.intact():
            self.rebuild()
        return self.flat

    def version(self):
        return sum(p._version for p in self.params)

    def grad_views(self, gflat):
        out = []
        for name, p in self.named:
            if p.requires_grad:
                off, n = self.offsets[name]
                out.append(gflat[off:off + n].view(p.shape))
            else:
                out.append(None)
        return out
