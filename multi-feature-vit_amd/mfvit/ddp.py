"""Data-parallel gradient exchange over RCCL (torch.distributed backend 'nccl' on ROCm) for arena-backed modules.

One process per GPU.  The encoder backward runs in groups of stages (mfvit_vit_backward stage_hi..stage_lo, by default 4 ViT
blocks per group: 4 x 1.77 M floats = 28 MB f32, contiguous in the flat gradient arena); after each group its slice is
all-reduced asynchronously, so the exchange of a group overlaps the backward of the groups below it.  xGMI is
point-to-point (7 links x ~153 GB/s per GPU): per-block buckets keep every message large enough for the ring/direct
algorithms while leaving 11 blocks of compute to hide each one behind.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, group=None):
        self.group = group
        self.handles = []
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        # RCCL reduces with AVG in one pass; gloo (CPU tests) has no AVG: SUM then scale
        self.avg = self.enabled and dist.get_backend(group) == "nccl"

    def _all_reduce(self, t, async_op):
        """Mean over ranks, in place.  ONE call shape for every backend: RCCL reduces with AVG in a single pass; gloo (CPU tests,
        one-GPU rehearsals) has no AVG, so the tensor is pre-scaled by 1/world and summed - the asynchronous handles issued from the
        backward hooks are the same objects either way."""
        if self.avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        t.div_(self.world)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    # ---- encoder: called from VisionTransformerMoCo._run_backward after every stage
    def attach(self, vit, bucket_layers=None):
        import os
        if bucket_layers is None:
            bucket_layers = int(os.environ.get("MFVIT_GRAD_BUCKET_LAYERS", "4"))
        vit._grad_bucket_layers = bucket_layers
        if self.enabled:
            vit._grad_stage_hook = self._stage_hook
        elif os.environ.get("MFVIT_FORCE_BUCKETS"):            # single-GPU measurement of what the grouped backward alone costs
            vit._grad_stage_hook = lambda *_: None
        return vit

    def _stage_hook(self, vit, hi, lo, gflat):
        """Called after backward stages hi .. lo (depth = final norm, depth-1 .. 0 = blocks, -1 = embedding) have been queued: the
        blocks of the group are one contiguous slice of the flat gradient arena; everything outside the blocks (cls / pos / patch
        embedding in front, final norm + head behind) goes with the group that contains the embedding stage."""
        b_hi, b_lo = min(hi, vit.depth - 1), max(lo, 0)
        if b_hi >= b_lo:
            a, _ = vit.block_slice(b_lo)
            e, n = vit.block_slice(b_hi)
            self._push(self._all_reduce(gflat[a:e + n], True))
        if lo <= -1:
            a0, _ = vit.block_slice(0)
            self._push(self._all_reduce(gflat[:a0], True))
            e, n = vit.block_slice(vit.depth - 1)
            self._push(self._all_reduce(gflat[e + n:], True))

    # ---- everything else (fusion arena, heads): one flat exchange after backward
    def reduce_grads(self, params):
        if not self.enabled:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        self._all_reduce(flat, False)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def _push(self, h):
        if h is not None and hasattr(h, "wait"):
            self.handles.append(h)

    def finish(self):
        for h in self.handles:
            h.wait()
        self.handles = []
