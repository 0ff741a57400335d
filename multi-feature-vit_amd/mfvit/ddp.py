"""Data-parallel gradient exchange over RCCL (torch.distributed backend 'nccl' on ROCm) for arena-backed modules.

One process per GPU.  The encoder backward runs stage by stage (mfvit_vit_backward stage_hi..stage_lo); after each
stage the gradient slice of that stage (one ViT block = 1.77 M floats = 7.1 MB f32, contiguous in the flat gradient
arena) is all-reduced asynchronously, so the exchange of block l overlaps the backward of blocks l-1..0.  xGMI is
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
        if self.avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        h = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=False)
        t.div_(self.world)
        return h

    # ---- encoder: called from VisionTransformerMoCo._run_backward after every stage
    def attach(self, vit):
        if self.enabled:
            vit._grad_stage_hook = self._stage_hook
        return vit

    def _stage_hook(self, vit, stage, gflat):
        if stage == vit.depth:          # final norm (2 x 384 floats): sent together with the embed stage
            return
        if stage >= 0:
            a, n = vit.block_slice(stage)
            self._push(self._all_reduce(gflat[a:a + n], True))
        else:
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
