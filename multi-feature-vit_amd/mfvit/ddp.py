"""Data-parallel gradient exchange over RCCL (torch.distributed backend 'nccl' on ROCm) for arena-backed modules.

One process per GPU.  The encoder backward runs in groups of stages (mfvit_vit_backward stage_hi..stage_lo, by default 4 ViT
blocks per group: 4 x 1.77 M floats = 28 MB f32, contiguous in the flat gradient arena); after each group its slice is
all-reduced asynchronously, so the exchange of a group overlaps the backward of the groups below it.  xGMI is
point-to-point (7 links x ~153 GB/s per GPU): per-block buckets keep every message large enough for the ring/direct
algorithms while leaving 11 blocks of compute to hide each one behind.

``bucket_dtype=torch.bfloat16`` halves the bytes on the links: every group is cast into a persistent bf16 bucket (same offsets as
the gradient arena), the bucket is all-reduced, and the result is written back over the f32 slice when the group is joined.
Joins are per owner: ``finish(vit)`` waits for one encoder's groups only, so the optimizer can start on the arenas whose exchange
is complete (``optimizer.before_group``) while the rest is still on the links.

``exchange='rs_ag'`` (``MFVIT_GRAD_EXCHANGE=rs_ag``) issues every bucket as reduce-scatter + all-gather instead of one
all-reduce: each rank reduces 1/W of the bucket and the shards are gathered back - on 8 fully connected xGMI peers both phases use
all seven links at once where a ring is bound by one.  Combine with ``MFVIT_GRAD_BUCKET_LAYERS=1`` for the per-block buckets of the
survey.  The all-reduce stays the default until a measured scaling curve says otherwise (RCCL picks its own algorithm for it).
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, group=None, bucket_dtype=None, exchange=None):
        import os
        self.group = group
        self.exchange = exchange or os.environ.get("MFVIT_GRAD_EXCHANGE", "allreduce")
        if self.exchange not in ("allreduce", "rs_ag"):
            raise ValueError(f"unknown gradient exchange {self.exchange!r} (allreduce | rs_ag)")
        self._shards = {}                  # (data_ptr, numel, dtype) -> this rank's persistent shard buffer
        self.handles = []                  # (owner key, work handle, write-back or None)
        if bucket_dtype is None and os.environ.get("MFVIT_GRAD_BUCKET_DTYPE", "f32") in ("bf16", "bfloat16"):
            bucket_dtype = torch.bfloat16
        self.bucket_dtype = bucket_dtype if bucket_dtype not in (None, torch.float32) else None
        self._buckets = {}                 # id(vit) -> persistent low-precision copy of the gradient arena
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        # RCCL reduces with AVG in one pass; gloo (CPU tests) has no AVG: SUM then scale
        self.avg = self.enabled and dist.get_backend(group) == "nccl"
        # diagnostics (bench.py --gpus N): with `timing = True` every finish() brackets its waits with events on the current stream, and
        # finish_wait_ms() tells how long the compute stream sat waiting for exchanges that had not completed - the part of the
        # all-reduce that backward did NOT hide
        self.timing = False
        self._wait_events = []

    def _all_reduce(self, t, async_op):
        """Mean over ranks, in place.  ONE call shape for every backend: RCCL reduces with AVG in a single pass; gloo (CPU tests,
        one-GPU rehearsals) has no AVG, so the tensor is pre-scaled by 1/world and summed - the asynchronous handles issued from the
        backward hooks are the same objects either way."""
        if self.avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        t.div_(self.world)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _reduce_async(self, t):
        """Asynchronous mean over ranks of the flat tensor `t`, in place: the work handles to join (in order)."""
        if self.exchange != "rs_ag" or t.numel() < self.world:
            return [self._all_reduce(t, True)]
        w = self.world
        rank = dist.get_rank(self.group)
        n0 = (t.numel() // w) * w
        main, sh = t[:n0], n0 // w
        key = (t.data_ptr(), n0, t.dtype)
        mine = self._shards.get(key)
        if mine is None:
            if len(self._shards) > 256:    # (gradient arenas are reused step after step: the keys repeat; a caller that is not is bounded here)
                self._shards.clear()
            mine = self._shards[key] = torch.empty(sh, dtype=t.dtype, device=t.device)
        hs = []
        if not self.avg:
            t.div_(w)
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        h1 = dist.reduce_scatter_tensor(mine, main, op=op, group=self.group, async_op=True)
        if not self.avg:
            h1.wait()                      # gloo (tests, rehearsals) runs asynchronous work on a thread pool: no issue order to rely on
        else:
            hs.append(h1)                  # RCCL: both collectives are queued on the communicator's stream, in order
        hs.append(dist.all_gather_into_tensor(main, mine, group=self.group, async_op=True))
        if n0 < t.numel():                 # the < W elements that do not divide over the ranks
            hs.append(dist.all_reduce(t[n0:], op=op, group=self.group, async_op=True))
        return hs

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
            self._exchange(vit, gflat, a, e + n)
        if lo <= -1:
            a0, _ = vit.block_slice(0)
            self._exchange(vit, gflat, 0, a0)
            e, n = vit.block_slice(vit.depth - 1)
            self._exchange(vit, gflat, e + n, gflat.numel())

    def _exchange(self, vit, gflat, a, b):
        """Asynchronous mean over ranks of gflat[a:b]; joined by finish(vit) / finish()."""
        if b <= a:
            return
        if self.bucket_dtype is None:
            for h in self._reduce_async(gflat[a:b]):
                self._push(h, id(vit))
            return
        buf = self._buckets.get(id(vit))
        if buf is None or buf.numel() != gflat.numel() or buf.device != gflat.device:
            buf = self._buckets[id(vit)] = torch.empty(gflat.numel(), dtype=self.bucket_dtype, device=gflat.device)
        src, dst = gflat[a:b], buf[a:b]
        if self.exchange == "rs_ag":
            dst.copy_(src)                                               # (_reduce_async averages: AVG on RCCL, 1 / W + SUM on gloo)
            hs = self._reduce_async(dst)
            for h in hs[:-1]:
                self._push(h, id(vit))
            self._push(hs[-1], id(vit), lambda: src.copy_(dst))
            return
        if self.avg:
            dst.copy_(src)
        else:
            dst.copy_(src / self.world)                                  # scale in f32, round once
        h = dist.all_reduce(dst, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._push(h, id(vit), lambda: src.copy_(dst))

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

    def _push(self, h, owner=None, write_back=None):
        if h is not None and hasattr(h, "wait"):
            self.handles.append((owner, h, write_back))

    def pending(self, owner=None):
        key = id(owner) if owner is not None else None
        return sum(1 for k, _, _ in self.handles if key is None or k == key)

    def finish(self, owner=None):
        """Join the exchanges issued for `owner` (an attached encoder); without an argument, all of them."""
        key = id(owner) if owner is not None else None
        rest = []
        ev0 = None
        if self.timing and torch.cuda.is_available() and any(key is None or k == key for k, _, _ in self.handles):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for k, h, write_back in self.handles:
            if key is not None and k != key:
                rest.append((k, h, write_back))
                continue
            h.wait()
            if write_back is not None:
                write_back()
        self.handles = rest
        if ev0 is not None:
            ev1.record()
            self._wait_events.append((ev0, ev1))

    def finish_wait_ms(self, reset=True):
        """Total time (ms) the current stream spent inside finish() since the last reset: waits for exchanges still on the links plus the
        write-backs of low-precision buckets.  Call after a synchronize."""
        t = sum(a.elapsed_time(b) for a, b in self._wait_events)
        if reset:
            self._wait_events = []
        return t
