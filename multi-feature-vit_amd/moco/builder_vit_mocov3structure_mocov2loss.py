"""Drop-in for the reference's MoCo builder (moco_pretraining/moco/moco/builder_vit_mocov3structure_mocov2loss.py):
``MoCo`` / ``MoCo_ViT`` with the same constructor ``(base_encoder, args, dim=256, mlp_dim=4096, T=1.0)``, the same
``forward(im_q, im_k, m) -> (logits (n, 1+K), labels (n,))``, the same buffers (``queue``, ``queue_ptr``) and state-dict keys,
running on the gfx950 kernels (ViT encoders, MFMA Linear layers, SyncBN-semantics BatchNorm, EMA, InfoNCE logits).

MI355X-first differences that do not change results:
  * momentum update = one launch per flat arena instead of a Python loop over ~157 tensors x 3 kernels (BLD:83-89);
  * the queue is stored key-major ((K, C) rows contiguous; ``queue`` keeps the reference's (C, K) shape through strides), so
    enqueue writes whole rows and l_neg streams contiguous keys (the per-step copy of BLD:185 is kept: the backward needs
    the keys as they were before the enqueue);
  * batch shuffle for BatchNorm (BLD:107-152): with SyncBatchNorm (the only mode the reference supports, MAIN_MOCO:297) the
    statistics are global and the ViT has no cross-sample op, so shuffling cannot change any value; ``shuffle_bn=True``
    reproduces the reference's collectives exactly, the default skips the 617 MB image all_gather.
The ``_noprediction_q`` variant (its line 175) is ``predict_keys=False``.
"""
import torch
import torch.nn as nn

from mfvit.arena import ParamArena
from mfvit.mlp import FusedReLUSlot, HipBatchNorm1d, HipLinear
from mfvit.moco_ops import ema_update_, l2_normalize, neg_logits, pos_logits


class MoCo(nn.Module):
    def __init__(self, base_encoder, args, dim=256, mlp_dim=4096, T=1.0, shuffle_bn=False, predict_keys=True):
        super().__init__()
        self.T = T
        self.K = 65536                                                            # BLD:25 (hard-coded)
        self.shuffle_bn = shuffle_bn
        self.predict_keys = predict_keys
        if not args.arch.startswith('vit'):
            raise NotImplementedError("only the ViT branch (BLD:28-30) is on the accelerated path; ResNet/MNASNet/DenseNet "
                                      "encoders (BLD:31-48) are out of scope (SURVEY.md §2 row 17)")
        self.base_encoder = base_encoder(num_classes=mlp_dim)
        self.momentum_encoder = base_encoder(num_classes=mlp_dim)
        self.precision = getattr(self.base_encoder, "precision", "bf16")
        self._build_projector_and_predictor_mlps(dim, mlp_dim)
        for param_b, param_m in zip(self.base_encoder.parameters(), self.momentum_encoder.parameters()):
            param_m.data.copy_(param_b.data)                                      # BLD:52-54
            param_m.requires_grad = False
        queue = nn.functional.normalize(torch.randn(dim, self.K), dim=0)          # BLD:57-58
        self.register_buffer("queue", queue.t().contiguous().t())                 # (dim, K) view of key-major storage
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self._proj_arenas = None

    # ------------------------------------------------------------------ MLPs (BLD:62-78)
    def _build_mlp(self, num_layers, input_dim, mlp_dim, output_dim, last_bn=True):
        mlp = []
        for l in range(num_layers):
            dim1 = input_dim if l == 0 else mlp_dim
            dim2 = output_dim if l == num_layers - 1 else mlp_dim
            mlp.append(HipLinear(dim1, dim2, precision=self.precision))
            if l < num_layers - 1:
                mlp.append(HipBatchNorm1d(dim2, relu=True))
                mlp.append(FusedReLUSlot())
            elif last_bn:
                mlp.append(HipBatchNorm1d(dim2, affine=False, out_f32=True))      # SimCLR-style last BN, no affine
        return nn.Sequential(*mlp)

    def _build_projector_and_predictor_mlps(self, dim, mlp_dim):
        pass

    # ------------------------------------------------------------------ momentum update (BLD:83-89)
    def _arenas(self):
        if self._proj_arenas is None or not all(a.intact() for a in self._proj_arenas):
            pb = ParamArena(list(self.base_encoder.head.named_parameters()))
            pm = ParamArena(list(self.momentum_encoder.head.named_parameters()))
            self._proj_arenas = (pb, pm)
        return self._proj_arenas

    @torch.no_grad()
    def _momentum_update_key_encoder(self, m):
        ema_update_(self.momentum_encoder.flat_parameters(), self.base_encoder.flat_parameters(), m,
                    self.momentum_encoder._arena_params)
        pb, pm = self._arenas()
        ema_update_(pm.ensure(), pb.ensure(), m, pm.params)

    # ------------------------------------------------------------------ queue (BLD:91-105)
    def _queue_t(self):
        q = self.queue.t()
        if not q.is_contiguous():            # a checkpoint / .to() may have re-laid it out: restore key-major storage
            self.queue = q.contiguous().t()
            q = self.queue.t()
        return q

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        keys = concat_all_gather(keys)
        batch_size = keys.shape[0]
        ptr = int(self.queue_ptr)
        assert self.K % batch_size == 0  # for simplicity                        # BLD:99
        self._queue_t()[ptr:ptr + batch_size].copy_(keys)                         # == queue[:, ptr:ptr+bs] = keys.T
        self.queue_ptr[0] = (ptr + batch_size) % self.K

    # ------------------------------------------------------------------ shuffle BN (BLD:107-152)
    @torch.no_grad()
    def _batch_shuffle_ddp(self, x):
        batch_size_this = x.shape[0]
        x_gather = concat_all_gather(x)
        batch_size_all = x_gather.shape[0]
        num_gpus = batch_size_all // batch_size_this
        idx_shuffle = torch.randperm(batch_size_all, device=x.device)
        torch.distributed.broadcast(idx_shuffle, src=0)
        idx_unshuffle = torch.argsort(idx_shuffle)
        gpu_idx = torch.distributed.get_rank()
        idx_this = idx_shuffle.view(num_gpus, -1)[gpu_idx]
        return x_gather[idx_this], idx_unshuffle

    @torch.no_grad()
    def _batch_unshuffle_ddp(self, x, idx_unshuffle):
        batch_size_this = x.shape[0]
        x_gather = concat_all_gather(x)
        num_gpus = x_gather.shape[0] // batch_size_this
        gpu_idx = torch.distributed.get_rank()
        idx_this = idx_unshuffle.view(num_gpus, -1)[gpu_idx]
        return x_gather[idx_this]

    # ------------------------------------------------------------------ forward (BLD:154-199)
    def embed_queries(self, im_q):
        return l2_normalize(self.predictor(self.base_encoder(im_q)))              # BLD:164-165

    @torch.no_grad()
    def embed_keys(self, im_k, m):
        self._momentum_update_key_encoder(m)                                      # BLD:169
        shuffle = self.shuffle_bn and torch.distributed.is_available() and torch.distributed.is_initialized()
        if shuffle:
            im_k, idx_unshuffle = self._batch_shuffle_ddp(im_k)                   # BLD:172
        k = self.momentum_encoder(im_k)
        if self.predict_keys:
            k = self.predictor(k)                                                 # BLD:174 (shared predictor, train-mode BN: Q6)
        k = l2_normalize(k)                                                       # BLD:175
        if shuffle:
            k = self._batch_unshuffle_ddp(k, idx_unshuffle)                       # BLD:178
        return k

    def forward(self, im_q, im_k, m):
        q = self.embed_queries(im_q)
        k = self.embed_keys(im_k, m)
        l_pos = pos_logits(q, k)                                                  # (n, 1)   BLD:183
        l_neg = neg_logits(q, self._queue_t())                                    # (n, K)   BLD:185
        logits = torch.cat([l_pos, l_neg], dim=1)                                 # BLD:188
        logits /= self.T                                                          # BLD:191
        labels = torch.zeros(logits.shape[0], dtype=torch.long, device=logits.device)   # BLD:194
        self._dequeue_and_enqueue(k)                                              # BLD:197
        return logits, labels


class MoCo_ResNet(MoCo):
    def __init__(self, *a, **k):
        raise NotImplementedError("MoCo_ResNet (BLD:202-212) is the CNN path: out of scope (SURVEY.md §2 row 17)")


class MoCo_ViT(MoCo):
    def _build_projector_and_predictor_mlps(self, dim, mlp_dim):
        hidden_dim = self.base_encoder.head.weight.shape[1]                       # BLD:217
        del self.base_encoder.head, self.momentum_encoder.head                    # BLD:218
        self.base_encoder.head = self._build_mlp(3, hidden_dim, mlp_dim, dim)     # BLD:221-222
        self.momentum_encoder.head = self._build_mlp(3, hidden_dim, mlp_dim, dim)
        self.predictor = self._build_mlp(2, dim, mlp_dim, dim)                    # BLD:225


@torch.no_grad()
def concat_all_gather(tensor):
    """BLD:229-240: all_gather + cat along dim 0 (no gradient).  Without an initialised process group (single-GPU runs
    of this build's harness) the tensor is returned as is; the reference itself cannot run in that mode (Q7)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_world_size() == 1:
        return tensor
    world = torch.distributed.get_world_size()
    out = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    torch.distributed.all_gather_into_tensor(out, tensor.contiguous())
    return out
