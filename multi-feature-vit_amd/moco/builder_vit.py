"""Drop-in for the reference's MoCo-v3 builder (moco_pretraining/moco/moco/builder_vit.py, BV below): ``MoCo`` / ``MoCo_ViT`` with
the constructor ``(base_encoder, args, dim=256, mlp_dim=4096, T=1.0)`` and ``forward(x1, x2, m) -> loss`` - the SYMMETRIC
contrastive loss without a queue - on the same gfx950 kernels as the queue-based builder next to it.

    q_i = predictor(base_encoder(x_i));  k_i = momentum_encoder(x_i) (no gradient, after the momentum update)
    loss = ctr(q1, k2) + ctr(q2, k1),  ctr(q, k) = CE(normalize(q) @ all_gather(normalize(k))^T / T, arange(N) + N * rank) * 2T

State-dict keys are BV's: base_encoder.*, momentum_encoder.*, predictor.* (no queue buffers).
"""
import torch

from mfvit.moco_ops import cross_entropy_rows, l2_normalize, neg_logits
from moco import builder_vit_mocov3structure_mocov2loss as _q


class MoCo(_q.MoCo):
    def __init__(self, base_encoder, args, dim=256, mlp_dim=4096, T=1.0):
        super().__init__(base_encoder, args, dim, mlp_dim, T)
        del self.queue, self.queue_ptr                                            # BV has no queue (BV:23-55)
        self.K = 0

    def _update_momentum_encoder(self, m):                                        # BV:78-82 (one launch per flat arena)
        self._momentum_update_key_encoder(m)

    def contrastive_loss(self, q, k):
        """BV:84-94.  q carries the gradient; k comes from the momentum encoder (no gradient, as in BV:109-115)."""
        q = l2_normalize(q)
        k = l2_normalize(k.detach())
        k = concat_all_gather(k)                                                  # BV:89
        n, n_all = q.shape[0], k.shape[0]
        pad = (-n_all) % 128                                                      # the MFMA tile kernel wants N % 128 == 0
        if pad:
            k = torch.nn.functional.pad(k, (0, 0, 0, pad))
        logits = neg_logits(q, k.contiguous())                                    # (n, n_all + pad) = q @ k^T, gradient to q only
        if pad:
            logits = logits[:, :n_all]
        logits = logits / self.T                                                  # BV:91
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        labels = torch.arange(n, dtype=torch.long, device=q.device) + n * rank    # BV:93
        return cross_entropy_rows(logits.contiguous(), labels) * (2 * self.T)     # BV:94

    def forward(self, x1, x2, m):
        q1 = self.predictor(self.base_encoder(x1))                                # BV:107-108
        q2 = self.predictor(self.base_encoder(x2))
        with torch.no_grad():
            self._update_momentum_encoder(m)                                      # BV:111
            k1 = self.momentum_encoder(x1)                                        # BV:114-115
            k2 = self.momentum_encoder(x2)
        return self.contrastive_loss(q1, k2) + self.contrastive_loss(q2, k1)      # BV:117


class MoCo_ResNet(MoCo):
    def __init__(self, *a, **k):
        raise NotImplementedError("MoCo_ResNet (BV:120-131) is the CNN path: out of scope (SURVEY.md §2 row 17)")


class MoCo_ViT(MoCo):
    def _build_projector_and_predictor_mlps(self, dim, mlp_dim):                  # BV:134-145
        hidden_dim = self.base_encoder.head.weight.shape[1]
        del self.base_encoder.head, self.momentum_encoder.head
        self.base_encoder.head = self._build_mlp(3, hidden_dim, mlp_dim, dim)
        self.momentum_encoder.head = self._build_mlp(3, hidden_dim, mlp_dim, dim)
        self.predictor = self._build_mlp(2, dim, mlp_dim, dim)


concat_all_gather = _q.concat_all_gather
