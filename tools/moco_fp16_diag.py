"""Per-tensor gradient error of the MoCo step against the f64 oracle, per precision (diagnostic for tests/test_moco_gpu.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest  # noqa: F401  (puts the package on sys.path)
import torch
import test_moco_gpu as t
from conftest import rng_tensor
from oracle import ref_vit
from mfvit.moco_ops import cross_entropy_rows

DEV = "cuda:0"
for precision in sys.argv[1:] or ["fp16", "bf16", "fp32"]:
    depth, mlp_dim, dim, T, n, mval = 2, 512, 256, 0.2, 8, 0.99
    m = t.make_moco(depth=depth, mlp_dim=mlp_dim, dim=dim, T=T, predict_keys=True, precision=precision)
    with torch.no_grad():
        m.base_encoder.load_state_dict(ref_vit.seeded_params(701, num_classes=0, depth=depth), strict=False)
        m.momentum_encoder.load_state_dict(ref_vit.seeded_params(702, num_classes=0, depth=depth), strict=False)
        for i, (_, p) in enumerate(list(m.base_encoder.head.named_parameters()) + list(m.predictor.named_parameters())
                                   + list(m.momentum_encoder.head.named_parameters())):
            if p.ndim == 1:
                p.copy_(1.0 + 0.1 * rng_tensor(710 + i, p.shape) if "weight" in _ else 0.05 * rng_tensor(710 + i, p.shape))
            else:
                p.copy_(rng_tensor(710 + i, p.shape) / p.shape[1] ** 0.5)
    m = m.to(DEV).train()
    im_q, im_k = rng_tensor(720, (n, 3, 224, 224)), rng_tensor(721, (n, 3, 224, 224))
    ref, ref_grad, gmax = t._oracle_step(m, im_q, im_k, mval, T, True)
    logits, labels = m(im_q.to(DEV), im_k.to(DEV), mval)
    loss = cross_entropy_rows(logits, labels)
    (loss * 4096.0).backward()
    print(f"== {precision}: logits err {t.scale_err(logits, ref['logits']):.2e}  gmax {gmax:.3e}")
    rows = []
    for name, p in m.named_parameters():
        rg = ref_grad(name)
        if rg is None or p.grad is None:
            continue
        g = (p.grad / 4096.0).double().cpu()
        rg = rg.double()
        rows.append((float((g - rg).abs().max() / rg.abs().max().clamp_min(1e-30)), float((g - rg).norm() / rg.norm().clamp_min(1e-30)),
                     float(rg.abs().max()), name))
    for mx, l2, rmax, name in sorted(rows, reverse=True)[:12]:
        print(f"   max-rel {mx:9.2e}  L2-rel {l2:9.2e}  |ref|max {rmax:9.2e} ({rmax / gmax:8.1e} of gmax)  {name}")
