"""GPU: a whole CA train step replayed as ONE HIP graph (mfvit.graph.GraphedStep, mfvit.optim.Adam(capturable=True); VERDICT r4 task 4a) against the
same steps run eagerly: same losses, same weights after the same number of optimizer steps (the weight-gradient atomics make any two runs differ
at rounding level), Adam's device-side step count advancing with every replay."""
import importlib

import pytest
import torch

from conftest import rng_tensor
from oracle import ref_fusion, ref_vit

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FUS_MOD = ("model.crossvit_2vits_2additionaloutputs_changenormlayer_location_removeextralclayer_"
           "changemodelinputlocation_std002_sum")


def _build(capturable, depth=2):
    import vits_returnftrs as vits
    from mfvit.optim import Adam
    fus = importlib.import_module(FUS_MOD)
    backs = []
    for i in range(2):
        m = vits.vit_small(num_classes=3, depth=depth)
        m.load_state_dict(ref_vit.seeded_params(901 + i, num_classes=3, depth=depth))
        backs.append(m.to(DEV))
    model = fus.Fus_CrossViT(backs[0], backs[1])
    model.load_state_dict(ref_fusion.seeded_fusion_params(903))
    model = model.to(DEV)
    params = list(model.parameters()) + [p for m in backs for p in m.parameters() if p.requires_grad]
    opt = Adam(params, lr=1e-3, capturable=capturable)
    return model, backs, opt, params


def test_graphed_step_matches_eager_steps():
    from mfvit.graph import GraphedStep
    from mfvit.losses import cross_entropy
    B = 4
    x, xe = rng_tensor(911, (B, 3, 224, 224)).to(DEV), rng_tensor(912, (B, 3, 224, 224)).to(DEV)
    y = torch.tensor([0, 1, 2, 1], device=DEV)

    def make_step(model, backs, opt):
        def step():
            opt.zero_grad(set_to_none=True)
            fused, xc, xe_ = model(backs[0], backs[1], x, xe)
            loss, _ = cross_entropy(fused + xc + xe_, y)
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    model, backs, opt, params = _build(False)
    w_0 = torch.cat([p.detach().reshape(-1) for p in params]).clone()
    eager = make_step(model, backs, opt)
    losses_e = [float(eager()) for _ in range(5)]
    w_e = torch.cat([p.detach().reshape(-1) for p in params]).clone()

    model, backs, opt, params = _build(True)
    gs = GraphedStep(make_step(model, backs, opt), warmup=2)          # 2 eager steps, then one capture (recorded, not executed)
    losses_g = [float(gs()) for _ in range(3)]                          # steps 3, 4, 5
    torch.cuda.synchronize()
    assert opt.step_count(0) == 5
    assert all(abs(a - b) < 2e-4 * max(1.0, abs(a)) for a, b in zip(losses_e[2:], losses_g)), (losses_e, losses_g)
    assert losses_g[-1] < losses_g[0]                                   # it trains: every replay applied an update
    # the two runs took the same five updates: the distance between their weights is small against the distance both travelled (Adam turns
    # rounding-noise gradient elements into +- lr steps whose sign differs between ANY two runs - the weight-gradient atomics - so single
    # elements, and tensors as small as the 1e-6 cls token, differ by a few lr)
    w_g = torch.cat([p.detach().reshape(-1) for p in params])
    moved, apart = float((w_e - w_0).norm()), float((w_g - w_e).norm())
    assert apart < 0.05 * moved, (apart, moved)


def test_capturable_adam_matches_the_host_stepped_one():
    from mfvit.optim import Adam, AdamW
    g = torch.Generator().manual_seed(5)
    for cls, kw in ((Adam, dict(weight_decay=0.01)), (AdamW, dict(weight_decay=0.1))):
        ps = [[torch.nn.Parameter(torch.randn(n, generator=g).to(DEV)) for n in (1000, 20000, 7)] for _ in range(2)]
        for a, b in zip(*ps):
            b.data.copy_(a.data)
        opts = [cls(ps[0], lr=3e-3, capturable=False, **kw), cls(ps[1], lr=3e-3, capturable=True, **kw)]
        for it in range(6):
            grads = [torch.randn(p.shape, generator=g).to(DEV) for p in ps[0]]
            for k in range(2):
                for p, gr in zip(ps[k], grads):
                    p.grad = gr.clone()
                if it == 3:
                    opts[k].param_groups[0]["lr"] = 1e-3                # a scheduler step between updates
                opts[k].step()
        assert opts[1].step_count(0) == 6
        for a, b in zip(*ps):
            assert float((a - b).abs().max()) < 1e-6 * float(a.abs().max()), cls.__name__
