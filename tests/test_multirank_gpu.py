"""world_size = 2 execution of the code that only runs with more than one rank, as written:
both ranks share cuda:0 and talk over gloo (RCCL needs one GPU per rank; the driver's 8-GPU node runs that).

  (a) HipBatchNorm1d with SyncBatchNorm semantics (MAIN_MOCO:297; mfvit/mlp.py: statistics all_gather + Chan combine, backward
      all_reduce) on two half batches == float64 BatchNorm on the concatenated batch (y, dx, running stats; dgamma / dbeta are the
      rank-local sums, as torch's SyncBatchNorm returns them, and add up to the full-batch gradient).
  (b) MoCo_ViT(shuffle_bn=True).forward (BLD:107-152: image all_gather, broadcast permutation, unshuffle) gives the logits and
      queue of shuffle_bn=False - the claim DESIGN.md makes for skipping the shuffle by default.
  (c) GradSync: the asynchronous per-bucket all-reduce issued from the encoder's backward hooks (the handles the RCCL path uses)
      == the mean over ranks of the unsynchronised gradients.
"""
import os
import types
from functools import partial

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rng_tensor

pytestmark = pytest.mark.gpu


def _err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _sync_bn(rank, world):
    from mfvit.mlp import HipBatchNorm1d
    n, C = 6, 256
    x_all = rng_tensor(901, (world * n, C), 1.5) + 0.3
    dy_all = rng_tensor(902, (world * n, C))
    gamma, beta = 1 + 0.1 * rng_tensor(903, (C,)), 0.1 * rng_tensor(904, (C,))
    # float64 reference on the concatenated batch
    xd = x_all.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    yr = torch.relu(torch.nn.functional.batch_norm(xd, rm, rv, gd, bd, training=True, momentum=0.1, eps=1e-5))
    yr.backward(dy_all.double())
    bn = HipBatchNorm1d(C, relu=True).to("cuda:0")
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    sl = slice(rank * n, (rank + 1) * n)
    x = x_all[sl].to("cuda:0").requires_grad_(True)
    y = bn(x)
    y.backward(dy_all[sl].to("cuda:0"))
    errs = dict(y=_err(y, yr[sl]), dx=_err(x.grad, xd.grad[sl]), running_mean=_err(bn.running_mean, rm), running_var=_err(bn.running_var, rv))
    # weight / bias gradients are rank-local sums (torch SyncBatchNorm convention; DDP then averages them): their sum is the full one
    dg, db = bn.weight.grad.clone(), bn.bias.grad.clone()
    dist.all_reduce(dg)
    dist.all_reduce(db)
    errs.update(dgamma=_err(dg, gd.grad), dbeta=_err(db, bd.grad))
    assert int(bn.num_batches_tracked) == 1
    return errs


def _moco_pair(rank, world):
    import vits
    import moco.builder_vit_mocov3structure_mocov2loss as bld
    models = []
    for shuffle in (False, True):
        torch.manual_seed(77)                                        # identical initialisation (and queue) for both models, on both ranks
        m = bld.MoCo_ViT(partial(vits.vit_small, stop_grad_conv1=True, depth=1, precision="fp32"), types.SimpleNamespace(arch="vit_small"),
                         256, 128, 0.2, shuffle_bn=shuffle).to("cuda:0")
        models.append(m)
    n = 4
    x1 = rng_tensor(911 + rank, (n, 3, 224, 224)).to("cuda:0")
    x2 = rng_tensor(921 + rank, (n, 3, 224, 224)).to("cuda:0")
    outs = []
    for m in models:
        logits, labels = m(x1, x2, 0.99)
        logits.float().sum().backward()
        outs.append((logits.detach(), m.queue.detach().clone(), int(m.queue_ptr), m.predictor[0].weight.grad.detach().clone()))
    (l0, q0, p0, g0), (l1, q1, p1, g1) = outs
    assert p0 == p1 == world * n and l0.shape == (n, 1 + 65536)
    # the first world * n queue columns are the gathered keys, in rank order (BLD:94-102)
    return dict(logits=_err(l1, l0), queue=_err(q1[:, :world * n], q0[:, :world * n]), rest_untouched=float((q1[:, world * n:] - q0[:, world * n:]).abs().max()),
                pred_grad=_err(g1, g0))


def _grad_sync(rank, world):
    import vits
    from mfvit.ddp import GradSync
    torch.manual_seed(5)
    m = vits.vit_small(num_classes=3, depth=6, precision="fp32").to("cuda:0")
    x = rng_tensor(931 + rank, (2, 3, 224, 224)).to("cuda:0")
    w = rng_tensor(941 + rank, (2, 197, 384)).to("cuda:0")

    def grads():
        for p in m.parameters():
            p.grad = None
        (m.features3D(x) * w).sum().backward()
        return m._last_grad_arena.detach().clone()

    local = grads()
    want = local.clone()
    dist.all_reduce(want)
    want /= world
    sync = GradSync()
    assert sync.enabled and sync.world == world
    sync.attach(m, bucket_layers=2)                                  # 3 block groups + embedding / norm pieces: 5 asynchronous handles
    seen = []
    orig = sync._push
    sync._push = lambda h: (seen.append(h), orig(h))[1]
    for p in m.parameters():
        p.grad = None
    (m.features3D(x) * w).sum().backward()
    assert len(seen) >= 5 and all(hasattr(h, "wait") for h in seen)  # handles of async_op=True collectives, issued by the hooks
    sync.finish()
    got = m._last_grad_arena
    return dict(sync=_err(got, want), differs_from_local=float((want - local).abs().max()) > 0)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = dict(bn=_sync_bn(rank, world), moco=_moco_pair(rank, world), sync=_grad_sync(rank, world))
        q.put((rank, res, None))
    except Exception as e:   # noqa: BLE001 - reported to the parent, which fails the test
        import traceback
        q.put((rank, None, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_syncbn_shufflebn_gradsync():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, r, err in res:
        assert err is None, f"rank {rank}: {err}"
        report = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_multirank.txt")
        os.makedirs(os.path.dirname(report), exist_ok=True)
        with open(report, "a") as f:
            f.write(f"rank {rank}: {r}\n")
        for k, v in r["bn"].items():
            assert v < 2e-5, (rank, "bn", k, v)
        assert r["moco"]["logits"] < 1e-4 and r["moco"]["queue"] < 1e-4 and r["moco"]["rest_untouched"] == 0.0, (rank, r["moco"])
        assert r["moco"]["pred_grad"] < 1e-3, (rank, r["moco"])
        assert r["sync"]["sync"] < 1e-6 and r["sync"]["differs_from_local"], (rank, r["sync"])
