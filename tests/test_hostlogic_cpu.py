"""CPU: host-side logic of the product - schedules vs the oracle, MoCo drop-in surface / state-dict layout vs the
reference-generated golden key list, optimizers' constructor surface, and the N > 1 paths on 2 gloo ranks
(GradSync flat exchange, concat_all_gather ordering, enqueue of gathered keys)."""
import os
import types
from functools import partial

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN
from oracle import ref_moco


def test_schedules_match_oracle():
    from mfvit import schedules
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    lr0, E, W = 1.5e-4 * 1024 / 4, 300, 40                       # MAIN_MOCO:286-288 lr scaling
    for e in (0.0, 0.5, 20.0, 39.99, 40.0, 41.3, 170.0, 299.9, 300.0):
        a = schedules.adjust_learning_rate(opt, e, lr0, E, W)
        assert a == ref_moco.adjust_learning_rate(e, lr0, E, W) == opt.param_groups[0]["lr"]
        assert schedules.adjust_moco_momentum(e, E, 0.99) == ref_moco.adjust_moco_momentum(e, E, 0.99)
    assert schedules.adjust_learning_rate(None, 65, 0.1, 90, 0, cos=False, schedule=(30, 60)) == \
        ref_moco.adjust_learning_rate(65, 0.1, 90, 0, cos=False, schedule=(30, 60))


def _moco(depth=1, mlp_dim=128, **kw):
    import vits
    import moco.builder_vit_mocov3structure_mocov2loss as bld
    return bld, bld.MoCo_ViT(partial(vits.vit_small, stop_grad_conv1=True, depth=depth), types.SimpleNamespace(arch="vit_small"),
                             256, mlp_dim, 0.2, **kw)


def test_moco_drop_in_surface_and_state_dict_layout():
    g = np.load(os.path.join(GOLDEN, "moco_pieces.npz"), allow_pickle=False)
    bld, m = _moco()
    for n in ("MoCo", "MoCo_ViT", "MoCo_ResNet", "concat_all_gather"):                    # MAIN_MOCO:35,273-282
        assert hasattr(bld, n)
    keys = set(m.state_dict().keys())
    ref_keys = {k for k in g["state_keys"] if k.startswith(("predictor", "queue")) or ".head." in k}
    assert ref_keys <= keys                                                                # BLD:215-225 layout, reference-generated
    assert "predictor.4.weight" not in keys and "base_encoder.head.7.weight" not in keys  # last BN affine=False
    assert m.K == 65536 and tuple(m.queue.shape) == (256, 65536) and m.queue_ptr.dtype == torch.long
    torch.testing.assert_close(m.queue.norm(dim=0), torch.ones(65536), rtol=1e-5, atol=1e-5)   # BLD:57-58
    for pb, pm in zip(m.base_encoder.parameters(), m.momentum_encoder.parameters()):       # BLD:52-54
        assert torch.equal(pb, pm) and not pm.requires_grad
    assert not m.base_encoder.patch_embed.proj.weight.requires_grad                        # stop_grad_conv1 (MAIN_MOCO:274)
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n_train == sum(p.numel() for n_, p in m.named_parameters()
                          if not n_.startswith("momentum_encoder") and p.requires_grad)
    with pytest.raises(NotImplementedError):
        bld.MoCo_ViT(lambda **k: None, types.SimpleNamespace(arch="resnet50"))
    import moco.builder_vit_mocov3structure_mocov2loss_noprediction_q as np_bld
    assert np_bld.MoCo_ViT is not bld.MoCo_ViT
    from moco.optimizer import LARS
    opt = LARS(m.parameters(), 0.3, weight_decay=1e-6, momentum=0.9)                       # MAIN_MOCO:335-337
    assert opt.defaults["trust_coefficient"] == 0.001
    from mfvit import MfvitError
    with pytest.raises(MfvitError):
        m(torch.zeros(2, 3, 224, 224), torch.zeros(2, 3, 224, 224), 0.99)                  # CPU tensors: loud failure


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mfvit.ddp import GradSync
        bld, m = _moco()
        # (1) GradSync: flat exchange of small gradients = mean over ranks
        ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
        for i, p in enumerate(ps):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        sync = GradSync()
        assert sync.enabled and sync.world == world and not sync.avg
        sync.reduce_grads(ps)
        sync.finish()
        ok1 = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(ps))
        # (2) encoder stage hook: per-block slices of a flat gradient arena are averaged, everything is covered exactly once
        vit = m.base_encoder
        sync.attach(vit)
        gflat = torch.full_like(vit.flat_parameters(), float(rank + 1))
        for s in range(vit.depth, -2, -1):                    # one stage per group
            vit._grad_stage_hook(vit, s, s, gflat)
        sync.finish()
        ok2 = bool(torch.allclose(gflat, torch.full_like(gflat, 1.5)))
        gflat = torch.full_like(vit.flat_parameters(), float(rank + 1))
        vit._grad_stage_hook(vit, vit.depth, -1, gflat)        # everything in one group
        sync.finish()
        ok2 = ok2 and bool(torch.allclose(gflat, torch.full_like(gflat, 1.5)))
        # (2b) joins are per owner, and the bf16 bucket option: cast -> all-reduce of the bf16 bucket -> write-back at the join
        other = torch.nn.Linear(1, 1)
        gflat = torch.full_like(vit.flat_parameters(), float(rank + 1))
        vit._grad_stage_hook(vit, vit.depth, -1, gflat)
        ok2 = ok2 and sync.pending(vit) == 3 and sync.pending(other) == 0
        sync.finish(other)
        ok2 = ok2 and sync.pending() == 3
        sync.finish(vit)
        ok2 = ok2 and sync.pending() == 0 and bool(torch.allclose(gflat, torch.full_like(gflat, 1.5)))
        sync16 = GradSync(bucket_dtype=torch.bfloat16)
        sync16.attach(vit)
        gen = torch.Generator().manual_seed(17 + rank)
        gflat = torch.randn(vit.flat_parameters().numel(), generator=gen)
        want = gflat.clone()
        dist.all_reduce(want)
        want /= world
        vit._grad_stage_hook(vit, vit.depth, -1, gflat)
        before = gflat.clone()
        sync16.finish(vit)
        err16 = float((gflat - want).abs().max() / want.abs().max())   # every rank's term is rounded to bf16 (2^-9 of ITS size), then the mean
        ok2 = ok2 and sync16._buckets[id(vit)].dtype == torch.bfloat16 and gflat.dtype == torch.float32 and err16 < 8e-3 \
            and not torch.equal(before, gflat)                # the f32 arena is written at the join, not before
        # (2c) reduce-scatter + all-gather exchange: same means as the all-reduce, bucket sizes that do and do not divide over
        # the ranks, one block per bucket, f32 and bf16 buckets
        for bdt in (None, torch.bfloat16):
            rs = GradSync(exchange="rs_ag", bucket_dtype=bdt)
            rs.attach(vit, bucket_layers=1)
            gen = torch.Generator().manual_seed(29 + rank)
            gflat = torch.randn(vit.flat_parameters().numel(), generator=gen)
            want = gflat.clone()
            dist.all_reduce(want)
            want /= world
            for s in range(vit.depth, -2, -1):
                vit._grad_stage_hook(vit, s, s, gflat)
            rs.finish(vit)
            e = float((gflat - want).abs().max() / want.abs().max())
            ok2 = ok2 and rs.pending() == 0 and e < (1e-6 if bdt is None else 8e-3)
        odd = torch.arange(7, dtype=torch.float32) * (rank + 1)           # 7 elements over 2 ranks: 6 by reduce-scatter, 1 by all-reduce
        for h in GradSync(exchange="rs_ag")._reduce_async(odd):
            h.wait()
        ok2 = ok2 and bool(torch.allclose(odd, torch.arange(7, dtype=torch.float32) * 1.5))
        vit._grad_stage_hook = None
        # (3) concat_all_gather order + enqueue of the gathered keys (BLD:91-105, 229-240)
        keys = torch.nn.functional.normalize(torch.full((4, 256), float(rank + 1)) + torch.arange(4).float()[:, None], dim=1)
        allk = bld.concat_all_gather(keys)
        ok3 = allk.shape == (8, 256) and torch.equal(allk[rank * 4:(rank + 1) * 4], keys)
        m.queue_ptr[0] = m.K - 8
        m._dequeue_and_enqueue(keys)
        ok4 = int(m.queue_ptr) == 0 and torch.equal(m.queue[:, m.K - 8:], allk.t())
        q.put((rank, ok1, ok2, bool(ok3), bool(ok4)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_paths():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3, ok4 in res:
        assert ok1 and ok2 and ok3 and ok4, (rank, ok1, ok2, ok3, ok4)


def _worker8(rank, world, port, q):
    """The N > 1 host paths at the world size configs[3] / [4] run at (8 ranks; VERDICT r5 task 6): every rank-count-dependent piece of
    arithmetic - shard sizes, remainders, pointer wrap, permutation slices, the order of Chan's combine - with W = 8 instead of 2."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mfvit.ddp import GradSync
        bld, m = _moco()
        ok, notes = {}, {}
        vit = m.base_encoder
        n_arena = vit.flat_parameters().numel()
        mean_of_ranks = (world + 1) / 2.0                       # ranks hold rank + 1
        # (1) all-reduce and reduce-scatter + all-gather, f32 and bf16 buckets, one block per bucket and everything in one bucket;
        #     bucket sizes that do NOT divide over 8 ranks (the arena slices of a block: 1,774,464 = 8 x 221,808 does; the embedding / head
        #     groups and the explicit odd tensors below do not)
        for exch in ("allreduce", "rs_ag"):
            for bdt in (None, torch.bfloat16):
                for layers in (1, 4):
                    sy = GradSync(exchange=exch, bucket_dtype=bdt)
                    sy.attach(vit, bucket_layers=layers)
                    gen = torch.Generator().manual_seed(100 + rank)
                    gflat = torch.randn(n_arena, generator=gen)
                    want = gflat.clone()
                    dist.all_reduce(want)
                    want /= world
                    if layers == 1:
                        for s in range(vit.depth, -2, -1):
                            vit._grad_stage_hook(vit, s, s, gflat)
                    else:
                        vit._grad_stage_hook(vit, vit.depth, -1, gflat)
                    pend = sy.pending(vit)
                    other = torch.nn.Linear(1, 1)
                    sy.finish(other)                              # joins are per owner: somebody else's join leaves these on the links
                    held = sy.pending(vit) == pend and pend > 0
                    sy.finish(vit)
                    e = float((gflat - want).abs().max() / want.abs().max())
                    # bf16 buckets: each of the 8 terms is rounded to bf16 (2^-9 of ITS size) before the sum, the mean once more
                    ok[f"{exch}/{'bf16' if bdt else 'f32'}/{layers}"] = held and sy.pending() == 0 and e < (2e-6 if bdt is None else 1.6e-2)
                    notes[f"{exch}/{'bf16' if bdt else 'f32'}/{layers}"] = e
        vit._grad_stage_hook = None
        for n in (5, 13, 8 * 3 + 7, 64):                          # fewer elements than ranks; 8 + 5 remainder; 24 + 7; divisible
            for bdt in (torch.float32, torch.bfloat16):
                t = (torch.arange(n, dtype=torch.float32) * (rank + 1)).to(bdt)
                for h in GradSync(exchange="rs_ag")._reduce_async(t):
                    h.wait()
                ok[f"odd{n}/{bdt}"] = bool(torch.allclose(t.float(), torch.arange(n, dtype=torch.float32) * mean_of_ranks, rtol=1e-2 if bdt is torch.bfloat16 else 1e-6))
        ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
        for i, p in enumerate(ps):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        GradSync().reduce_grads(ps)
        ok["flat"] = all(torch.allclose(p.grad, torch.full_like(p, mean_of_ranks * (i + 1))) for i, p in enumerate(ps))
        # (2) SyncBN: the product's rank-local triple, its all-gather layout (mfvit/mlp.py::_BNFn.forward) and Chan's combine over 8 partial
        #     statistics in rank order against float64 BatchNorm on the concatenated batch.  Unequal means per rank (what shuffle-BN exists for).
        n, C = 16, 24
        gen = torch.Generator().manual_seed(7)
        xall = torch.randn(world * n, C, generator=gen, dtype=torch.float64) * torch.linspace(0.5, 3.0, C, dtype=torch.float64) \
            + torch.arange(world, dtype=torch.float64).repeat_interleave(n)[:, None] * 0.7
        st = ref_moco.bn_partial_stats(xall[rank * n:(rank + 1) * n].float())
        flat = torch.empty(world * (2 * C + 1))
        dist.all_gather_into_tensor(flat, st)
        mu, var, invstd, uvar = ref_moco.bn_chan_combine(flat, C)
        y_ref, mu_ref, var_ref = ref_moco.batchnorm_train(xall, None, None)
        e_bn = max(float((mu.double() - mu_ref).abs().max() / mu_ref.abs().max()), float((var.double() - var_ref).abs().max() / var_ref.abs().max()),
                   float((uvar.double() - xall.var(0, unbiased=True)).abs().max() / var_ref.abs().max()))
        ok["syncbn"] = e_bn < 5e-6 and float(flat.view(world, -1)[:, 2 * C].sum()) == world * n
        notes["syncbn"] = e_bn
        # (3) concat_all_gather order, enqueue of N_all = 8 n keys with a pointer that wraps, and a pointer in the middle (BLD:91-105, 229-240)
        nk = 4
        keys = torch.nn.functional.normalize(torch.full((nk, 256), float(rank + 1)) + torch.arange(nk).float()[:, None], dim=1)
        allk = bld.concat_all_gather(keys)
        ok["gather"] = allk.shape == (world * nk, 256) and torch.equal(allk[rank * nk:(rank + 1) * nk], keys)
        q0 = m.queue.clone()
        m.queue_ptr[0] = m.K - world * nk
        m._dequeue_and_enqueue(keys)
        ok["enqueue_wrap"] = int(m.queue_ptr) == 0 and torch.equal(m.queue[:, m.K - world * nk:], allk.t()) and \
            torch.equal(m.queue[:, :m.K - world * nk], q0[:, :m.K - world * nk])
        m._dequeue_and_enqueue(keys * 0.5)
        ok["enqueue_next"] = int(m.queue_ptr) == world * nk and torch.equal(m.queue[:, :world * nk], allk.t() * 0.5)
        # (4) shuffle / unshuffle (BLD:107-152) at W = 8: rank 0's permutation everywhere, every sample exactly once, the round trip is the identity
        x = torch.arange(rank * nk, (rank + 1) * nk, dtype=torch.float32)[:, None].repeat(1, 3)
        torch.manual_seed(1000 + rank)                             # different local RNG states: the broadcast must win
        xs, idx_un = m._batch_shuffle_ddp(x)
        seen = bld.concat_all_gather(xs)[:, 0].long()
        back = m._batch_unshuffle_ddp(xs, idx_un)
        idx0 = idx_un.clone()
        dist.broadcast(idx0, src=0)
        ok["shuffle"] = xs.shape == x.shape and sorted(seen.tolist()) == list(range(world * nk)) and torch.equal(back, x) and torch.equal(idx0, idx_un) \
            and seen.tolist() != list(range(world * nk))
        q.put((rank, ok, notes, None))
    except Exception as e:   # noqa: BLE001 - reported to the parent, which fails the test
        import traceback
        q.put((rank, None, None, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_paths():
    """World size 8 (configs[3] / [4]; BLD:91-152,229-240): GradSync all-reduce and rs_ag in f32 and bf16 buckets with bucket sizes that do not
    divide by 8 and per-owner joins, SyncBN's Chan combine over 8 partial statistics against float64 BatchNorm on the concatenated batch, the MoCo
    enqueue with N_all = 8 n including the wrap, the shuffle / unshuffle permutation round trip."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, ok, notes, err in res:
        assert err is None, f"rank {rank}: {err}"
        assert all(ok.values()), (rank, {k: v for k, v in ok.items() if not v}, notes)


def test_checkpoint_layouts_and_pretrained_load(tmp_path):
    """the dict layouts the drivers save (MAIN_MOCO:461-467, MAIN_SS:567-575, MAIN_CA:712-720, 1002-1011) and the
    key surgery that loads a MoCo pretraining checkpoint into a finetune backbone (MAIN_SS:326-337), on this package's modules."""
    import vits
    from mfvit import checkpoint as ck
    from moco.optimizer import LARS
    bld, m = _moco()
    opt = LARS(m.parameters(), 0.3, weight_decay=1e-6, momentum=0.9)
    state = ck.pretrain_checkpoint(m, opt, epoch=4, arch="vit_small")
    assert list(state.keys()) == ["epoch", "arch", "state_dict", "optimizer"] and state["epoch"] == 5
    # what DistributedDataParallel would have saved: every key behind 'module.'
    state["state_dict"] = {"module." + k: v for k, v in state["state_dict"].items()}
    path = ck.save_checkpoint(str(tmp_path), state, is_best=False, filename="checkpoint_0004.pth.tar")
    assert os.path.basename(path) == "checkpoint_0004.pth.tar"
    assert os.path.basename(ck.save_checkpoint(str(tmp_path), state, is_best=True)) == "model_best.pth.tar"   # MAIN_CA:1006-1009

    torch.manual_seed(123)
    ft = vits.vit_small(num_classes=3, depth=1)                                              # MAIN_SS:276 + :309-style 3-class head (depth 1 as _moco())
    msg = ck.load_pretrained_backbone(ft, path)
    assert set(msg.missing_keys) == {"head.weight", "head.bias"} and not msg.unexpected_keys
    sd = ft.state_dict()
    for k, v in m.base_encoder.state_dict().items():
        if not k.startswith("head."):
            assert torch.equal(sd[k], v), k
    assert ck.sanity_check(ft.state_dict(), path)                                            # nothing but the head differs
    with torch.no_grad():
        ft.blocks[0].mlp.fc1.bias.add_(1.0)
    with pytest.raises(AssertionError, match="is changed in linear classifier training"):
        ck.sanity_check(ft.state_dict(), path)
    assert ck.sanity_check(ft.state_dict(), path, semi_supervised=True)                      # skipped, MAIN_CA:1018-1020

    # a model trained with this package's GradSync is not wrapped in DDP: the bare 'base_encoder.*' layout loads the same way
    bare = ck.pretrain_checkpoint(m, opt, epoch=4, arch="vit_small")
    assert not any(k.startswith("module.") for k in bare["state_dict"])
    ft2 = vits.vit_small(num_classes=3, depth=1)
    msg2 = ck.load_pretrained_backbone(ft2, bare)
    assert set(msg2.missing_keys) == {"head.weight", "head.bias"} and not msg2.unexpected_keys
    assert ck.sanity_check(ft2.state_dict(), bare)
    # optimizer state dicts never carry the HIP optimizers' runtime caches (raw device addresses)
    assert all(not k.startswith("_mfvit") for g in state["optimizer"]["param_groups"] for k in g)

    opt2 = torch.optim.SGD(ft.parameters(), lr=0.1)
    fin = ck.finetune_checkpoint(ft, opt2, epoch=0, arch="vit_small", best_metric_val=0.9, best_metric_val_test=0.8, best_metric_test=0.85)
    assert list(fin.keys()) == ["epoch", "arch", "state_dict", "best_metric_val_test", "best_metric_val", "best_metric_test", "optimizer"]
    fin_ca = ck.finetune_checkpoint(ft, opt2, epoch=0, arch="vit_small", best_metric_val=0.9)
    assert list(fin_ca.keys()) == ["epoch", "arch", "state_dict", "best_metric_val", "optimizer"]
    # the stripped dict keeps nothing of the momentum encoder / predictor / queue
    stripped = ck.strip_moco_prefix(dict(torch.load(path, map_location="cpu")["state_dict"]))
    assert all(not k.startswith(("module.", "head.")) for k in stripped) and "cls_token" in stripped


def test_moco_v3_builder_surface():
    """moco/builder_vit.py drop-in (symmetric loss): constructor, module names and state-dict layout against the reference-generated
    key list (projector / predictor keys; the toy encoder body of the generator is not part of the comparison), no queue buffers."""
    import vits
    import moco.builder_vit as bv
    g = np.load(os.path.join(GOLDEN, "moco_v3.npz"), allow_pickle=False)
    m = bv.MoCo_ViT(partial(vits.vit_small, stop_grad_conv1=True, depth=1), types.SimpleNamespace(arch="vit_small"), 256, 64, 0.2)
    for n in ("MoCo", "MoCo_ViT", "MoCo_ResNet", "concat_all_gather"):
        assert hasattr(bv, n)
    mine = {k for k in m.state_dict().keys() if k.startswith("predictor.") or ".head." in k}
    ref = {k for k in g["state_keys"] if k.startswith("predictor.") or ".head." in k}
    assert mine == ref
    assert not any(k.startswith("queue") for k in m.state_dict().keys())
    for pb, pm in zip(m.base_encoder.parameters(), m.momentum_encoder.parameters()):
        assert torch.equal(pb, pm) and not pm.requires_grad
    with pytest.raises(NotImplementedError):
        bv.MoCo_ResNet(None, None)


def test_vmem_hazard_checker_flags_reads_of_registers_in_flight(tmp_path):
    """tools/check_vmem_hazards.py guards the inline-asm load pipeline of the tile GEMM (gemm.cuh::NtLoopDeep) at build time: it must
    accept a load -> counted wait -> use sequence and reject a copy of a register whose load is still outstanding."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_vmem_hazards.py")
    good = """_Z4kernv: ; @_Z4kernv
\t;;#ASMSTART
\tglobal_load_dwordx4 v[2:5], v1, s[0:1]
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[6:9], v1, s[2:3]
\t;;#ASMEND
\tv_add_u32_e32 v10, v11, v12
\t;;#ASMSTART
\ts_waitcnt vmcnt(1)
\t;;#ASMEND
\tds_write_b128 v20, v[2:5]
\ts_waitcnt vmcnt(0)
\tv_mfma_f32_32x32x16_bf16 v[32:47], v[6:9], v[6:9], v[32:47]
\ts_endpgm
"""
    bad = good.replace("\tv_add_u32_e32 v10, v11, v12\n", "\tv_mov_b32_e32 v10, v7\n")
    # a load the COMPILER issued and counts itself (outside the asm markers) is not a hazard candidate, but keeps its vmcnt slot;
    # an LDS-DMA has no register destination
    own = good.replace("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[6:9], v1, s[2:3]\n\t;;#ASMEND\n", "\tglobal_load_dwordx4 v[6:9], v1, s[2:3]\n") \
        .replace("\tv_add_u32_e32 v10, v11, v12\n", "\tv_mov_b32_e32 v10, v7\n")
    dma = good.replace("\tv_add_u32_e32 v10, v11, v12\n", "\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v1, s[2:3]\n\t;;#ASMEND\n\tv_mov_b32_e32 v1, v30\n") \
        .replace("s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(2)")
    for i, (text, rc) in enumerate(((good, 0), (own, 0), (dma, 0), (bad, 1))):
        f = tmp_path / f"k{i}.s"
        f.write_text(text)
        r = subprocess.run([sys.executable, tool, str(f), "kern"], capture_output=True, text=True)
        assert r.returncode == rc, (i, rc, r.stdout, r.stderr)
    assert "HAZARD" in r.stdout


def test_shipped_kernels_with_asm_loads_pass_the_hazard_scan():
    """The device assembly that build() keeps next to the objects (build/<name>.s, produced with the flags of the objects that ship) of
    every kernel that issues loads from inline asm - the deep-pipelined tile GEMM, the tall-tile row kernel, the ping-pong GEMM - is
    scanned on every build(); this test repeats the scan on whatever assembly is present (no recompilation: the driver's build check has
    produced it)."""
    import __graft_entry__ as ge
    objdir = os.path.join(ge.PKG, "build")
    present = [n for n in ge.ASM_LOAD_KERNELS if os.path.exists(os.path.join(objdir, n[:-4] + ".s"))]
    if not present:
        pytest.skip("no device assembly under build/ (library not built here)")
    ge.check_asm_load_hazards(objdir, regenerate=False)


def test_mfma_asm_hazard_checker_flags_asm_reads_of_accumulators_in_flight(tmp_path):
    """tools/check_mfma_asm_hazards.py (round 6; run by build() on every device-assembly file): an instruction inside an asm statement that reads an MFMA
    destination within the MFMA's wait states is a finding - directly behind it or behind a branch; the compiler's own reads (it places the s_nops), a read
    behind enough wait states, and a register a compiler instruction has overwritten in between are not."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_mfma_asm_hazards.py")
    head = "_Z4kernv: ; @_Z4kernv\n\tv_mfma_f32_32x32x16_f16 v[46:61], v[0:3], v[4:7], v[46:61]\n"
    read = "\t;;#ASMSTART\n\tv_max3_f32 v4, v46, v47, v48\n\t;;#ASMEND\n\ts_endpgm\n"
    cases = [(head + read, 1),                                                                                    # right behind the MFMA
             (head + "\ts_cbranch_vccnz .LBB0_2\n\tv_mov_b32_e32 v9, v8\n.LBB0_2:\n" + read, 1),               # behind a taken branch
             (head + "\ts_nop 15\n\ts_nop 3\n" + read, 0),                                                     # 20 wait states later
             (head + "\ts_nop 10\n\tv_max_f32_e32 v46, v46, v46\n\tv_max_f32_e32 v47, v47, v47\n\tv_max_f32_e32 v48, v48, v48\n" + read, 0),   # rewritten by the compiler
             (head + "\ts_nop 10\n\tv_max3_f32 v4, v46, v47, v48\n\ts_endpgm\n", 0)]                          # the compiler's own read
    # the two neighbouring hazards: a transcendental result read by asm in the next issue slot; a VALU-written SGPR read by an asm LDS-DMA within 5 wait states
    k = "_Z4kernv: ; @_Z4kernv\n"
    cvt = "\t;;#ASMSTART\n\tv_cvt_pk_f16_f32 v6, v5, v7\n\t;;#ASMEND\n\ts_endpgm\n"
    dma = "\t;;#ASMSTART\n\ts_mov_b32 s9, m0\n\ts_mov_b32 m0, s8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v2, s[4:5]\n\ts_mov_b32 m0, s9\n\t;;#ASMEND\n\ts_endpgm\n"
    cases += [(k + "\tv_exp_f32_e32 v5, v4\n" + cvt, 1),
              (k + "\tv_exp_f32_e32 v5, v4\n\tv_add_f32_e32 v9, v9, v5\n" + cvt, 0),
              (k + "\tv_readfirstlane_b32 s4, v1\n\tv_readfirstlane_b32 s5, v2\n" + dma, 1),          # 3 wait states inside the statement + 1
              (k + "\tv_readfirstlane_b32 s4, v1\n\tv_readfirstlane_b32 s5, v2\n\ts_nop 1\n" + dma, 0),
              (k + "\tv_readfirstlane_b32 s4, v1\n\ts_mov_b32 s4, s20\n\ts_mov_b32 s5, s21\n" + dma, 0)]   # rewritten by the scalar unit in between
    for i, (text, rc) in enumerate(cases):
        f = tmp_path / f"k{i}.s"
        f.write_text(text)
        r = subprocess.run([sys.executable, tool, str(f)], capture_output=True, text=True)
        assert r.returncode == rc, (i, rc, r.stdout, r.stderr)


def test_shipped_hot_kernels_use_no_scratch():
    """The hot split-bf16 kernels (tile GEMM, tall-tile row kernels, LDS-DMA weight gradients, persistent attention) must compile without scratch
    memory - epilogues included.  Round 5 shipped a LayerNorm-backward row kernel with 148 bytes of spills in its epilogue for most of the round
    (one asm store statement tipped the register allocator): + 10 us per launch, found only by a same-box A/B against the round-4 library.  Checked
    on the device assembly build() keeps under build/ (skipped when the library was not built here)."""
    import __graft_entry__ as ge
    objdir = os.path.join(ge.PKG, "build")
    present = {n: rx for n, rx in ge.NO_SCRATCH_KERNELS.items() if os.path.exists(os.path.join(objdir, n[:-4] + ".s"))}
    if not present:
        pytest.skip("no device assembly under build/ (library not built here)")
    for n, rx in present.items():
        assert ge.kernels_with_scratch(os.path.join(objdir, n[:-4] + ".s"), rx) == [], n


def test_inline_asm_statements_declare_what_they_write(tmp_path):
    """tools/check_inline_asm.py: an asm statement that writes SCC / vcc must clobber it, m0 / exec must be restored inside the statement.
    (Round 4: `s_and_b64 exec, exec, vcc` without an "scc" clobber corrupted a compare the compiler held in SCC - only once an unrelated
    branch made it do so.)  The shipped sources pass; the statement of the bug and a vcc variant are flagged."""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "check_inline_asm.py")
    srcs = sorted(glob.glob(os.path.join(root, "multi-feature-vit_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "multi-feature-vit_amd", "csrc", "*.cuh")))
    assert srcs
    r = subprocess.run([sys.executable, tool] + srcs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    bad = tmp_path / "bad.hip"
    bad.write_text('void f() {\n'
                   '  asm volatile("s_mov_b64 %0, exec\\n\\tv_cmp_gt_i32 vcc, %1, %2\\n\\ts_and_b64 exec, exec, vcc\\n\\ts_mov_b64 exec, %0" : "=&s"(k) : "s"(a), "v"(b) : "memory", "vcc");\n'
                   '  asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "s"(a), "v"(b) : "memory");\n'
                   '  asm volatile("s_mov_b32 m0, %0\\n\\tglobal_load_lds_dwordx4 %1, off" :: "s"(a), "v"(b) : "memory");\n'
                   '  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(a), "v"(b) : "memory");\n'
                   '}\n')
    r = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1
    assert "writes SCC" in r.stdout and "writes vcc" in r.stdout and "writes m0" in r.stdout, r.stdout
    # round 5: a > 64-bit store inside an asm statement needs its wait states inside the statement (the data registers are read late)
    assert "store of more than 64 bits" in r.stdout, r.stdout
    ok = tmp_path / "ok.hip"
    ok.write_text('void f() { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\\n\\ts_nop 1" :: "v"(a), "v"(b) : "memory"); }\n')
    assert subprocess.run([sys.executable, tool, str(ok)], capture_output=True, text=True).returncode == 0


def test_loop_drain_scanner_flags_vmcnt0_and_spills_inside_mfma_loops(tmp_path):
    """tools/scan_loop_waits.py (run by build() over the hot LDS-DMA kernels): a `vmcnt(0)` or a scratch access inside an innermost MFMA loop is
    a finding - hipcc cannot count asm loads, its own waits drain the whole queue; a counted wait and a drain outside the loop are not."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "scan_loop_waits.py")
    mf = "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n" * 12

    def kernel(name, body):
        return f"{name}:\n\ts_waitcnt vmcnt(0)\n.LBB0_1:\n{mf}{body}\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n"
    good, bad1, bad2 = tmp_path / "good.s", tmp_path / "bad1.s", tmp_path / "bad2.s"
    good.write_text(kernel("_Z4hotAv", "\ts_waitcnt vmcnt(4)\n"))
    bad1.write_text(kernel("_Z4hotBv", "\ts_waitcnt vmcnt(0)\n"))
    bad2.write_text(kernel("_Z4hotCv", "\tscratch_load_dword v1, off, off offset:8\n"))
    run = lambda f, rx="hot": subprocess.run([sys.executable, tool, "--fail", rx, str(f)], capture_output=True, text=True)
    assert run(good).returncode == 0
    assert run(bad1).returncode == 1 and "vmcnt(0)" in run(bad1).stdout
    assert run(bad2).returncode == 1 and "scratch_load" in run(bad2).stdout
    assert run(bad1, "other_kernel").returncode == 2                  # a pattern that matches NO kernel is an error of its own (a stale regex passed vacuously)

