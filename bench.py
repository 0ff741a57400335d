#!/usr/bin/env python3
"""Contract benchmark: images/sec of the two-stream 224^2 vit_small MF-CA train step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full pass of the hot path over one synthetic batch that is already resident in HBM:
  Fus_CrossViT forward (two ViT-S/16 encoders + cross-attention exchange + heads) -> output sum -> cross entropy ->
  backward -> (N>1: RCCL gradient all-reduce overlapped with backward) -> Adam step.
--mode T (default): both backbones trainable - full forward + backward, 55.88 GFLOP / pair.
--mode F: backbones frozen except their heads (reference README default, MAIN_CA:298-305), 19.10 GFLOP / pair.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel class, timed with HIP events on the launch stream inside
the timed region) and `cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multi-feature-vit_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FUS_MOD = ("model.crossvit_2vits_2additionaloutputs_changenormlayer_location_removeextralclayer_"
           "changemodelinputlocation_std002_sum")
GFLOP_PER_PAIR = {"T": 55.88, "F": 19.10}      # algorithmic work, each ViT counted once
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # MI355X_MICROARCH.md: dense MFMA peak per dtype
PEAK_HBM_GBS = 8000.0
NCLS = 10


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="pairs per GPU (BASELINE configs[2])")
    ap.add_argument("--mode", choices=["T", "F"], default="T")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--workload", choices=["ca", "single", "moco"], default="ca",
                    help="ca = BASELINE configs[2] (the metric's configuration); single = configs[1]; moco = configs[3] per-GPU slice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serialize-streams", action="store_true",
                    help="run the whole benchmark on ONE stream (no second encoder stream, no wgrad side stream): the mode whose "
                         "rocprofv3 kernel durations the roofline's per-kernel numbers are checked against")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def build_models(args, dev):
    import importlib
    import vits_returnftrs as vits
    fus = importlib.import_module(FUS_MOD)
    torch.manual_seed(0)
    backs = []
    for _ in range(2):
        m = vits.__dict__["vit_small"](precision=args.precision, img_size=args.img)      # MAIN_CA:289-290
        if args.mode == "F":                                                             # MAIN_CA:298-305
            for name, prm in m.named_parameters():
                if name not in ("head.weight", "head.bias"):
                    prm.requires_grad = False
        m.head = torch.nn.Linear(m.head.in_features, 3)                                  # MAIN_CA:309-310
        m.head.weight.data.normal_(mean=0.0, std=0.01)
        m.head.bias.data.zero_()
        backs.append(m.to(dev))
    model = fus.Fus_CrossViT(backs[0], backs[1]).to(dev)                                 # MAIN_CA:393
    return model, backs


def cpu_baseline(args, model, backs, x, xe, target):
    """The CPU oracle (oracle/ref_fusion.ca_step + backward) on this host's cores, bounded sample."""
    from oracle import ref_fusion
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, 16))          # a 1-GPU box owns a 16-core share of the host (more threads only oversubscribe)
    torch.set_num_threads(ncpu)
    nb = 4
    fp = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    vits_p = []
    for m in backs:
        train = args.mode == "T"
        vits_p.append({k: v.detach().cpu().clone().requires_grad_((train and k != "pos_embed") or k.startswith("head"))
                       for k, v in m.state_dict().items()})
    xc, xec, tc = x[:nb].cpu(), xe[:nb].cpu(), target[:nb].cpu()

    def step():
        out, preds, loss, _ = ref_fusion.ca_step(fp, vits_p[0], vits_p[1], xc, xec, tc)
        loss.backward()
        for d in (fp, vits_p[0], vits_p[1]):
            for v in d.values():
                v.grad = None
        return out.detach()

    t0 = time.perf_counter()
    out = step()  # warm-up (also the sample if the host is slow)
    warm = time.perf_counter() - t0
    print(f"[bench] cpu_baseline warm-up step: {warm:.2f} s on {ncpu} threads", file=sys.stderr, flush=True)
    n = 0 if warm > 15.0 else max(1, min(args.cpu_steps, int(20.0 / max(warm, 1e-3))))
    if n:
        t0 = time.perf_counter()
        for _ in range(n):
            out = step()
        dt = (time.perf_counter() - t0) / n
    else:
        dt = warm
    return dict(value=nb / dt, unit="images/sec", cores=ncpu, kind="port",
                sample=f"oracle (torch f32 CPU restatement) CA step mode {args.mode}: {nb} pairs x {max(n, 1)} step(s)"
                       f"{' after 1 warm-up' if n else ' (the warm-up itself, host too slow for more)'}, {ncpu} threads"), out


def other_workloads(args, world, rank, dev, lib):
    """configs[1] (single-stream fwd/bwd, B=64) and configs[3] (MoCo pretrain step, 128 samples / GPU): same timing contract,
    reported with their own metric names (they are NOT the headline metric)."""
    import types
    from functools import partial
    import vits
    from mfvit.ddp import GradSync
    from mfvit.losses import cross_entropy
    from mfvit.optim import SGD, AdamW
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234 + rank)
    sync = GradSync()
    if args.workload == "single":
        B = 64 if args.batch == 128 else args.batch
        model = vits.vit_small(num_classes=3, precision=args.precision, img_size=args.img).to(dev)     # MAIN_SS:276,290
        sync.attach(model)
        x = torch.randn(B, 3, args.img, args.img, generator=g).to(dev)
        y = torch.randint(0, 3, (B,), generator=g).to(dev)
        opt = SGD(model.parameters(), lr=1e-3, momentum=0.9)                                           # MAIN_SS:386-398

        def step():
            opt.zero_grad(set_to_none=True)
            loss, _ = cross_entropy(model(x), y)                                                       # MAIN_SS:711-714
            loss.backward()
            sync.reduce_grads(list(model.head.parameters()))
            sync.finish()
            opt.step()
            return loss
        gflop, name = 27.59, "images/sec (single-stream vit_small fwd+bwd step, BASELINE configs[1])"
    else:
        import moco.builder_vit_mocov3structure_mocov2loss as bld
        from mfvit.moco_ops import cross_entropy_rows
        from mfvit.schedules import adjust_moco_momentum
        B = args.batch
        model = bld.MoCo_ViT(partial(vits.vit_small, stop_grad_conv1=True, precision=args.precision, img_size=args.img),
                             types.SimpleNamespace(arch="vit_small"), 256, 4096, 0.2).to(dev)          # MAIN_MOCO:273-275, README.md:33
        sync.attach(model.base_encoder)
        x1 = torch.randn(B, 3, args.img, args.img, generator=g).to(dev)
        x2 = torch.randn(B, 3, args.img, args.img, generator=g).to(dev)
        train = [p for p in model.parameters() if p.requires_grad]
        small = [p for n, p in model.named_parameters() if p.requires_grad and (".head." in n or n.startswith("predictor"))]
        opt = AdamW(train, lr=1.5e-4, weight_decay=0.1)                                                # MAIN_MOCO:338-340
        it = [0]

        def step():
            m = adjust_moco_momentum(it[0] / 1000.0, 300, 0.99)                                        # MAIN_MOCO:525-526
            it[0] += 1
            logits, labels = model(x1, x2, m)                                                          # MAIN_MOCO:534
            loss = cross_entropy_rows(logits, labels)                                                  # MAIN_MOCO:535
            opt.zero_grad(set_to_none=True)
            loss.backward()                                                                            # MAIN_MOCO:546
            sync.reduce_grads(small)
            sync.finish()
            opt.step()
            return loss
        gflop, name = 36.80, "samples/sec (MoCo vit_small pretrain step, BASELINE configs[3] per-GPU slice of 128)"
    for _ in range(max(args.warmup, 1)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    if rank == 0:
        print(json.dumps(dict(metric=name, value=B * world * args.steps / dt, unit="images/sec", n_gpus=world, steps=args.steps,
                              warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype=args.precision, data="synthetic",
                              config=dict(workload=args.workload, per_gpu_batch=B, global_batch=B * world, parallelism=f"dp{world}"),
                              model_tflops=gflop * B * world * args.steps / dt / 1e3 if args.img == 224 else None,
                              loss=float(loss.detach()))), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ('nccl' on ROCm) in production; MFVIT_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals
        dist.init_process_group(os.environ.get("MFVIT_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from mfvit import _lib
    from mfvit.ddp import GradSync
    from mfvit.losses import cross_entropy
    lib = _lib.lib()

    if args.workload != "ca":
        return other_workloads(args, world, rank, dev, lib)
    model, backs = build_models(args, dev)
    g = torch.Generator().manual_seed(1234 + rank)                                       # SURVEY.md 8(d) synthetic inputs
    B = args.batch
    x = torch.randn(B, 3, args.img, args.img, generator=g).to(dev)
    xe = torch.randn(B, 3, args.img, args.img, generator=g).to(dev)
    target = torch.randint(0, 3, (B,), generator=g).to(dev)

    params = list(model.parameters())
    for m in backs:
        params += [p for p in m.parameters() if p.requires_grad]
    from mfvit.optim import Adam
    opt = Adam(params, lr=1e-4, betas=(0.9, 0.999))                                      # MAIN_CA:455-459 (multi-tensor HIP kernel)
    sync = GradSync()
    for m in backs:
        sync.attach(m)
    small = list(model.parameters()) + [p for m in backs for p in m.head.parameters()]

    def step():
        opt.zero_grad(set_to_none=True)
        fused, x_c, x_e = model(backs[0], backs[1], x, xe)                               # MAIN_CA:862
        output = fused + x_c + x_e                                                       # MAIN_CA:868
        loss, preds = cross_entropy(output, target)                                      # MAIN_CA:870-873
        loss.backward()                                                                  # MAIN_CA:880
        sync.reduce_grads(small)
        sync.finish()
        opt.step()                                                                       # MAIN_CA:882
        return loss, output

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.serialize_streams:
        model._two_streams = False
        lib.mfvit_set_wgrad_stream(0)

    # warm-up; the first warm-up steps time every kernel class to pick the dominant one
    buf = (ctypes.c_double * (NCLS * 4))()
    lib.mfvit_prof_enable((1 << NCLS) - 1)
    for i in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    lib.mfvit_prof_collect(buf, NCLS)
    cls_ms = [buf[c * 4 + 1] for c in range(NCLS)]
    dom = max(range(NCLS), key=lambda c: cls_ms[c])
    lib.mfvit_prof_enable(1 << dom)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, output = step()
    barrier()
    dt = time.perf_counter() - t0
    lib.mfvit_prof_collect(buf, NCLS)
    lib.mfvit_prof_enable(0)
    # attribution pass (untimed, after the timed region): same step with the two encoder streams and the wgrad side stream
    # serialised, so that every kernel's event-timed duration is its own rather than a share of a co-scheduled GPU
    solo = (ctypes.c_double * (NCLS * 4))()
    solo_ms_per_step = 0.0
    if True:   # every rank runs it (the step contains the gradient all-reduce); rank 0 reports
        model._two_streams = False
        lib.mfvit_set_wgrad_stream(0)
        step()
        torch.cuda.synchronize()
        lib.mfvit_prof_enable((1 << NCLS) - 1)
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        solo_ms_per_step = 1e3 * (time.perf_counter() - t1) / 3
        lib.mfvit_prof_collect(solo, NCLS)
        lib.mfvit_prof_enable(0)
        if not args.serialize_streams:
            lib.mfvit_set_wgrad_stream(1)
            model._two_streams = True
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)

    if rank == 0:
        # The dominant kernel's own duration comes from the serialized pass: HIP-event pairs around a launch on one of four
        # co-scheduled streams also count the time that stream waits for CUs (they read 2-3x the rocprofv3 kernel duration), so the
        # timed region's event numbers are reported beside it, not as the kernel's roofline.
        name = lib.mfvit_prof_class_name(dom).decode()

        def roof_of(launches, ms, flops, bts, per_step):
            r = dict(launches_per_step=launches / per_step, avg_us=1e3 * ms / max(launches, 1))
            if flops > 0:
                ach = flops / (ms * 1e-3) / 1e12
                r.update(bound="mfma", achieved=ach, peak=PEAK_TFLOPS[args.precision], unit="TFLOP/s", frac=ach / PEAK_TFLOPS[args.precision])
            else:
                ach = bts / (ms * 1e-3) / 1e9
                r.update(bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS)
            return r

        roof = dict(kernel=name)
        roof.update(roof_of(*(solo[dom * 4 + i] for i in range(4)), 3))
        roof["traffic"] = None
        roof["measured"] = ("HIP events around every launch of this kernel class in a 3-step pass with the streams serialized, right "
                            "after the timed region (same process, same tensors); agrees with the rocprofv3 kernel durations of "
                            "`bench.py --serialize-streams` (profiles/)")
        tr = roof_of(*(buf[dom * 4 + i] for i in range(4)), args.steps)
        tr["note"] = "same kernel class timed inside the timed region: event pairs on co-scheduled streams include waiting for CUs"
        roof["timed_region"] = tr
        try:   # HBM bytes per launch of this kernel class from the committed PMC passes (profiles/README.md); null if absent
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_by_class.json")))
            if args.batch == 128 and args.img == 224 and args.precision == "bf16" and name in tj["per_class"]:
                roof["traffic"] = tj["per_class"][name]["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/r01_hbm_traffic_by_class.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except (OSError, ValueError, KeyError):
            pass
        peak_t = PEAK_TFLOPS[args.precision]
        per = {}
        for c in range(NCLS):
            n, ms_c, fl, by = (solo[c * 4 + i] for i in range(4))
            if n > 0:
                e = dict(launches_per_step=n / 3, avg_us=round(1e3 * ms_c / n, 2), ms_per_step=round(ms_c / 3, 3))
                if fl > 0:
                    e.update(tflops=round(fl / (ms_c * 1e-3) / 1e12, 1), frac_of_mfma_peak=round(fl / (ms_c * 1e-3) / 1e12 / peak_t, 4))
                elif by > 0:
                    e.update(gbs=round(by / (ms_c * 1e-3) / 1e9, 1), frac_of_hbm_peak=round(by / (ms_c * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))
                per[lib.mfvit_prof_class_name(c).decode()] = e
        roof["serialized_pass"] = dict(note="3-step attribution pass after the timed region: one stream, no wgrad side stream",
                                       ms_per_step=round(solo_ms_per_step, 3), per_class=per)
        total = sum(cls_ms) or 1.0
        roof["warmup_time_share_by_class"] = {lib.mfvit_prof_class_name(c).decode(): round(cls_ms[c] / total, 4)
                                              for c in range(NCLS) if cls_ms[c] > 0}
        out = dict(metric="images/sec (two-stream 224^2 vit_small MF-CA train step)", value=B * world * args.steps / dt,
                   unit="images/sec", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps,
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.precision, data="synthetic",
                   config=dict(workload=f"BASELINE configs[2]: two-stream MF-ViT CA finetune step, {B} CXR+Enh pairs/GPU at "
                                        f"{args.img}x{args.img}, 2x vit_small + cross-attention fusion + CE + backward + Adam; "
                                        f"mode {args.mode} ({'full backward through both backbones' if args.mode == 'T' else 'frozen backbones (README default)'})",
                               global_batch=B * world, mode=args.mode, parallelism=f"dp{world}", streams="serialized" if args.serialize_streams else "two encoder streams + wgrad side streams",
                               algorithmic_gflop_per_pair=GFLOP_PER_PAIR[args.mode] if args.img == 224 else None),
                   model_tflops=(GFLOP_PER_PAIR[args.mode] * B * world * args.steps / dt / 1e3) if args.img == 224 else None,
                   loss=float(loss.detach()), roofline=roof)
        print("[bench] gpu " + json.dumps(out), file=sys.stderr, flush=True)
        if not args.no_cpu_baseline:
            cb, ref_out = cpu_baseline(args, model, backs, x, xe, target)
            out["cpu_baseline"] = cb
            with torch.no_grad():                       # same (final) weights as the oracle copy, same first pairs
                f_, xc_, xe_ = model(backs[0], backs[1], x, xe)
            got = (f_ + xc_ + xe_)[:ref_out.shape[0]].float().cpu()
            out["parity_vs_cpu_oracle"] = dict(
                logits_max_rel_err=float((got - ref_out).abs().max() / ref_out.abs().max()),
                argmax_equal=bool((got.argmax(1) == ref_out.argmax(1)).all()),
                note="bench-precision logits vs the f32 CPU oracle on the same first 4 pairs and the same final weights; the "
                     "1e-3 parity gate is asserted in precision='fp32' by tests/ (bf16 is the throughput mode)")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
