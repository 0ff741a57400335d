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
# MI355X_MICROARCH.md: dense MFMA peak per dtype.  bf16x3 (split bf16) issues three bf16 MFMAs per algorithmic product, so the ceiling
# for ALGORITHMIC flops in that mode is a third of the bf16 peak.
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "bf16x3": 2500.0 / 3.0, "fp32": 157.3}
PEAK_NOTE = {"bf16": "dense bf16 MFMA", "fp16": "dense fp16 MFMA", "fp32": "dense f32 MFMA",
             "bf16x3": "dense bf16 MFMA / 3 (split bf16: three MFMAs per algorithmic product)"}
ARITH = {"bf16": "bf16", "fp16": "f16", "fp32": "f32", "bf16x3": "bf16x3 (split bf16 hi+lo operands, 3 MFMAs per product, f32 accumulate)"}
PEAK_HBM_GBS = 8000.0
NCLS = 10
TRAFFIC_FILE = "r06_hbm_traffic_by_class.json"      # newest PMC traffic summary (tools/profile_bench.sh), used only when its source hash matches the tree


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="pairs per GPU (BASELINE configs[2])")
    ap.add_argument("--mode", choices=["T", "F"], default="T")
    ap.add_argument("--precision", choices=["bf16x3", "bf16", "fp16", "fp32"], default="bf16x3",
                    help="bf16x3 (default): split-bf16 products, the mode that meets the 1e-3 logits gate on the bf16 matrix core "
                         "(the reference's CA finetune is fp32); bf16: throughput mode; fp16: the reference's autocast arithmetic "
                         "(MoCo pretraining, with GradScaler); fp32: exact f32 MFMA")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (bf16 throughput mode, mode F) of the N=1 run")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--workload", choices=["ca", "single", "moco"], default="ca",
                    help="ca = BASELINE configs[2] (the metric's configuration); single = configs[1]; moco = configs[3] per-GPU slice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serialize-streams", action="store_true",
                    help="run the whole benchmark on ONE stream (no second encoder stream, no wgrad side stream): the mode whose "
                         "rocprofv3 kernel durations the roofline's per-kernel numbers are checked against")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--arch", choices=["vit_small", "vit_base"], default="vit_small",
                    help="backbone of the `single` workload (MAIN_MOCO:50 `-a`): vit_base runs the unfused encoder path (tile GEMMs + LayerNorm row passes)")
    return ap.parse_args()


def build_models(args, dev, precision=None, mode=None):
    import importlib
    import vits_returnftrs as vits
    fus = importlib.import_module(FUS_MOD)
    precision = precision or args.precision
    mode = mode or args.mode
    torch.manual_seed(0)
    backs = []
    for _ in range(2):
        m = vits.__dict__["vit_small"](precision=precision, img_size=args.img)           # MAIN_CA:289-290
        if mode == "F":                                                                  # MAIN_CA:298-305
            for name, prm in m.named_parameters():
                if name not in ("head.weight", "head.bias"):
                    prm.requires_grad = False
        m.head = torch.nn.Linear(m.head.in_features, 3)                                  # MAIN_CA:309-310
        m.head.weight.data.normal_(mean=0.0, std=0.01)
        m.head.bias.data.zero_()
        backs.append(m.to(dev))
    model = fus.Fus_CrossViT(backs[0], backs[1]).to(dev)                                 # MAIN_CA:393
    return model, backs


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _timed(fn, warmups, want, budget_s):
    """`warmups` untimed calls, then up to `want` timed ones (at least 1; fewer only when the host is too slow for the time budget)."""
    t0 = time.perf_counter()
    out = None
    for _ in range(warmups):
        out = fn()
    warm = (time.perf_counter() - t0) / max(warmups, 1)
    n = max(1, min(want, int(budget_s / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    return (time.perf_counter() - t0) / n, n, out


def cpu_baseline(args, model, backs, x, xe, target):
    """The CPU oracle on this host's cores, bounded samples: (1) the metric's own workload - the two-stream CA train step
    (oracle/ref_fusion.ca_step + backward) on 4 pairs; (2) BASELINE configs[0] - single-stream vit_small forward + CE on 4 images
    (MAIN_SS:711-714).  3 warm-up + up to --cpu-steps timed iterations each."""
    from oracle import ref_fusion, ref_vit
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

    def cfg1():
        with torch.no_grad():
            logits = ref_vit.head_linear(vits_p[0], ref_vit.features3d(vits_p[0], xc)[:, 0])
            return torch.nn.functional.cross_entropy(logits, tc)

    dt, n, out = _timed(step, 3, args.cpu_steps, 20.0)
    dt1, n1, _ = _timed(cfg1, 3, args.cpu_steps, 8.0)
    print(f"[bench] cpu_baseline: CA step {dt:.3f} s x {n}, cfg1 fwd+CE {dt1:.3f} s x {n1} on {ncpu} threads", file=sys.stderr, flush=True)
    return dict(value=nb / dt, unit="images/sec", cores=ncpu, kind="port", cpu_model=cpu_model_name(),
                sample=f"oracle (torch f32 CPU restatement) two-stream CA train step mode {args.mode}: {nb} pairs x {n} timed steps after 3 warm-up, "
                       f"{ncpu} threads",
                cfg1=dict(value=nb / dt1, unit="images/sec",
                          sample=f"BASELINE configs[0]: single-stream vit_small forward + CE, {nb} x 3 x {args.img} x {args.img}, {n1} timed "
                                 f"iterations after 3 warm-up, {ncpu} threads")), out


def attribution_pass(step, lib, precision, nsteps=3):
    """The per-class table of a workload: `nsteps` steps with every kernel class timed by HIP events on its launch stream (mfvit_prof_*; the caller
    has serialised whatever streams the workload uses).  Returns (ms per step, {class: launches, avg us, ms per step, TFLOP/s or GB/s, fraction})."""
    solo = (ctypes.c_double * (NCLS * 4))()
    step()
    torch.cuda.synchronize()
    lib.mfvit_prof_enable((1 << NCLS) - 1)
    t1 = time.perf_counter()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t1) / nsteps
    lib.mfvit_prof_collect(solo, NCLS)
    lib.mfvit_prof_enable(0)
    peak_t, per = PEAK_TFLOPS[precision], {}
    for c in range(NCLS):
        n, ms_c, fl, by = (solo[c * 4 + i] for i in range(4))
        if n > 0:
            e = dict(launches_per_step=n / nsteps, avg_us=round(1e3 * ms_c / n, 2), ms_per_step=round(ms_c / nsteps, 3))
            if fl > 0:
                e.update(tflops=round(fl / (ms_c * 1e-3) / 1e12, 1), frac_of_mfma_peak=round(fl / (ms_c * 1e-3) / 1e12 / peak_t, 4))
            elif by > 0:
                e.update(gbs=round(by / (ms_c * 1e-3) / 1e9, 1), frac_of_hbm_peak=round(by / (ms_c * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))
            per[lib.mfvit_prof_class_name(c).decode()] = e
    return ms, per


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
        model = vits.__dict__[args.arch](num_classes=3, precision=args.precision, img_size=args.img).to(dev)     # MAIN_SS:276,290
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
        if args.arch == "vit_base":        # 3 x (12 blocks x 2 x 197 x (4 + 8) x 768^2 + attention 4 x 197^2 x 768 per block + patch embedding) flops per image
            gflop, name = 3 * (12 * (2 * 197 * 12 * 768 * 768 + 4 * 197 * 197 * 768) + 2 * 196 * 768 * 768) / 1e9, "images/sec (single-stream vit_base fwd+bwd step)"
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

        from mfvit.amp import GradScaler
        scaler = GradScaler(enabled=args.precision == "fp16")                                          # MAIN_MOCO:349 (fp16 autocast path)

        def step():
            m = adjust_moco_momentum(it[0] / 1000.0, 300, 0.99)                                        # MAIN_MOCO:525-526
            it[0] += 1
            logits, labels = model(x1, x2, m)                                                          # MAIN_MOCO:534
            loss = cross_entropy_rows(logits, labels)                                                  # MAIN_MOCO:535
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()                                                              # MAIN_MOCO:546
            sync.reduce_grads(small)
            sync.finish()
            scaler.step(opt)                                                                           # MAIN_MOCO:547
            scaler.update()                                                                            # MAIN_MOCO:548
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
    ser_ms, per = attribution_pass(step, lib, args.precision)     # one stream by construction in these workloads
    if rank == 0:
        print(json.dumps(dict(metric=name, value=B * world * args.steps / dt, unit="images/sec", n_gpus=world, steps=args.steps,
                              warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype=ARITH[args.precision], data="synthetic",
                              config=dict(workload=args.workload, precision=args.precision, per_gpu_batch=B, global_batch=B * world, parallelism=f"dp{world}"),
                              model_tflops=gflop * B * world * args.steps / dt / 1e3 if args.img == 224 else None,
                              loss=float(loss.detach()),
                              serialized_pass=dict(note="3-step attribution pass after the timed region (HIP events per kernel class, mfvit_prof_*)",
                                                   ms_per_step=round(ser_ms, 3), peak_tflops=PEAK_TFLOPS[args.precision], per_class=per))), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher (what MAIN_MOCO:207 does with mp.spawn(main_worker, nprocs=ngpus)): start N ranks
    as a CHILD process group through torch.distributed.run, relay rank 0's JSON line and fail unless it reports n_gpus == N.  The
    parent never touches the GPU (no HIP call before or after the child) and never exec()s."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"[bench] the {args.gpus}-rank child failed with exit code {proc.returncode}", file=sys.stderr)
        sys.exit(proc.returncode or 1)
    if line is None:
        print("[bench] the ranks printed no result line", file=sys.stderr)
        sys.exit(1)
    if json.loads(line).get("n_gpus") != args.gpus:
        print(f"[bench] asked for --gpus {args.gpus} but the ranks report n_gpus={json.loads(line).get('n_gpus')}", file=sys.stderr)
        sys.exit(1)
    print(line, flush=True)


class CaRun:
    """One two-stream CA workload instance: models, optimizer, synthetic batch, step()."""

    def __init__(self, args, dev, rank, precision, mode):
        from mfvit.ddp import GradSync
        from mfvit.losses import cross_entropy
        from mfvit.optim import Adam
        self.args, self.precision, self.mode = args, precision, mode
        self.model, self.backs = build_models(args, dev, precision, mode)
        g = torch.Generator().manual_seed(1234 + rank)                                   # SURVEY.md 8(d) synthetic inputs
        B = args.batch
        self.x = torch.randn(B, 3, args.img, args.img, generator=g).to(dev)
        self.xe = torch.randn(B, 3, args.img, args.img, generator=g).to(dev)
        self.target = torch.randint(0, 3, (B,), generator=g).to(dev)
        # one param group per gradient exchange: the fusion arena + classifier heads (flat synchronous exchange after backward), then
        # each encoder arena (asynchronous per-bucket exchange from its backward hooks) - the optimizer joins an encoder's buckets just
        # before that group's kernel is queued, so the update of what has arrived overlaps what is still on the links
        self.small = list(self.model.parameters()) + [p for m in self.backs for p in m.head.parameters()]
        heads = {id(p) for m in self.backs for p in m.head.parameters()}
        groups, owners = [{"params": self.small}], [None]
        for m in self.backs:
            ps = [p for p in m.parameters() if p.requires_grad and id(p) not in heads]
            if ps:                                                                       # mode F: frozen backbones have no group
                groups.append({"params": ps})
                owners.append(m)
        self.opt = Adam(groups, lr=1e-4, betas=(0.9, 0.999))                             # MAIN_CA:455-459 (multi-tensor HIP kernel)
        self.sync = GradSync()
        for m in self.backs:
            self.sync.attach(m)
        self.opt.before_group = lambda gi: self.sync.finish(owners[gi]) if owners[gi] is not None else None
        self._ce = cross_entropy

    def step(self):
        self.opt.zero_grad(set_to_none=True)
        fused, x_c, x_e = self.model(self.backs[0], self.backs[1], self.x, self.xe)      # MAIN_CA:862
        output = fused + x_c + x_e                                                       # MAIN_CA:868
        loss, preds = self._ce(output, self.target)                                      # MAIN_CA:870-873
        loss.backward()                                                                  # MAIN_CA:880
        self.sync.reduce_grads(self.small)
        self.opt.step()                                                                  # MAIN_CA:882 (joins each encoder's exchange per group)
        assert not self.sync.pending()
        # detached: a loss that outlives the step keeps the autograd graph - and the parameters' AccumulateGrad nodes, with the stream
        # they were created under - alive into the next forward (torch then warns when a later pass runs an encoder on another stream)
        return loss.detach(), output.detach()

    def logits(self, n, train_path=True):
        """Logits of the first n pairs on the current weights.  train_path (default): the NEED-GRAD forward, i.e. the very kernels the timed
        step runs (activations saved); train_path=False: the no-grad forward (same kernels, no activation copies kept), reported beside it."""
        if train_path:
            f_, xc_, xe_ = self.model(self.backs[0], self.backs[1], self.x, self.xe)
        else:
            with torch.no_grad():
                f_, xc_, xe_ = self.model(self.backs[0], self.backs[1], self.x, self.xe)
        return (f_ + xc_ + xe_).detach()[:n].float().cpu()


def timed_region(run, steps, warmup, world, dev):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; max over ranks."""
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    step = run.step
    for _ in range(max(warmup, 1)):
        run.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = step()
    t_host = time.perf_counter() - t0                       # the host has queued every step (stderr only: how far it runs ahead of the GPU)
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench] host queued {steps} steps in {1e3 * t_host:.1f} ms of the {1e3 * dt:.1f} ms they took", file=sys.stderr)
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    timed_region.rank_seconds = [dt]
    if world > 1:
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                           # per-rank times: the first multi-GPU run diagnoses its own stragglers
        timed_region.rank_seconds = [float(x) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t), loss


def parity_vs_oracle(run, ref_out):
    got = run.logits(ref_out.shape[0])
    nog = run.logits(ref_out.shape[0], train_path=False)
    floor = 0.05 * ref_out.abs().max()
    return dict(logits_max_rel_err=float((got - ref_out).abs().max() / ref_out.abs().max()),
                logits_elementwise_rel_err=float(((got - ref_out).abs() / ref_out.abs().clamp_min(floor)).max()),
                elementwise_floor="|ref| floored at 0.05 x max|ref|",
                argmax_equal=bool((got.argmax(1) == ref_out.argmax(1)).all()),
                path="need-grad forward: the kernels of the timed step",
                no_grad_forward_max_rel_err=float((nog - ref_out).abs().max() / ref_out.abs().max()))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} does not match WORLD_SIZE {world}: refusing to report a {world}-rank number as "
                  f"{args.gpus} GPUs", file=sys.stderr)
        sys.exit(2)
    local = local % max(torch.cuda.device_count(), 1)      # (device_count() does not initialise the GPU)
    torch.cuda.set_device(local)                           # bind the rank to its GPU BEFORE the communicator is created
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ('nccl' on ROCm) in production; MFVIT_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals
        dist.init_process_group(os.environ.get("MFVIT_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    from mfvit import _lib
    lib = _lib.lib()

    if args.workload != "ca":
        return other_workloads(args, world, rank, dev, lib)
    run = CaRun(args, dev, rank, args.precision, args.mode)
    B = args.batch
    if args.serialize_streams:
        run.model._two_streams = False
        lib.mfvit_set_wgrad_stream(0)

    run.sync.timing = world > 1
    dt, loss = timed_region(run, args.steps, args.warmup, world, dev)
    rank_s = list(timed_region.rank_seconds)
    wait_ms = run.sync.finish_wait_ms() if world > 1 else 0.0
    run.sync.timing = False
    wait_all = [wait_ms]
    if world > 1:
        w = torch.tensor([wait_ms], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(every, w)
        wait_all = [float(x) for x in every]

    # attribution pass (untimed, after the timed region): the same step with the two encoder streams and the wgrad side stream
    # serialised, every kernel class timed with HIP events on its launch stream by the library (mfvit_prof_*): each kernel's
    # duration is then its own rather than a share of a co-scheduled GPU.  The roofline kernel is the class with the largest time
    # share IN THIS PASS.  Every rank runs it (the step contains the gradient all-reduce); rank 0 reports.
    solo = (ctypes.c_double * (NCLS * 4))()
    run.model._two_streams = False
    lib.mfvit_set_wgrad_stream(0)
    run.step()
    torch.cuda.synchronize()
    lib.mfvit_prof_enable((1 << NCLS) - 1)
    t1 = time.perf_counter()
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()
    solo_ms_per_step = 1e3 * (time.perf_counter() - t1) / 3
    lib.mfvit_prof_collect(solo, NCLS)
    tags = (ctypes.c_double * 9)()
    lib.mfvit_prof_collect_tags(tags, 3)
    lib.mfvit_prof_enable(0)
    if not args.serialize_streams:
        lib.mfvit_set_wgrad_stream(1)
        run.model._two_streams = True

    if rank == 0:
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
        dom = max(range(NCLS), key=lambda c: solo[c * 4 + 1])
        name = lib.mfvit_prof_class_name(dom).decode()
        launches, ms, flops, bts = (solo[dom * 4 + i] for i in range(4))
        roof = dict(kernel=name, launches_per_step=launches / 3, avg_us=1e3 * ms / max(launches, 1))
        if flops > 0:
            ach = flops / (ms * 1e-3) / 1e12
            roof.update(bound="mfma", achieved=ach, peak=peak_t, unit="TFLOP/s", frac=ach / peak_t, peak_note=PEAK_NOTE[args.precision])
        else:
            ach = bts / (ms * 1e-3) / 1e9
            roof.update(bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS)
        roof["traffic"] = None
        roof["measured"] = ("HIP events (recorded by the library on the launch stream) around every launch of this kernel class in a 3-step "
                            "pass with the streams serialized, right after the timed region (same process, same tensors); the class is the "
                            "one with the largest time share in that pass; rocprofv3 kernel durations of `bench.py --serialize-streams` "
                            "are under profiles/")
        # HBM bytes per launch from the committed PMC passes: only when they were measured on THESE kernel sources at this precision
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
            if (tj.get("source_hash") == _lib.source_hash() and tj.get("precision") == args.precision and args.batch == 128
                    and args.img == 224 and name in tj["per_class"]):
                roof["traffic"] = tj["per_class"][name]["hbm_bytes_per_launch"]
                roof["traffic_source"] = (f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                          f"kernel sources {tj['source_hash'][:12]})")
            else:
                roof["traffic_note"] = "committed PMC traffic was measured on other kernel sources / another precision: not reported"
        except (OSError, ValueError, KeyError):
            pass
        roof["serialized_pass"] = dict(note="3-step attribution pass after the timed region: one stream, no wgrad side stream",
                                       ms_per_step=round(solo_ms_per_step, 3), per_class=per)
        # The fused multi-head self-attention figure of BASELINE.json's metric / SURVEY.md 8d (Attention.forward, model/module.py:52-64: qkv projection +
        # softmax(q k^T) v + output projection, 292.0 MFLOP per image and layer at 224^2): here three launches - the qkv tile GEMM, the attention core, the
        # output projection with its residual + LayerNorm epilogue - so the figure is their summed algorithmic FLOPs over their summed time, beside the
        # core alone.  MFMA-busy % of the core from the PMC passes: profiles/pmc/.
        if tags[1 * 3] > 0 and tags[2 * 3] > 0 and solo[4 * 4] > 0:
            fl = tags[1 * 3 + 2] + tags[2 * 3 + 2] + solo[4 * 4 + 2]
            ms_f = tags[1 * 3 + 1] + tags[2 * 3 + 1] + solo[4 * 4 + 1]
            roof["fused_mhsa"] = dict(
                note="forward MHSA = qkv projection + attention core + output projection (+ residual + LayerNorm), three launches: summed algorithmic "
                     "FLOPs / summed time of the serialized pass",
                parts_avg_us=dict(qkv=round(1e3 * tags[4] / tags[3], 2), core=round(1e3 * solo[4 * 4 + 1] / solo[4 * 4], 2),
                                  proj_res_ln=round(1e3 * tags[7] / tags[6], 2)),
                tflops=round(fl / (ms_f * 1e-3) / 1e12, 1), frac_of_mfma_peak=round(fl / (ms_f * 1e-3) / 1e12 / peak_t, 4),
                core_only=dict(tflops=round(solo[4 * 4 + 2] / (solo[4 * 4 + 1] * 1e-3) / 1e12, 1),
                               frac_of_mfma_peak=round(solo[4 * 4 + 2] / (solo[4 * 4 + 1] * 1e-3) / 1e12 / peak_t, 4)),
                target="BASELINE.json: >= 60 % MFMA on the fused-attention kernel - not met; the ceiling at head_dim 32 is stated in DESIGN.md 5")
        out = dict(metric="images/sec (two-stream 224^2 vit_small MF-CA train step)", value=B * world * args.steps / dt,
                   unit="images/sec", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps,
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=ARITH[args.precision], data="synthetic",
                   config=dict(workload=f"BASELINE configs[2]: two-stream MF-ViT CA finetune step, {B} CXR+Enh pairs/GPU at "
                                        f"{args.img}x{args.img}, 2x vit_small + cross-attention fusion + CE + backward + Adam; "
                                        f"mode {args.mode} ({'full backward through both backbones' if args.mode == 'T' else 'frozen backbones (README default)'})",
                               global_batch=B * world, mode=args.mode, precision=args.precision, parallelism=f"dp{world}",
                               streams="serialized" if args.serialize_streams else "two encoder streams (weight gradients on the encoder's own stream)",
                               algorithmic_gflop_per_pair=GFLOP_PER_PAIR[args.mode] if args.img == 224 else None),
                   model_tflops=(GFLOP_PER_PAIR[args.mode] * B * world * args.steps / dt / 1e3) if args.img == 224 else None,
                   loss=float(loss.detach()), roofline=roof)
        if world > 1:
            # self-diagnosis of the data-parallel run: per-rank time of the timed region (stragglers) and how long each rank's compute stream
            # sat inside GradSync.finish() - the part of the gradient all-reduce that the backward did not hide (includes the warm-up steps'
            # share only if they overlapped the timed region: the events are reset by finish_wait_ms())
            n_finish = max(args.steps + max(args.warmup, 1), 1)
            out["data_parallel"] = dict(rank_ms_per_step=[round(1e3 * t / args.steps, 3) for t in rank_s],
                                        rank_spread_pct=round(100.0 * (max(rank_s) - min(rank_s)) / max(rank_s), 2),
                                        grad_sync_wait_ms_per_step=[round(w / n_finish, 3) for w in wait_all],
                                        bucket_layers=int(os.environ.get("MFVIT_GRAD_BUCKET_LAYERS", "4")),
                                        bucket_dtype=os.environ.get("MFVIT_GRAD_BUCKET_DTYPE", "f32"),
                                        note="grad_sync_wait = time the compute stream spent waiting in GradSync.finish() per step (events on the "
                                             "stream, warm-up steps included in the average): all-reduce time NOT hidden behind the backward")
        print("[bench] gpu " + json.dumps(out), file=sys.stderr, flush=True)
        ref_out = None
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is an N = 1 measurement (the other ranks would only wait for it)
            cb, ref_out = cpu_baseline(args, run.model, run.backs, run.x, run.xe, run.target)
            out["cpu_baseline"] = cb
            par = parity_vs_oracle(run, ref_out)      # same (final) weights as the oracle copy, same first pairs
            par["note"] = (f"precision '{args.precision}' logits vs the f32 CPU oracle on the same first {ref_out.shape[0]} pairs and the same final "
                           "weights (gate: 1e-3 relative, argmax equal)")
            out["parity_vs_cpu_oracle"] = par
        if world == 1 and not args.no_extras and not args.serialize_streams:
            # secondary lines, same timing contract with fewer steps: the bf16 throughput mode and the reference README's frozen-
            # backbone mode F.  They are NOT the headline (`value`).
            also = {}
            final_sd = ({k: v.detach().clone() for k, v in run.model.state_dict().items()},
                        [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in run.backs])
            del run
            torch.cuda.empty_cache()
            for tag, prec, mode in (("bf16_mode_" + args.mode, "bf16", args.mode), (args.precision + "_mode_F", args.precision, "F")):
                if (prec, mode) == (args.precision, args.mode):
                    continue
                r2 = CaRun(args, dev, rank, prec, mode)
                dt2, loss2 = timed_region(r2, 10, 3, 1, dev)
                e = dict(value=B * 10 / dt2, unit="images/sec", ms_per_step=1e3 * dt2 / 10, steps=10, warmup=3, precision=prec, mode=mode,
                         model_tflops=(GFLOP_PER_PAIR[mode] * B * 10 / dt2 / 1e3) if args.img == 224 else None)
                if ref_out is not None and mode == args.mode:
                    r2.model.load_state_dict(final_sd[0])            # the weights the oracle output was computed on
                    for m, sd in zip(r2.backs, final_sd[1]):
                        m.load_state_dict(sd)
                    e["parity_vs_cpu_oracle"] = parity_vs_oracle(r2, ref_out)
                also[tag] = e
                del r2
                torch.cuda.empty_cache()
            out["also"] = also
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
