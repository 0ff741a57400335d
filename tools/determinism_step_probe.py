"""usage (GPU box): python tools/determinism_step_probe.py [batch] [mode T|F]: forward + loss + backward of the two-stream CA step three times on the same weights and
inputs (no optimizer step) - which outputs and parameter gradients are the same bits from run to run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-feature-vit_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv.pop(1)) if len(sys.argv) > 1 else 128
mode = sys.argv.pop(1) if len(sys.argv) > 1 else "T"
args = bench.parse()
args.batch = B
run = bench.CaRun(args, torch.device("cuda:0"), 0, "bf16x3", mode)
names = {}
for tag, mod in (("fusion", run.model), ("backbone0", run.backs[0]), ("backbone1", run.backs[1])):
    for n, p in mod.named_parameters():
        names[id(p)] = f"{tag}.{n}"


def once():
    run.opt.zero_grad(set_to_none=True)
    fused, x_c, x_e = run.model(run.backs[0], run.backs[1], run.x, run.xe)
    out = fused + x_c + x_e
    loss, _ = run._ce(out, run.target)
    loss.backward()
    torch.cuda.synchronize()
    g = {"logits": out.detach().clone(), "loss": loss.detach().clone()}
    for mod in (run.model, run.backs[0], run.backs[1]):
        for p in mod.parameters():
            if p.grad is not None:
                g[names[id(p)]] = p.grad.detach().clone()
    return g


once()
a, b, c = once(), once(), once()
diff = [n for n in a if not (torch.equal(a[n], b[n]) and torch.equal(a[n], c[n]))]
print(f"CA step, batch {B}, mode {mode}: {len(a) - len(diff)} of {len(a)} tensors bit-identical over three runs; differing: {diff}")
