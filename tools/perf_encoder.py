"""Quick timing of the encoder alone (development aid, not the contract bench): fwd and fwd+bwd at a given batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch  # noqa: E402
import vits  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
m = vits.vit_small(num_classes=3, precision=prec).to("cuda:0")
x = torch.randn(B, 3, 224, 224, device="cuda:0")
r = torch.randn(B, 197, 384, device="cuda:0")


def run(train):
    if train:
        f = m.features3D(x)
        (f * r).sum().backward()
        m.zero_grad(set_to_none=True)
    else:
        with torch.no_grad():
            m.features3D(x)


for train in (False, True):
    for _ in range(3):
        run(train)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(train)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gflop = 9.197 * B * (3 if train else 1)
    print(f"B={B} {prec} {'fwd+bwd' if train else 'fwd'}: {dt*1e3:.2f} ms  {B/dt:.0f} img/s  {gflop/dt/1e3:.1f} TFLOP/s", flush=True)
