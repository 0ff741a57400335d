"""f-2 measurement: fused input-pipeline kernel on resident data vs the Pillow chain on one host core."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import numpy as np
import torch
from mfvit import input_pipeline as ip
from mfvit._lib import check, lib, ptr, stream
dev = torch.device("cuda:0")
B, H, W, S, C = 128, 320, 390, 256, 224
rng = np.random.Generator(np.random.PCG64(0))
imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(B)]
tf = ip.GpuTransform("CheXpert-v1.0-small", S, C, 10, True)
params = tf.sample_params(B, torch.Generator().manual_seed(0))
t0 = time.perf_counter(); out = tf(imgs, params); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"end-to-end call incl. host tables + H2D of {B*H*W*3/1e6:.1f} MB: {1e3*(t1-t0):.2f} ms (first call)")
t0 = time.perf_counter(); out = tf(imgs, params); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"end-to-end call, tables cached: {1e3*(t1-t0):.2f} ms = {B/(t1-t0):.0f} img/s")
# kernel alone on resident data
desc = np.zeros((B, 20), dtype=np.int64)
ksx, tx = ip.axis_table(W, S); ksy, ty = ip.axis_table(H, S)
for s, (f, a, i, j) in enumerate(params):
    mode, terms = ip.rotation_terms(a, S)
    desc[s] = [s * H * W * 3, H, W, 0, tx.size, ksx, ksy, int(f), mode, *terms, (i << 32) | j, W * 3, 0, 0, 0]
src = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs])).to(dev)
dsc = torch.from_numpy(desc).to(dev)
tab = torch.from_numpy(np.concatenate([tx.reshape(-1), ty.reshape(-1)])).to(dev)
o = torch.empty(B, 3, C, C, device=dev)
mean = (ctypes.c_float * 3)(*tf.mean); std = (ctypes.c_float * 3)(*tf.std)
def run():
    check(lib().mfvit_input_transform(ptr(src), ptr(dsc), ptr(tab), B, S, C, ctypes.cast(mean, ctypes.c_void_p), ctypes.cast(std, ctypes.c_void_p), ptr(o), stream()), "x")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
byt = B * 3 * C * C * 4 + B * H * W * 3
print(f"kernel: {us:.1f} us for {B} images = {B/us*1e6:.0f} img/s; algorithmic bytes {byt/1e6:.1f} MB -> {byt/us/1e3:.0f} GB/s ({byt/us/1e3/8000:.3f} of 8 TB/s)")
assert torch.equal(o, out)
try:
    from PIL import Image
    t0 = time.perf_counter()
    for im, (f, a, i, j) in list(zip(imgs, params))[:32]:
        x = Image.fromarray(im).resize((S, S), Image.BILINEAR)
        if f: x = x.transpose(Image.FLIP_LEFT_RIGHT)
        x = x.rotate(a, Image.NEAREST, expand=False, fillcolor=0).crop((j, i, j + C, i + C))
        t = (torch.from_numpy(np.asarray(x)).permute(2, 0, 1).float().div(255) - torch.tensor(tf.mean).view(3, 1, 1)) / torch.tensor(tf.std).view(3, 1, 1)
    dt = time.perf_counter() - t0
    print(f"Pillow chain on one host core (what a DataLoader worker runs per image): {32/dt:.0f} img/s")
except ImportError:
    pass
