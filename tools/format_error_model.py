"""Error model of GEMM operand formats (numpy, CPU): y = A @ W.T with A [M][K] activations-like, W [N][K] weights-like, against float64.
  x3      : split bf16, hi.hi + hi.lo + lo.hi (shipping `bf16x3`)
  f16     : plain fp16
  h16+2f8 : hi = fp16(x); correction products a8.wl8 + al8.w8 with a8 = e4m3(x), al8 = e4m3(x - hi), each with a power-of-two scale per block of 32 k values (MX style)
  h16+2f6 : the same with e2m3 (fp6) corrections
Reported: max |y - y64| / max |y64| (the scale-relative norm of the parity tests) and the rms ratio."""
import numpy as np

rng = np.random.default_rng(0)


def to_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).astype(np.float64)


def to_f16(x):
    return x.astype(np.float16).astype(np.float64)


def quant_block(x, mant_bits, emin, vmax):
    """round to a tiny float (1 sign, implicit 1, `mant_bits` mantissa bits, smallest normal 2^emin, largest magnitude vmax) with a shared power-of-two scale per 32 k values"""
    M, K = x.shape
    xb = x.reshape(M, K // 32, 32)
    amax = np.abs(xb).max(axis=2, keepdims=True)
    scale = np.where(amax > 0, 2.0 ** np.floor(np.log2(np.maximum(amax, 1e-300) / vmax) + 1), 1.0)   # smallest power of two with amax / scale <= vmax
    v = xb / scale
    e = np.floor(np.log2(np.maximum(np.abs(v), 1e-300)))
    e = np.maximum(e, emin)
    q = np.round(v / 2.0 ** (e - mant_bits)) * 2.0 ** (e - mant_bits)
    q = np.clip(q, -vmax, vmax)
    return (q * scale).reshape(M, K)


def e4m3(x):
    return quant_block(x, 3, -6, 448.0)


def e2m3(x):
    return quant_block(x, 3, 0, 7.5)


def report(name, y, y64):
    print(f"   {name:9s} max-rel {np.abs(y - y64).max() / np.abs(y64).max():9.2e}   rms-rel {np.sqrt(((y - y64) ** 2).mean() / (y64 ** 2).mean()):9.2e}")


for K, tag in ((384, "qkv / fc1 (K = 384)"), (1536, "fc2 (K = 1536)")):
    M, N = 512, 384
    A = rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, 1)) * 0.5)        # LayerNorm-output-like rows with a spread of row scales
    A[rng.random((M, K)) < 0.01] *= 8.0                                                 # a few outliers
    W = rng.standard_normal((N, K)) * 0.05
    y64 = A @ W.T
    print(tag)
    ah, wh = to_bf16(A), to_bf16(W)
    al, wl = to_bf16(A - ah), to_bf16(W - wh)
    report("x3", ah @ wh.T + ah @ wl.T + al @ wh.T, y64)
    report("f16", to_f16(A) @ to_f16(W).T, y64)
    ah, wh = to_f16(A), to_f16(W)
    for nm, q in (("h16+2f8", e4m3), ("h16+2f6", e2m3)):
        report(nm, ah @ wh.T + q(A) @ q(W - wh).T + q(A - ah) @ q(W).T, y64)
