import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multi-feature-vit_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# The library reads its A/B switches (MFVIT_ROWP, MFVIT_TN2_*, MFVIT_ATTN_*, ...) once per process - unless MFVIT_AB_LIVE=1, which makes it
# read them at every launch: the kernel-variant tests flip them with monkeypatch inside ONE pytest process.  Must be set before the first
# launch of the process (the flag itself is read once).
os.environ.setdefault("MFVIT_AB_LIVE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's matmuls are small: on a 256-core host torch's default (one thread per core) oversubscribes them and the wall time
    # of the suite swings by 2 x between boxes (270 - 580 s measured); 32 threads are both faster and steadier
    torch.set_num_threads(min(32, os.cpu_count() or 1))


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def rng_tensor(seed, shape, scale=1.0, dtype=torch.float32):
    """Same recipe as oracle/make_golden.py::rng_tensor (numpy PCG64, float64 draw, cast)."""
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(size=shape, dtype=np.float64) * scale).to(dtype)


def check_sampled(npz, key, t, rtol=1e-9, atol=1e-10):
    """Compare a tensor with a strided-sample fixture entry written by make_golden.put()."""
    f = t.detach().double().flatten().cpu()
    idx = torch.from_numpy(npz[f"{key}.idx"])
    val = torch.from_numpy(npz[f"{key}.val"])
    torch.testing.assert_close(f[idx], val, rtol=rtol, atol=atol)
    s, a = float(npz[f"{key}.sum"]), float(npz[f"{key}.abssum"])
    assert abs(float(f.sum()) - s) <= rtol * 10 * a + atol * f.numel()
    assert abs(float(f.abs().sum()) - a) <= rtol * 10 * a + atol * f.numel()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
