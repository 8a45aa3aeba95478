"""Latency of the drop-in sxt_* entry points with HOST buffers (PCIe-inclusive), warm."""
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import torch  # noqa: F401,E402  (HIP runtime load order, see tests/conftest.py)
from blitzar_amd import api  # noqa: E402

n = 1 << 20
api.init(api.SXT_GPU_BACKEND, n)  # n built-in generators precomputed (resident Z = 1 addends)
rng = np.random.default_rng(0)
g = api.get_generators(n, 0).view(np.uint8).reshape(n, 160)
for cols_n in (1, 10):
    cols = []
    for _ in range(cols_n):
        s = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        s[:, 31] &= 0x0f
        cols.append((s, False))
    for name, gens in (("caller generators (host)", g), ("built-in generators", None)):
        for _ in range(2):
            api.compute_pedersen_commitments(0, cols, generators=gens)
        t0 = time.perf_counter()
        for _ in range(5):
            out = api.compute_pedersen_commitments(0, cols, generators=gens)
        dt = (time.perf_counter() - t0) / 5
        print(f"{cols_n} x 2^20 x 32 B, {name}: {dt * 1e3:.2f} ms per sxt_ call, "
              f"{cols_n * n / dt:.3e} exponentiations/s", out[0, :2])
