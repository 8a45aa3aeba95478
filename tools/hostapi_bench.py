"""Latency of the drop-in sxt_* entry points with HOST buffers at config 2 (PCIe-inclusive)."""
import time, sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from blitzar_amd import api
n = 1 << 20
api.init(api.SXT_GPU_BACKEND, n)  # n built-in generators precomputed (resident Z = 1 addends)
rng = np.random.default_rng(0)
s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0f
g = api.get_generators(n, 0).view(np.uint8).reshape(n, 160)
for name, gens in (("caller generators (host)", g), ("built-in generators", None)):
    for _ in range(2): api.compute_pedersen_commitments(0, [(s, False)], generators=gens)
    t0 = time.perf_counter()
    for _ in range(5): out = api.compute_pedersen_commitments(0, [(s, False)], generators=gens)
    dt = (time.perf_counter() - t0) / 5
    print(name, round(dt * 1e3, 2), "ms per sxt_ call", out[0, :4])
