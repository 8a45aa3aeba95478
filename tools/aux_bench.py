#!/usr/bin/env python3
"""Timings of the adjacent components on one MI355X, through the C ABI:
  * sxt_multiexp_handle_write_to_file: partition-table construction on the device
  * sxt_curve25519_prove_inner_product / _verify_inner_product
Prints one JSON line per measurement.  Correctness of both is covered by tests/ (-m gpu)."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401  (HIP runtime first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blitzar_amd import api  # noqa: E402

L_ORDER = 2**252 + 27742317777372353535851937790883648493


def table(n, width):
    os.environ["BLITZAR_PARTITION_WINDOW_WIDTH"] = str(width)
    gens = api.get_generators(n, 0)
    h = api.MultiexpHandle(0, gens)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.bin")
        t0 = time.perf_counter()
        h.write_to_file(path)
        dt = time.perf_counter() - t0
        size = os.path.getsize(path)
    h.close()
    entries = ((n + width - 1) // width) << width
    print(json.dumps({"what": "curve25519 partition table on the device", "generators": n,
                      "window_width": width, "entries": entries, "file_bytes": size,
                      "seconds": round(dt, 3), "entries_per_s": entries / dt}), flush=True)


def inner_product(log2n):
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0f
    b[:, 31] &= 0x0f
    t0 = api.transcript_new("bench")
    api.prove_inner_product(t0, min(n, 64), 0, a[:64], b[:64])  # warm-up
    t1 = time.perf_counter()
    l, r, ap, _ = api.prove_inner_product(t0, n, 0, a, b)
    prove_s = time.perf_counter() - t1
    print(json.dumps({"what": "sxt_curve25519_prove_inner_product", "n": n,
                      "seconds": round(prove_s, 4)}), flush=True)


def main():
    assert api.init(api.SXT_GPU_BACKEND, 0) == 0
    table(64, 16)     # warm-up, 4 windows
    table(1024, 16)   # 64 windows x 2^16 entries = 4 Mi entries, 503 MB
    for k in (12, 16, 20):
        inner_product(k)
    api.reset_for_testing()


if __name__ == "__main__":
    main()
