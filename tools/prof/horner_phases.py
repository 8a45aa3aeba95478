#!/usr/bin/env python3
"""k_horner alone, by number of populated windows: lone calls of a short column (2^14 rows) whose
scalars use the low 4 .. 32 bytes, the `combine` span of the engine's stage clock.  k_horner starts its
chain at the highest populated window, so combine(bytes) = window fold + encoding + (windows - 1) x
(c doublings + 1 addition): the slope is the chain, the intercept the rest.  Timing only: the
"generators" are random field elements (the complete formulas run the same instructions on any
input), nothing is checked here -- parity is tests/.

    python tools/prof/horner_phases.py [--lib path]   (on the GPU box)
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blitzar_amd import api  # noqa: E402

STAGES = ["prepare_addends", "recode", "bucket_sort", "accumulate", "reduce", "combine"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=14)
    ap.add_argument("--full-width-only", action="store_true",
                    help="only the 32-byte case (for a rocprofv3 --kernel-trace --stats run: every "
                         "kernel of the trace then belongs to a lone full-width call)")
    args = ap.parse_args()
    lib = api.load()
    api.init(api.SXT_GPU_BACKEND)
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(5)
    lib.bzamd_slow_instruction_fetch.restype = ctypes.c_int
    print("slow_instruction_fetch:", lib.bzamd_slow_instruction_fetch())
    n = 1 << args.log2n
    out = {}
    for cid, name in ((0, "curve25519"), (1, "bls12-381"), (2, "bn254"), (3, "grumpkin")):
        stride, out_size = api.CURVE_LAYOUT[cid][0], api.CURVE_LAYOUT[cid][1]
        if cid == 0:
            g = api.get_generators(n)
            gens = torch.from_numpy(np.ascontiguousarray(g).view(np.uint8).reshape(n, -1)).to(dev)
        else:
            host = np.zeros((n, stride), dtype=np.uint8)
            words = host.view(np.uint64).reshape(n, -1)
            nl = (stride // 8 - 1) // 2 if stride % 16 else stride // 16
            words[:, 0:2] = rng.integers(1, 1 << 62, size=(n, 2), dtype=np.uint64)
            words[:, nl:nl + 2] = rng.integers(1, 1 << 62, size=(n, 2), dtype=np.uint64)
            gens = torch.from_numpy(host).to(dev)
        res = torch.zeros((1, out_size), dtype=torch.uint8, device=dev)
        row = {}
        for nbytes in ((32,) if args.full_width_only else (2, 4, 8, 16, 24, 32)):
            s = np.zeros((n, 32), dtype=np.uint8)
            s[:, :nbytes] = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
            if nbytes == 32:
                s[:, 31] &= 0x0f
            sc = torch.from_numpy(s).to(dev)
            desc = (api.sxt_sequence_descriptor * 1)()
            desc[0] = api.sxt_sequence_descriptor(32, n, sc.data_ptr(), 0)

            def call():
                lib.bzamd_msm_device(cid, ctypes.c_void_p(res.data_ptr()), 1, desc,
                                     ctypes.c_void_p(gens.data_ptr()), stream)
            for _ in range(2):
                call()
            torch.cuda.synchronize()
            best = None
            for _ in range(5):
                lib.bzamd_stage_timing_begin_masked(64, 0x3f)
                call()
                torch.cuda.synchronize()
                ms = (ctypes.c_double * 6)()
                lib.bzamd_stage_timing_collect(ms)
                if best is None or ms[5] < best[5]:
                    best = list(ms)
            row[nbytes] = {"combine_ms": round(best[5], 4), "reduce_ms": round(best[4], 4)}
        out[name] = row
        c = {b: row[b]["combine_ms"] for b in row}
        if args.full_width_only:
            print(f"{name:11s} lone call, 2^{args.log2n} rows: reduce {row[32]['reduce_ms']} ms, combine {c[32]} ms")
            continue
        per_window = (c[32] - c[8]) / 12.0
        print(f"{name:11s} combine ms by scalar bytes {c}  -> per window {per_window * 1e3:.1f} us, "
              f"rest {c[32] - 15 * per_window:.4f} ms")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
