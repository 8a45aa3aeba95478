#!/usr/bin/env python3
"""Digit-distribution sweep of the variable-base MSM (curve25519, 2^20 rows, device buffers):
uniform scalars against the skewed shapes real table columns have (constants, booleans, small
integers, mostly-zero columns).  Prints ms per call and the stage split.

    python tools/skew_bench.py [--log2n 20] [--steps 10]
"""
import argparse
import ctypes
import os
import sys

import torch  # first: the library binds to the HIP runtime torch loaded

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blitzar_amd import api  # noqa: E402

STAGES = ["prepare", "recode", "sort", "accumulate", "reduce", "combine"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    n = 1 << args.log2n
    lib = api.load()
    assert api.init(api.SXT_GPU_BACKEND, 0) == 0
    dev = torch.device("cuda", 0)
    sh = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    gens = torch.empty((n, 160), dtype=torch.uint8, device=dev)
    lib.bzamd_ristretto255_generators_device(ctypes.c_void_p(gens.data_ptr()), 0, n, sh)

    def rnd(nbytes):
        return torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device=dev, generator=g)

    cases = {}
    u = rnd(32)
    u[:, 31] &= 0x0f
    cases["uniform 252-bit"] = u
    c = u[:1].repeat(n, 1)
    cases["one constant 252-bit value"] = c.contiguous()
    cases["1-byte column of ones"] = torch.ones((n, 1), dtype=torch.uint8, device=dev)
    cases["1-byte booleans"] = (rnd(1) & 1).contiguous()
    cases["1-byte uniform"] = rnd(1)
    small = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
    small[:, :2] = rnd(2)
    small[:, 1] &= 0x03
    cases["8-byte integers < 1024"] = small
    sparse = rnd(32)
    sparse[:, 31] &= 0x0f
    mask = (torch.rand((n, 1), device=dev, generator=g) < 0.9)
    cases["252-bit, 90% zero rows"] = torch.where(mask, torch.zeros_like(sparse), sparse).contiguous()
    out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    for name, scalars in cases.items():
        nbytes = scalars.shape[1]
        desc = (api.sxt_sequence_descriptor * 1)()
        desc[0] = api.sxt_sequence_descriptor(nbytes, n, scalars.data_ptr(), 0)
        for _ in range(2):
            lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 1, desc,
                                 ctypes.c_void_p(gens.data_ptr()), sh)
        torch.cuda.synchronize()
        lib.bzamd_stage_timing_begin(args.steps)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(args.steps):
            lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 1, desc,
                                 ctypes.c_void_p(gens.data_ptr()), sh)
        ev1.record()
        torch.cuda.synchronize()
        ms = (ctypes.c_double * 6)()
        calls = lib.bzamd_stage_timing_collect(ms)
        stages = " ".join(f"{STAGES[i]} {ms[i] / calls:.3f}" for i in range(6))
        print(f"{name:30s} {ev0.elapsed_time(ev1) / args.steps:8.3f} ms   {stages}", flush=True)
    api.reset_for_testing()


if __name__ == "__main__":
    main()
