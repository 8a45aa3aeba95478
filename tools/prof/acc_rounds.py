#!/usr/bin/env python3
"""k_accumulate's duration against the number of workgroups of its grid (config-2 shape, the row
count varied): does a grid that is not a multiple of the machine's 768 resident workgroups pay
for a whole extra round?  

    python tools/prof/acc_rounds.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from blitzar_amd import api  # noqa: E402
import baseline_workloads as wl  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    os.environ["BLITZAR_AMD_NUM_DEVICES"] = "1"
    lib = api.load()
    assert api.init(api.SXT_GPU_BACKEND, 0) == 0
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nmax = 3 << 19
    scalars = torch.from_numpy(wl.mt19937_scalars(1, nmax, 32, top_mask=0x0f, seed=0)[0]).to(dev)
    generators = torch.empty((nmax, 160), dtype=torch.uint8, device=dev)
    lib.bzamd_ristretto255_generators_device(ctypes.c_void_p(generators.data_ptr()), 0, nmax, stream)
    out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    calls = 40
    for wgs in (1536, 1600, 1792, 2048, 2176, 2240, 2304, 2368, 2560, 3072):
        n = wgs * 512  # 16 windows, 32 entries per lane, 256 lanes per workgroup
        desc = (api.sxt_sequence_descriptor * 1)()
        desc[0] = api.sxt_sequence_descriptor(32, n, scalars.data_ptr(), 0)

        def call():
            lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 1, desc,
                                 ctypes.c_void_p(generators.data_ptr()), stream)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        lib.bzamd_stage_timing_begin_masked(calls, 0x3f)
        for _ in range(calls):
            call()
        torch.cuda.synchronize()
        ms = (ctypes.c_double * 6)()
        lib.bzamd_stage_timing_collect(ms)
        acc = ms[3] / calls
        print(f"n = {n:8d} ({n / (1 << 20):.3f} x 2^20), {wgs} workgroups = {wgs / 768:.2f} rounds: "
              f"k_accumulate {acc:.4f} ms, {acc / n * 1e9 / 16:.1f} ps per addition", flush=True)


if __name__ == "__main__":
    main()
