#!/usr/bin/env python3
"""Step time of config 2 call by call over long sequences in throughput mode, with idle gaps of
several lengths in front: where the device's clock management puts a short timed region
(bench.py's leg order).  Prints one line per scenario: the mean ms per call of
consecutive blocks of `--block` calls (an event after every call on the caller's stream).

    python tools/prof/clock_course.py [--calls 600] [--block 25]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from blitzar_amd import api  # noqa: E402
import baseline_workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=600)
    ap.add_argument("--block", type=int, default=25)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    os.environ["BLITZAR_AMD_NUM_DEVICES"] = "1"
    lib = api.load()
    assert api.init(api.SXT_GPU_BACKEND, 0) == 0
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 1 << 20
    scalars = torch.from_numpy(wl.mt19937_scalars(1, n, 32, top_mask=0x0f, seed=0)[0]).to(dev)
    generators = torch.empty((n, 160), dtype=torch.uint8, device=dev)
    lib.bzamd_ristretto255_generators_device(ctypes.c_void_p(generators.data_ptr()), 0, n, stream)
    out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    desc = (api.sxt_sequence_descriptor * 1)()
    desc[0] = api.sxt_sequence_descriptor(32, n, scalars.data_ptr(), 0)
    torch.cuda.synchronize()

    def sequence(calls):
        events = [torch.cuda.Event(enable_timing=True) for _ in range(calls + 1)]
        events[0].record()
        for k in range(calls):
            lib.bzamd_pipeline_next()
            lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 1, desc,
                                 ctypes.c_void_p(generators.data_ptr()), stream)
            events[k + 1].record()
        lib.bzamd_pipeline_flush(stream)
        torch.cuda.synchronize()
        blocks = []
        for b in range(0, calls, args.block):
            e = min(b + args.block, calls)
            blocks.append(events[b].elapsed_time(events[e]) / (e - b))
        return blocks

    sequence(10)  # workspace, streams
    for gap in (2.0, 0.0, 0.02, 0.1, 0.5):
        time.sleep(gap)
        blocks = sequence(args.calls)
        print(f"idle {gap:4.2f} s, then {args.calls} calls: " +
              " ".join(f"{b:.3f}" for b in blocks), flush=True)
    # the bench's shape: ~55 calls, a flush, 5 calls, a flush, then 20 calls
    for gap in (2.0, 0.0):
        time.sleep(gap)
        a = sequence(55)
        b = sequence(5)
        c = sequence(20)
        print(f"idle {gap:4.2f} s, 55 + 5 + 20 calls with flushes between: "
              f"{sum(a) / len(a):.3f} | {b[0]:.3f} | {c[0]:.3f}", flush=True)


if __name__ == "__main__":
    main()
