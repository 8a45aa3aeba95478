"""The front of call k + 1 beside the accumulation of call k (BLITZAR_AMD_OVERLAP_FRONT=1) in a
PyTorch process, on a torch side stream (torch then creates its pool of 32 streams, which share the
hardware queues of the library's internal streams) against a HIP stream created through ctypes (no
pool).  One process per arrangement (the knob is read at context creation):

    python tools/prof/front_overlap_under_torch.py <torch|hip> [steps]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from blitzar_amd import api  # noqa: E402
import baseline_workloads as wl  # noqa: E402

kind = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 1 << 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.load()
assert api.init(api.SXT_GPU_BACKEND, 0) == 0
if kind == "torch":
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
else:
    hip = ctypes.CDLL("libamdhip64.so")
    handle = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(handle), 1) == 0  # hipStreamNonBlocking
    stream = handle
scalars = torch.from_numpy(wl.mt19937_scalars(1, n, 32, top_mask=0x0f)[0]).to(dev)
gens = torch.empty((n, 160), dtype=torch.uint8, device=dev)
lib.bzamd_ristretto255_generators_device(ctypes.c_void_p(gens.data_ptr()), 0, n, stream)
outs = torch.zeros((steps, 32), dtype=torch.uint8, device=dev)
desc = (api.sxt_sequence_descriptor * 1)()
desc[0] = api.sxt_sequence_descriptor(32, n, scalars.data_ptr(), 0)
torch.cuda.synchronize()


def run(count):
    for k in range(count):
        lib.bzamd_pipeline_next()
        lib.bzamd_msm_device(0, ctypes.c_void_p(outs[k % steps:k % steps + 1].data_ptr()), 1, desc,
                             ctypes.c_void_p(gens.data_ptr()), stream)
    lib.bzamd_pipeline_flush(stream)
    torch.cuda.synchronize()


run(60)
t0 = time.perf_counter()
run(steps)
dt = (time.perf_counter() - t0) / steps
same = bool((outs == outs[0]).all().item())
print(json.dumps({"stream": kind, "overlap_front": os.environ.get("BLITZAR_AMD_OVERLAP_FRONT", "0"),
                  "ms_per_step": dt * 1e3, "steps": steps, "outputs_agree": same}))
