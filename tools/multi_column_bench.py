"""Device-resident curve25519 calls of k columns x 2^20 rows in throughput mode: ms per call of a
sequence (bzamd_pipeline_next) against lone calls; msm_tuning::defer_max_columns (64) is the column count
from which the engine ignores the request."""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import torch  # noqa: E402
from blitzar_amd import api  # noqa: E402

n = 1 << 20
lib = api.load()
assert api.init(api.SXT_GPU_BACKEND, 0) == 0
dev = torch.device("cuda", 0)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
gens = torch.empty((n, 160), dtype=torch.uint8, device=dev)
lib.bzamd_ristretto255_generators_device(ctypes.c_void_p(gens.data_ptr()), 0, n, stream)
g = torch.Generator(device=dev)
g.manual_seed(1)
for k in (2, 4, 8, 16, 32):
    scalars = torch.randint(0, 256, (k, n, 32), dtype=torch.uint8, device=dev, generator=g)
    scalars[:, :, 31] &= 0x0f
    desc = (api.sxt_sequence_descriptor * k)()
    for c in range(k):
        desc[c] = api.sxt_sequence_descriptor(32, n, scalars[c].data_ptr(), 0)
    out = torch.zeros((k, 32), dtype=torch.uint8, device=dev)

    def call(deferred):
        if deferred:
            lib.bzamd_pipeline_next()
        lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), k, desc,
                             ctypes.c_void_p(gens.data_ptr()), stream)

    res = {}
    for deferred in (False, True):
        for _ in range(3):
            call(deferred)
        lib.bzamd_pipeline_flush(stream)
        torch.cuda.synchronize()
        steps = max(4, 64 // k)
        t0 = time.perf_counter()
        for _ in range(steps):
            call(deferred)
        lib.bzamd_pipeline_flush(stream)
        torch.cuda.synchronize()
        res[deferred] = 1e3 * (time.perf_counter() - t0) / steps
        ref = out.cpu().numpy().copy() if not deferred else ref
        assert np.array_equal(out.cpu().numpy(), ref)
    print(f"{k:3d} columns: lone {res[False]:8.3f} ms, in sequence {res[True]:8.3f} ms, "
          f"{k * n / (res[True] * 1e-3):.3e} scalar-point ops/s")
