"""Round-6 additions that have no file of their own: stage timing on a SAMPLE of the calls (what
bench.py's timed region uses: an event pair around one k_accumulate in four), and the planner's
choice of accumulation segments for short launches."""
import ctypes

import numpy as np
import pytest

from tests import hooks, util


def test_short_launches_take_short_segments():
    """below one wavefront per SIMD (2^16 lanes) a launch takes fewer entries per lane, down to 8:
    k_accumulate is then a chain of that many dependent additions (2^14 rows: 0.10 -> 0.04 ms)"""
    seg = lambda ns, bits: int(hooks.plan(ns, bits, [0] * len(ns))[1][6])   # noqa: E731
    assert [seg([1 << k], [256]) for k in (10, 12, 14, 16, 17, 18, 20)] == [3, 3, 3, 4, 5, 5, 5]
    assert [seg([1 << k], [8]) for k in (12, 16, 18, 20, 22)] == [3, 3, 3, 4, 5]
    # the rule looks at the LAUNCH: ten such columns fill the machine at 32 entries per lane
    assert seg([1 << 16] * 10, [256] * 10) == 5


@pytest.mark.gpu
def test_stage_timing_on_a_sample_of_the_calls(gpu_backend, oracle):
    import torch
    api = gpu_backend
    lib = api.load()
    n = 5000
    rng = np.random.default_rng(6)
    gens = util.generators_for(0, n)
    g = torch.from_numpy(np.ascontiguousarray(util.api_generators(0, gens))).cuda()
    col = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d_col = torch.from_numpy(col).cuda()
    desc = (api.sxt_sequence_descriptor * 1)()
    desc[0] = api.sxt_sequence_descriptor(32, n, d_col.data_ptr(), 0)
    out = torch.zeros((1, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    lib.bzamd_stage_timing_begin_sampled(100, 1 << 3, 4)       # accumulate only, one call in four
    for _ in range(10):
        lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 1, desc,
                             ctypes.c_void_p(g.data_ptr()), None)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 6)()
    assert lib.bzamd_stage_timing_collect(ms) == 3             # calls 0, 4, 8
    assert ms[3] > 0 and all(ms[i] == 0 for i in (0, 1, 2, 4, 5))
    assert np.array_equal(out.cpu().numpy(), oracle.commit(0, [(col, False)], gens))


BUSY_CHILD = r"""
import ctypes, json, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, {root!r})
from blitzar_amd import api
from tests import util
lib = api.load()
assert api.init(api.SXT_GPU_BACKEND, 0) == 0
n = 1 << 16
rng = np.random.default_rng(16)
col = rng.integers(0, 256, (n, 32), dtype=np.uint8)
col[:, 31] &= 0x0f                      # 252-bit scalars: the top window's records sit in a few buckets
gens = util.generators_for(0, n)
g = torch.from_numpy(np.ascontiguousarray(util.api_generators(0, gens))).cuda()
d_col = torch.from_numpy(col).cuda()
desc = (api.sxt_sequence_descriptor * 1)()
desc[0] = api.sxt_sequence_descriptor(32, n, d_col.data_ptr(), 0)
out = torch.zeros((1, 32), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
probe = (ctypes.c_double * 4)()
busy = threading.Thread(target=lambda: lib.bzamd_probe_mad_rate(1500.0, probe))
busy.start()                            # every SIMD at 6 waves of multiplies for 1.5 s, on its own stream
time.sleep(0.3)
t0 = time.time()
for _ in range(20):
    lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 1, desc, ctypes.c_void_p(g.data_ptr()), None)
torch.cuda.synchronize()
took = time.time() - t0
busy.join()
np.save({cols_file!r}, col)
print("RESULT" + json.dumps({{"out": out.cpu().numpy().tolist(), "took_s": took, "probe_ms": probe[3]}}))
"""


@pytest.mark.gpu
def test_short_uniform_column_beside_a_long_running_kernel(oracle, tmp_path):
    """a short launch sends the top window's oversized bucket group through the chunked
    path of pass 2, whose workers meet at a barrier INSIDE one launch and must all be resident.  Another
    stream that holds every SIMD's wave slots for a long time (the ALU probe: 1.5 s of 6 waves per
    SIMD) can delay them -- its workgroups finish and the workers get their slots -- but never hang
    them: twenty 2^16-row commitments complete beside it, with the reference's bytes.  A child process
    under a timeout: a hang fails, it does not hang the suite."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cols_file = str(tmp_path / "col.npy")
    r = subprocess.run([sys.executable, "-c", BUSY_CHILD.format(root=root, cols_file=cols_file)],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT"))[6:])
    assert got["probe_ms"] >= 1000.0                      # the neighbour really ran that long
    col = np.load(cols_file)
    want = oracle.commit(0, [(col, False)], util.generators_for(0, 1 << 16))
    assert np.array_equal(np.array(got["out"], dtype=np.uint8), want)


@pytest.mark.gpu
@pytest.mark.parametrize("curve_id", [0, 1, 2, 3])
@pytest.mark.parametrize("n", [200, 6000])
def test_first_entry_of_a_segment_is_the_identity_generator(gpu_backend, oracle, curve_id, n):
    """k_accumulate loads the first entry of a segment instead of adding it to the identity (C::first).
    Here that entry is known: row 5 -- the identity among the Weierstrass generators of tests/util.py --
    is the only row whose low scalar bits are 1, every other row's are 2 or 3, so it sorts to the front
    of window 0 whatever the window width; the other columns put a negative digit (two's complement
    -1) and a lone entry (every other scalar zero) there."""
    api = gpu_backend
    rng = np.random.default_rng(600 + 7 * curve_id + n)
    gens = util.generators_for(curve_id, n)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0f
    a[:, 0] = rng.integers(2, 4, n, dtype=np.uint8)
    a[:, 1] = 0
    a[5, 0] = 1
    signed = rng.integers(0, 128, (n, 8), dtype=np.uint8)      # positive 64-bit values ...
    signed[:, 0] = rng.integers(2, 4, n, dtype=np.uint8)
    signed[:, 1] = 0
    signed[5] = 0xff                                           # ... and -1 in row 5
    lone = np.zeros((n, 32), dtype=np.uint8)
    lone[5, 0] = 1
    lone[n - 1, 3] = 7
    cols = [(a, False), (signed, True), (lone, False)]
    got = api.compute_pedersen_commitments(curve_id, cols, generators=util.api_generators(curve_id, gens))
    want = oracle.commit(curve_id, cols, gens)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"columns {bad.tolist()} differ"
