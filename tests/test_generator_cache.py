"""BLITZAR_AMD_GENERATOR_CACHE=1 (opt-in; include/blitzar_amd.h, api/state.h): the blocking
sxt_*_compute_pedersen_commitments_with_generators entry points keep a caller's host generators on the
device across calls -- what the reference re-uploads on every call
(sxt/multiexp/bucket_method/accumulation.h:68-71).  The first call with a (pointer, count, curve,
sample hash) key only remembers it, the second registers the set, later ones are served from it;
every call returns the reference's bytes; a rewritten sampled row, another array and another length
are new keys; without the knob nothing is kept.  A child process: sxt_init reads the knob."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import ctypes, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from blitzar_amd import api
from tests import util
lib = api.load()
assert api.init(api.SXT_GPU_BACKEND, 0) == 0
def stats():
    h, b = ctypes.c_uint64(), ctypes.c_uint64()
    lib.bzamd_generator_cache_stats(ctypes.byref(h), ctypes.byref(b))
    return [int(h.value), int(b.value)]
rng = np.random.default_rng(31)
out = {{"commitments": [], "stats": []}}
for cid, n in {cases}:
    gens = util.generators_for(cid, n)
    g = np.ascontiguousarray(util.api_generators(cid, gens))       # ONE host array, call after call
    for k in range(4):
        cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
                (rng.integers(0, 256, (n - 7, 4), dtype=np.uint8), True)]
        out["commitments"].append(api.compute_pedersen_commitments(cid, cols, generators=g).tolist())
        out["stats"].append(stats())
    # row 0 is in the sample: rewriting it in place is a new key (generator 1 moves to row 0)
    g[0] = g[1]
    cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False)]
    out["commitments"].append(api.compute_pedersen_commitments(cid, cols, generators=g).tolist())
    out["stats"].append(stats())
    # a shorter call on the same array: another key (the count is part of it)
    cols = [(rng.integers(0, 256, (n - 256, 32), dtype=np.uint8), False)]
    out["commitments"].append(api.compute_pedersen_commitments(cid, cols, generators=g).tolist())
    out["stats"].append(stats())
print("RESULT" + json.dumps(out))
"""


def _expected(oracle, cases):
    rng = np.random.default_rng(31)
    want = []
    for cid, n in cases:
        gens = util.generators_for(cid, n)
        for _ in range(4):
            cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
                    (rng.integers(0, 256, (n - 7, 4), dtype=np.uint8), True)]
            want.append(oracle.commit(cid, cols, gens))
        moved = gens.copy()
        moved[0] = moved[1]
        cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False)]
        want.append(oracle.commit(cid, cols, moved))
        cols = [(rng.integers(0, 256, (n - 256, 32), dtype=np.uint8), False)]
        want.append(oracle.commit(cid, cols, moved))
    return want


def _run(cases, knob):
    env = {k: v for k, v in os.environ.items() if k != "BLITZAR_AMD_GENERATOR_CACHE"}
    if knob is not None:
        env["BLITZAR_AMD_GENERATOR_CACHE"] = knob
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, cases=cases)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT"))[6:])


def test_cached_generators_give_the_reference_bytes(oracle):
    cases = [(0, 1 << 14), (2, (1 << 14) + 40)]
    got = _run(cases, "1")
    want = _expected(oracle, cases)
    assert len(got["commitments"]) == len(want)
    for g, w in zip(got["commitments"], want):
        assert np.array_equal(np.array(g, dtype=np.uint8), w)
    # [hits, builds] after each call: remember, register, hit, hit | new key | new key
    s = got["stats"]
    assert s[:6] == [[0, 0], [0, 1], [1, 1], [2, 1], [2, 1], [2, 1]]
    assert s[6:] == [[2, 1], [2, 2], [3, 2], [4, 2], [4, 2], [4, 2]]


def test_without_the_knob_nothing_is_kept(oracle):
    cases = [(0, 1 << 14)]
    got = _run(cases, None)
    for g, w in zip(got["commitments"], _expected(oracle, cases)):
        assert np.array_equal(np.array(g, dtype=np.uint8), w)
    assert all(s == [0, 0] for s in got["stats"])
