"""Window tables of resident generator sets (msm/plan.h `window_table`): slices 2^(bits w) g_i kept
in HBM so that all windows of a column share one bucket set, at table widths 16, 17, 18 and 20.  The tables are built for sets of
2^14 generators or more; BLITZAR_AMD_WINDOW_TABLE_MIN lowers that so that oracle-sized inputs take
the merged path, and BLITZAR_AMD_FORCE_WINDOW_TABLES overrides the planner's cost model, which
would keep such short columns on separate windows (a child process: the environment is read when
a set is registered / a context is created).  Every
resident entry point must give the reference's bytes with tables on: built-in generators with and
without an offset, bzamd_generators, fixed-base handles (plain / packed / vlen), columns that
merge next to columns that do not (signed, short, narrow)."""
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
import torch
sys.path.insert(0, {root!r})
from blitzar_amd import api
from tests import util
lib = api.load()
n = {n}
assert api.init(api.SXT_GPU_BACKEND, n) == 0
rng = np.random.default_rng({seed})
out = {{}}
def columns(rows):
    return [(rng.integers(0, 256, (rows, 32), dtype=np.uint8), False),        # merges: 17 windows
            (rng.integers(0, 256, (rows - 5, 20), dtype=np.uint8), False),    # merges: 11 windows
            (rng.integers(0, 256, (rows, 2), dtype=np.uint8), False),         # one window: not merged
            (rng.integers(0, 256, (rows, 8), dtype=np.uint8), True),          # signed: not merged
            (rng.integers(0, 256, (rows // 3, 32), dtype=np.uint8), False),   # short: not merged
            (np.full((rows, 32), 0xff, np.uint8), False)]                      # 2^256 - 1: top carry
launches = lib.bzamd_kernel_launch_count()
out["builtin"] = api.compute_pedersen_commitments(0, columns(n)).tolist()
out["builtin_offset"] = api.compute_pedersen_commitments(0, columns(n - 40), offset_generators=40).tolist()
dev = torch.device("cuda", 0)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for cid in {curves}:
    gens = util.generators_for(cid, n)
    g_host = np.ascontiguousarray(util.api_generators(cid, gens))
    cols = columns(n)
    d_cols = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c, _ in cols]
    desc = (api.sxt_sequence_descriptor * len(cols))()
    for i, ((c, s), d) in enumerate(zip(cols, d_cols)):
        desc[i] = api.sxt_sequence_descriptor(c.shape[1], c.shape[0], d.data_ptr(), int(s))
    h = lib.bzamd_generators_new_host(cid, g_host.ctypes.data_as(ctypes.c_void_p), n)
    res = torch.zeros((len(cols), api.CURVE_LAYOUT[cid][1]), dtype=torch.uint8, device=dev)
    lib.bzamd_msm_device_resident(ctypes.c_void_p(res.data_ptr()), len(cols), desc, h, stream)
    torch.cuda.synchronize()
    out[f"resident{{cid}}"] = res.cpu().numpy().tolist()
    lib.bzamd_generators_free(h)
    proj = gens if cid == 0 else util.ref_oracle.affine_to_projective(cid, gens)
    hd = api.MultiexpHandle(cid, proj)
    bt = [8, 32, 256, 5, 1, 64, 200, 256]
    s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
    out[f"packed{{cid}}"] = hd.packed_multiexponentiation(bt, n, s).tolist()
    lengths = [0, 5, n // 2 + 1, n - 9, n - 9, n - 1, n, n]
    out[f"vlen{{cid}}"] = hd.vlen_multiexponentiation(bt, lengths, s).tolist()
    s2 = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    out[f"plain{{cid}}"] = hd.multiexponentiation(32, 2, n, s2).tolist()
    hd.close()
out["launches"] = int(lib.bzamd_kernel_launch_count() - launches)
print("RESULT" + json.dumps(out))
"""


def expected(oracle, curves, n, seed):
    from oracle import fixed_base
    rng = np.random.default_rng(seed)

    def columns(rows):
        return [(rng.integers(0, 256, (rows, 32), dtype=np.uint8), False),
                (rng.integers(0, 256, (rows - 5, 20), dtype=np.uint8), False),
                (rng.integers(0, 256, (rows, 2), dtype=np.uint8), False),
                (rng.integers(0, 256, (rows, 8), dtype=np.uint8), True),
                (rng.integers(0, 256, (rows // 3, 32), dtype=np.uint8), False),
                (np.full((rows, 32), 0xff, np.uint8), False)]
    want = {"builtin": oracle.commit(0, columns(n), oracle.ristretto_generators(n))}
    want["builtin_offset"] = oracle.commit(0, columns(n - 40), oracle.ristretto_generators(n - 40, 40))
    for cid in curves:
        gens = util.generators_for(cid, n)
        want[f"resident{cid}"] = oracle.commit(cid, columns(n), gens)
        bt = [8, 32, 256, 5, 1, 64, 200, 256]
        s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
        want[f"packed{cid}"] = oracle.commit(cid, fixed_base.unpack_columns(bt, n, s), gens)
        lengths = [0, 5, n // 2 + 1, n - 9, n - 9, n - 1, n, n]
        want[f"vlen{cid}"] = oracle.commit(cid, fixed_base.unpack_columns(bt, n, s, lengths), gens)
        s2 = rng.integers(0, 256, (n, 64), dtype=np.uint8)
        want[f"plain{cid}"] = oracle.commit(cid, fixed_base.unpack_columns([256, 256], n, s2), gens)
    return want


# bits: the window width the table is built for.  16 = int16 digits (the default of every set but the
# one below); 17 = what Weierstrass sets of 2^20 generators and more take by default (api/state.h,
# resident_table::build: 16 slices, 2^16 buckets per merged column), 18 and 20 = wider still: the wide
# form -- 32-bit digits -- with 2^17 / 2^19 buckets per merged column and 15 / 13 slices
@pytest.mark.parametrize("curves,n,bits", [([0], 3000, 16), ([1], 1200, 16), ([2, 3], 2000, 16),
                                           ([0], 2000, 17), ([1], 1000, 17), ([2, 3], 1800, 17),
                                           ([0], 3000, 18), ([1], 1200, 18), ([2, 3], 2000, 18),
                                           ([0], 1500, 20), ([1], 700, 20), ([2, 3], 1000, 20)])
def test_window_tables_match_oracle(oracle, curves, n, bits):
    seed = 900 + n
    env = dict(os.environ, BLITZAR_AMD_WINDOW_TABLE_MIN="64", BLITZAR_AMD_FORCE_WINDOW_TABLES="1",
               BLITZAR_AMD_WINDOW_TABLE_BITS=str(bits))
    code = CHILD.format(root=ROOT, n=n, seed=seed, curves=curves)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT"))[6:])
    want = expected(oracle, curves, n, seed)
    for key, w in want.items():
        g = np.array(got[key], dtype=np.uint8)
        if key.startswith(("packed", "vlen", "plain")):
            cid = int(key[-1])
            words = g.view(np.uint64).reshape(g.shape[0], -1)
            g = np.stack([np.ascontiguousarray(oracle.canonical(cid, p)).view(np.uint8).reshape(-1)
                          for p in words])[:, :w.shape[1]]
        bad = np.nonzero((g != w).any(axis=1))[0]
        assert bad.size == 0, f"{key}: outputs {bad.tolist()} differ"
