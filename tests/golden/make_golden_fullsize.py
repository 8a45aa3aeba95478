#!/usr/bin/env python3
"""Full-size golden commitments for the three Weierstrass curves, computed ONCE on the CPU box by
the reference's own CPU backend (oracle/_ref) and committed as tests/golden/fullsize_golden.npz:

    python tests/golden/make_golden_fullsize.py          # ~15 min, ~8 GB
    python tests/golden/make_golden_fullsize.py --only curve1_1x2^22   # add / refresh one case

Inputs are rebuilt from their recipe wherever the fixture is used (they are 100 MB apiece):
  * scalars: the `std::mt19937{0}` byte stream of the reference benchmarks
    (benchmark/multi_commitment/benchmark.m.cc:141-156), FULL-WIDTH 32-byte values (nothing masked),
    column-major;
  * generators: the reference's generate_random_element(rng{i + 1, i + 2}) for the first 1024
    indices, then the chain g_i = g_{i-1} + g_0, index 5 replaced by the identity
    (tests/util.py weierstrass_generators_big == weierstrass_generators).
The file holds the commitments (canonical bytes, as the C ABI returns them) and a SHA-256 of every
input array, so a test that rebuilds the inputs knows it committed to the same bytes.  Nothing the
product computes enters this file.
Reference tests of the same shape (random full-width columns against the CPU backend):
sxt/multiexp/test/multiexponentiation.cc:290-451, cbindings/pedersen.t.cc:368-460.
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import ref_oracle  # noqa: E402
from tests import util  # noqa: E402

# (curve id, columns, log2 rows)
# (the last one is BASELINE config 3's own shape: one bls12-381 column of 2^22 rows)
CASES = [(1, 1, 20), (2, 2, 20), (3, 1, 18), (1, 1, 22)]
DISTINCT_SEEDS = 1024


def inputs(cid, columns, log2n):
    """(scalars [columns, n, 32] uint8, generators [n, stride] uint8) of a case"""
    import baseline_workloads as wl
    n = 1 << log2n
    scalars = wl.mt19937_scalars(columns, n, 32)
    gens = util.weierstrass_generators_big(cid, n, DISTINCT_SEEDS)
    return scalars, gens


# rows the reference CPU backend takes in one call: its multiproduct index table counts 32-bit entries
# (one per set scalar bit: 2^22 rows x 256 bits overflow it -- std::bad_alloc from
# mtxpi::compute_multiexponentiation at BASELINE config 3's own shape).  A longer column is committed
# in row chunks of this size by the reference's MSM (raw projective results) and the chunks are summed
# with the reference's own addition and encoded by its own canonicaliser: exact group arithmetic, so
# the bytes are those of the whole column.
ROW_CHUNK = 1 << 20


def reference_commit(cid, scalars, gens):
    columns, n = scalars.shape[0], scalars.shape[1]
    if n <= ROW_CHUNK:
        return ref_oracle.commit(cid, [(scalars[c], False) for c in range(columns)], gens)
    total = None
    for lo in range(0, n, ROW_CHUNK):
        part = ref_oracle.msm_projective(
            cid, [(scalars[c, lo:lo + ROW_CHUNK], False) for c in range(columns)],
            gens[lo:lo + ROW_CHUNK])
        total = part if total is None else np.stack(
            [ref_oracle.add_projective(cid, total[c], part[c]) for c in range(columns)])
    return np.stack([ref_oracle.canonical(cid, total[c]) for c in range(columns)])


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def main():
    assert ref_oracle.available(), "build oracle/_ref first (python oracle/ref/build_ref.py)"
    out = {}
    path = os.path.join(HERE, "fullsize_golden.npz")
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only is not None and os.path.exists(path):
        out = dict(np.load(path))  # keep the other cases as they are
    for cid, columns, log2n in CASES:
        if only is not None and only != f"curve{cid}_{columns}x2^{log2n}":
            continue
        t0 = time.time()
        scalars, gens = inputs(cid, columns, log2n)
        t1 = time.time()
        got = reference_commit(cid, scalars, gens)
        print(f"curve {cid}: {columns} x 2^{log2n} rows: inputs {t1 - t0:.0f} s, reference CPU "
              f"backend {time.time() - t1:.0f} s", flush=True)
        key = f"curve{cid}_{columns}x2^{log2n}"
        out[key + "_commitments"] = got
        out[key + "_scalars_sha256"] = sha(scalars)
        out[key + "_generators_sha256"] = sha(gens)
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
