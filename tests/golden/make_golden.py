#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own CPU backend (oracle/_ref, compiled from
/root/reference by oracle/ref/build_ref.py).  Run where /root/reference is mounted:

    python tests/golden/make_golden.py

The fixtures are small seeded input/output vectors for the MSM hot path; they travel to the GPU
box (which has no /root/reference) and pin both the oracle build and the HIP path.
Also embedded: the only byte-level KATs the reference ships for this path,
rust/tests/src/main.rs:22-47.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import fixed_base, ref_oracle  # noqa: E402
from tests import util  # noqa: E402

RUST_KAT_DATA = [[2000, 7500, 5000, 1500], [5000, 0, 400000, 10], [7000, 7500, 405000, 1510]]
RUST_KAT = [
    [4, 105, 58, 131, 59, 69, 150, 106, 120, 137, 32, 225, 175, 244, 82, 115, 216, 180, 206, 150,
     21, 250, 240, 98, 251, 192, 146, 244, 54, 169, 199, 97],
    [2, 254, 178, 195, 198, 238, 44, 156, 24, 29, 88, 196, 37, 63, 157, 50, 236, 159, 61, 49, 153,
     181, 79, 126, 55, 188, 67, 1, 228, 248, 72, 51],
    [30, 237, 163, 234, 252, 111, 45, 133, 235, 227, 21, 117, 229, 188, 88, 149, 240, 109, 205, 90,
     6, 130, 199, 152, 5, 221, 57, 231, 168, 9, 141, 122],
]


def golden_columns(seed, n):
    """deterministic column set shared by the generator and the tests (tests/util.py)"""
    return util.mixed_columns(np.random.default_rng(seed), n)


def main():
    assert ref_oracle.available(), "build oracle/_ref first (python oracle/ref/build_ref.py)"
    out = {}
    out["rust_kat_data"] = np.array(RUST_KAT_DATA, dtype=np.uint32)
    out["rust_kat"] = np.array(RUST_KAT, dtype=np.uint8)
    # built-in generators (raw 5x51 limbs are observable) + one-commit prefix sums
    out["ristretto_generators_0_8"] = ref_oracle.ristretto_generators(8, 0)
    out["ristretto_generators_1000_4"] = ref_oracle.ristretto_generators(4, 1000)
    out["one_commit_0_1_5_33"] = np.stack([ref_oracle.one_commit(k) for k in (0, 1, 5, 33)])
    n = 48
    for cid in (0, 1, 2, 3):
        gens = util.generators_for(cid, n)
        cols = golden_columns(1000 + cid, n)
        out[f"curve{cid}_generators"] = util.api_generators(cid, gens)
        out[f"curve{cid}_commitments"] = ref_oracle.commit(cid, cols, gens)
        # fixed-base: packed bit table over 11 generators, window width 4, and the partition-table
        # file image the reference would write for them
        m = 11
        proj = gens[:m] if cid == 0 else ref_oracle.affine_to_projective(cid, gens[:m])
        table = fixed_base.PartitionTable(cid, proj, 4)
        bit_table = [3, 1, 8, 13, 64, 256]
        row = (sum(bit_table) + 7) // 8
        scalars = np.random.default_rng(2000 + cid).integers(0, 256, (m, row), dtype=np.uint8)
        res = fixed_base.multiexponentiate(table, bit_table, m, scalars)
        out[f"curve{cid}_fixed_projective_generators"] = proj
        out[f"curve{cid}_fixed_scalars"] = scalars
        out[f"curve{cid}_fixed_canonical"] = np.stack(
            [ref_oracle.canonical(cid, r).view(np.uint8).reshape(-1) for r in res])
        out[f"curve{cid}_table_w4_sha256"] = np.frombuffer(
            hashlib.sha256(table.file_bytes()).digest(), dtype=np.uint8)
    out["fixed_bit_table"] = np.array([3, 1, 8, 13, 64, 256], dtype=np.uint32)
    path = os.path.join(HERE, "msm_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes, {len(out)} arrays)")


if __name__ == "__main__":
    main()
