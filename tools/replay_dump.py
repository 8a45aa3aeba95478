#!/usr/bin/env python3
"""Replay a BLITZAR_DUMP_DIR recording of a fixed-base multiexponentiation through this library
and compare with the recorded result (after canonicalisation: fixed-base results are projective
and not canonical in either implementation).

    python tools/replay_dump.py <dump>/packed-multiexponentiation-0 [--backend gpu|cpu]

Works on directories written by the reference's GPU backend
(sxt/multiexp/pippenger2/multiexponentiation_serialization.h:70-151) and by this library
(blitzar_amd/csrc/fixed/dump.h): same files, same element layouts.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blitzar_amd import api  # noqa: E402

CURVE_OF_NAMESPACE = {"c21t": 0, "cg1t": 1, "cn1t": 2, "cgkt": 3}
COMPACT_BYTES = {0: 120, 1: 96, 2: 64, 3: 64}


def montgomery_one(curve_id):
    """R mod p as 64-bit limbs: the Y coordinate of the canonical identity {0, R, 1}"""
    nl = {1: 6, 2: 4, 3: 4}[curve_id]
    ident = api.fold_encode(curve_id, np.zeros((0, 1, api.CURVE_LAYOUT[curve_id][2]), np.uint8)) \
        if curve_id != 1 else None
    if ident is not None:
        return ident[0, 8 * nl:16 * nl].view(np.uint64)
    # bls12-381 commitments are compressed; R is a constant of the field (field/mont.h)
    return np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba,
                     0x77ce585370525745, 0x5c071a97a256ec6d, 0x15f65ec3fa80e493], np.uint64)


def expand_generators(curve_id, raw):
    """compact_element[] -> projective elements in the ABI layout"""
    c = raw.reshape(-1, COMPACT_BYTES[curve_id]).view(np.uint64)
    n = c.shape[0]
    if curve_id == 0:
        p = np.zeros((n, 20), np.uint64)
        p[:, 0:5], p[:, 5:10], p[:, 10], p[:, 15:20] = c[:, 0:5], c[:, 5:10], 1, c[:, 10:15]
        return p
    nl = c.shape[1] // 2
    one = montgomery_one(curve_id)
    p = np.zeros((n, 3 * nl), np.uint64)
    ident = c[:, nl - 1] == np.uint64(0xFFFFFFFFFFFFFFFF)
    p[:, :2 * nl] = c
    p[:, 2 * nl:] = one
    p[ident] = 0
    p[ident, nl:2 * nl] = one
    return p


def replay(path, backend):
    with open(os.path.join(path, "meta.txt")) as fh:
        meta = fh.read()
    curve_id = next(v for k, v in CURVE_OF_NAMESPACE.items() if k in meta)
    bit_table = np.fromfile(os.path.join(path, "output_bit_table.bin"), dtype=np.uint32)
    scalars = np.fromfile(os.path.join(path, "scalars.bin"), dtype=np.uint8)
    gens = expand_generators(curve_id, np.fromfile(os.path.join(path, "generators.bin"), np.uint8))
    width = int(np.fromfile(os.path.join(path, "window_width.bin"), dtype=np.uint64)[0])
    lengths_path = os.path.join(path, "output_lengths.bin")
    os.environ.pop("BLITZAR_DUMP_DIR", None)  # do not record the replay
    if api.load().bzamd_active_backend() == 0:
        assert api.init(backend, 0) == 0
    # the handle takes the recording's window width (read when it is created); the caller's own
    # setting is put back
    before = os.environ.get("BLITZAR_PARTITION_WINDOW_WIDTH")
    os.environ["BLITZAR_PARTITION_WINDOW_WIDTH"] = str(width)
    try:
        handle = api.MultiexpHandle(curve_id, gens)
    finally:
        if before is None:
            os.environ.pop("BLITZAR_PARTITION_WINDOW_WIDTH", None)
        else:
            os.environ["BLITZAR_PARTITION_WINDOW_WIDTH"] = before
    if os.path.exists(lengths_path):
        lengths = np.fromfile(lengths_path, dtype=np.uint32)
        got = handle.vlen_multiexponentiation(bit_table, lengths, scalars)
    else:
        row = (int(bit_table.sum()) + 7) // 8
        got = handle.packed_multiexponentiation(bit_table, scalars.size // row if row else 0,
                                                scalars)
    handle.close()
    want = np.fromfile(os.path.join(path, "result.bin"), dtype=np.uint8).reshape(got.shape)
    canon = [api.fold_encode(curve_id, x[None]) for x in (got, want)]
    return curve_id, len(bit_table), bool(np.array_equal(canon[0], canon[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--backend", choices=["gpu", "cpu"], default="gpu")
    args = ap.parse_args()
    backend = api.SXT_GPU_BACKEND if args.backend == "gpu" else api.SXT_CPU_BACKEND
    curve_id, outputs, ok = replay(args.directory, backend)
    print(f"{args.directory}: curve {curve_id}, {outputs} outputs: "
          f"{'matches the recorded result' if ok else 'MISMATCH'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
