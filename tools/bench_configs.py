#!/usr/bin/env python3
"""BASELINE.json configs[2..4] on one MI355X (device-resident operands), each with a full-size
parity check that does not need the oracle to run the whole MSM:

  generators g_i = (i + 1) * G  (bzamd_generator_multiples_device, G = the reference's
  generate_random_element(rng{1, 2})), so  sum_i a_i g_i = (sum_i a_i (i + 1) mod r) * G  and the
  right-hand side is one scalar multiplication done with the reference's own curve operations
  (oracle/_ref) and canonicalised by the reference's own encoder.

    python tools/bench_configs.py [--only 3,4,5] [--steps 3]

Prints one JSON line per config.  Not the driver's bench (that is bench.py, config 2).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blitzar_amd import api  # noqa: E402
from oracle import ref_oracle  # noqa: E402

ORDER = {
    1: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    2: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    3: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
}
STAGES = ["prepare_addends", "recode", "bucket_sort", "accumulate", "reduce", "combine"]


def vp(t):
    return ctypes.c_void_p(t.data_ptr())


def weighted_scalar_sum(scalars_u8, order):
    """sum_i a_i * (i + 1) mod order for little-endian rows (numpy, 16-bit limbs in uint64)"""
    n, nbytes = scalars_u8.shape
    if nbytes % 2:
        scalars_u8 = np.concatenate([scalars_u8, np.zeros((n, 1), np.uint8)], axis=1)
    limbs = np.ascontiguousarray(scalars_u8).view("<u2")
    idx = np.arange(1, n + 1, dtype=np.uint64)
    total = 0
    for k in range(limbs.shape[1]):
        total += int((limbs[:, k].astype(np.uint64) * idx).sum(dtype=np.uint64)) << (16 * k)
    return total % order


def scalar_mul(cid, base_projective, k):
    acc = None
    for bit in range(k.bit_length() - 1, -1, -1):
        if acc is not None:
            acc = ref_oracle.double_projective(cid, acc)
        if (k >> bit) & 1:
            acc = base_projective if acc is None else ref_oracle.add_projective(cid, acc,
                                                                                 base_projective)
    return acc


def expected_commitment(cid, base_affine, scalars_u8):
    s = weighted_scalar_sum(scalars_u8, ORDER[cid])
    g = ref_oracle.affine_to_projective(cid, base_affine)[0]
    if s == 0:
        return ref_oracle.canonical(cid, ref_oracle.affine_to_projective(
            cid, ref_oracle.identity_affine(cid))[0])
    return ref_oracle.canonical(cid, scalar_mul(cid, g, s))


def device_generators(lib, cid, n, dev, stream):
    base = ref_oracle.random_affine(cid, 1, 2)
    stride = api.CURVE_LAYOUT[cid][0]
    d_base = torch.from_numpy(base.copy()).to(dev)
    gens = torch.empty((n, stride), dtype=torch.uint8, device=dev)
    lib.bzamd_generator_multiples_device(cid, vp(gens), vp(d_base), n, stream)
    torch.cuda.synchronize()
    return base, gens


def timed(lib, fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    lib.bzamd_stage_timing_begin(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms = (ctypes.c_double * 6)()
    calls = lib.bzamd_stage_timing_collect(ms)
    stages = {STAGES[i]: round(ms[i] / max(calls, 1), 3) for i in range(6)}
    return dt, stages, calls


def variable_base(lib, cid, name, log2n, columns, steps, dev, stream, check_columns):
    n = 1 << log2n
    base, gens = device_generators(lib, cid, n, dev, stream)
    g = torch.Generator(device=dev)
    g.manual_seed(42 + cid)
    scalars = torch.randint(0, 256, (columns, n, 32), dtype=torch.uint8, device=dev, generator=g)
    scalars[:, :, 31] &= 0x0f  # < 2^252 < group order of all three curves
    out = torch.zeros((columns, api.CURVE_LAYOUT[cid][1]), dtype=torch.uint8, device=dev)
    desc = (api.sxt_sequence_descriptor * columns)()
    for c in range(columns):
        desc[c] = api.sxt_sequence_descriptor(32, n, scalars[c].data_ptr(), 0)

    def step():
        lib.bzamd_msm_device(cid, vp(out), columns, desc, vp(gens), stream)

    dt, stages, calls = timed(lib, step, steps)
    got = out.cpu().numpy()
    ok = True
    for c in check_columns:
        want = expected_commitment(cid, base, scalars[c].cpu().numpy())
        ok = ok and np.array_equal(got[c], want.view(np.uint8).reshape(-1)[:got.shape[1]])
    ops = columns * n
    return {"config": name, "curve_id": cid, "rows": n, "columns": columns,
            "ms_per_call": round(dt * 1e3, 3), "scalar_point_ops_per_s": ops / dt,
            "commitments_per_s": columns / dt, "batches_per_call": calls / steps,
            "stage_ms_per_batch": stages, "parity_full_size": bool(ok),
            "algorithmic_bytes": ops * 32 + n * api.CURVE_LAYOUT[cid][0]}


def fixed_base(lib, cid, name, log2n, outputs, steps, dev, stream, check_outputs):
    n = 1 << log2n
    base, gens = device_generators(lib, cid, n, dev, stream)
    # handle from projective host generators (sxt_multiexp_handle_new consumes element_p2)
    proj = ref_oracle.affine_to_projective(cid, gens.cpu().numpy())
    t0 = time.perf_counter()
    handle = api.MultiexpHandle(cid, proj)
    handle_s = time.perf_counter() - t0
    bit_table = np.array([(8, 32, 256)[i % 3] for i in range(outputs)], dtype=np.uint32)
    row_bytes = (int(bit_table.sum()) + 7) // 8
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    scalars = torch.randint(0, 256, (n, row_bytes), dtype=torch.uint8, device=dev, generator=g)
    # keep the 256-bit fields below the group order: clear the top nibble of each
    offs = (np.concatenate([[0], np.cumsum(bit_table)[:-1]]) // 8).astype(np.int64)
    wide = torch.from_numpy(offs[bit_table == 256] + 31).to(dev)
    scalars[:, wide] &= 0x0f
    psize = api.CURVE_LAYOUT[cid][2]
    res = torch.zeros((outputs, psize), dtype=torch.uint8, device=dev)
    bt = np.ascontiguousarray(bit_table)

    def step():
        lib.bzamd_fixed_packed_multiexponentiation_device(
            vp(res), handle._h, bt.ctypes.data_as(ctypes.c_void_p), None, outputs, n, vp(scalars),
            stream)

    dt, stages, calls = timed(lib, step, steps)
    got = res.cpu().numpy()
    ok = True
    for k in check_outputs:
        nb = int(bit_table[k]) // 8
        col = scalars[:, int(offs[k]):int(offs[k]) + nb].cpu().numpy()
        want = expected_commitment(cid, base, col)
        have = ref_oracle.canonical(cid, got[k].view(np.uint64))
        ok = ok and np.array_equal(have.view(np.uint8).reshape(-1), want.view(np.uint8).reshape(-1))
    handle.close()
    ops = outputs * n
    return {"config": name, "curve_id": cid, "rows": n, "outputs": outputs,
            "bits_per_row": int(bit_table.sum()), "ms_per_call": round(dt * 1e3, 3),
            "row_output_ops_per_s": ops / dt, "outputs_per_s": outputs / dt,
            "batches_per_call": calls / steps, "stage_ms_per_batch": stages,
            "handle_creation_s": round(handle_s, 3), "parity_full_size": bool(ok),
            "algorithmic_bytes": n * row_bytes + outputs * psize}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="3,4,5")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--scale", type=int, default=0, help="subtract from every log2 size (smoke)")
    args = ap.parse_args()
    only = {int(x) for x in args.only.split(",")}
    assert torch.cuda.is_available() and ref_oracle.available()
    dev = torch.device("cuda", 0)
    lib = api.load()
    assert api.init(api.SXT_GPU_BACKEND, 0) == 0
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    s = args.scale
    if 3 in only:
        print(json.dumps(variable_base(lib, 1, "3: bls12-381 G1 MSM n=2^22", 22 - s, 1, args.steps,
                                       dev, stream, [0])), flush=True)
    if 4 in only:
        # 256 columns over 8 GPUs = 32 columns per GPU
        print(json.dumps(variable_base(lib, 2, "4: bn254 G1 32 of 256 columns x 2^20 (one GPU's "
                                       "shard)", 20 - s, 32, args.steps, dev, stream, [0, 31])),
              flush=True)
    if 5 in only:
        # 1024 outputs over 8 GPUs = 128 outputs per GPU
        print(json.dumps(fixed_base(lib, 3, "5: grumpkin packed fixed-base 128 of 1024 outputs x "
                                    "2^18 (one GPU's shard)", 18 - s, 128 if s == 0 else 12,
                                    args.steps, dev, stream, [0, 1, 2])), flush=True)
    api.reset_for_testing()


if __name__ == "__main__":
    main()
