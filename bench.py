#!/usr/bin/env python3
"""Hot-path benchmark: variable-base MSM / Pedersen commitments on MI355X through the C ABI.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: on every rank one
commitment column of 2^20 uniformly random 252-bit scalars against 2^20 resident ristretto
generators (BASELINE.json configs[1]; generators = the built-in compute_base_element(i), the
recipe of benchmark/multi_commitment/benchmark.m.cc:141-156).  Inputs are resident in HBM before
the timed region.  With N ranks the job is N independent columns (columns shard),
followed by one RCCL all-gather of the N 32-byte commitments -- weak scaling.

Prints ONE JSON line (rank 0): metric = scalar-point ops / s over the whole job.
Extra legs, rank 0 / N = 1 only: `roofline` (dominant kernel k_accumulate: algorithmic bytes per
launch / its HIP-event duration vs 8 TB/s) and `cpu_baseline` (the reference's own CPU backend,
oracle/_ref, timed on a bounded sample of the same workload on this box's host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch  # first: libblitzar_amd.so must bind to the HIP runtime torch already loaded

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from blitzar_amd import api  # noqa: E402

WORKLOADS = {
    # name: (curve_id, log2 rows, scalar bytes, top-byte mask, generator bytes per row (C ABI))
    "curve25519_msm_n2^20_252bit": (api.SXT_CURVE_RISTRETTO255, 20, 32, 0x0f, 160),
}
STAGES = ["prepare_addends", "recode", "bucket_sort", "accumulate", "reduce", "combine"]
HBM_PEAK_GBS = 8000.0
# integer-ALU side of k_accumulate (the honest binding bound).  736 v_mad_u64_u32 per
# bucket addition is the count in the kernel's ISA (8 field products x 92: 81 limb products, 9
# wrap-arounds, 2 folds); the peak is the measured issue rate of that instruction with every SIMD
# busy, tools/ubench/valu_rates.hip: one wave-instruction per 5.6 cycles per SIMD at the 2.4 GHz
# the device reports, x 1024 SIMDs.
MADS_PER_ADDITION = 736
PEAK_WAVE_MADS_PER_S = 1024 * 2.4e9 / 5.6


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=None, help="override rows (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=20,
                    help="rows of the bounded CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(log2n):
    """reference CPU backend (oracle/_ref) on a bounded sample of the same workload"""
    from oracle import ref_oracle
    if not ref_oracle.available():
        return None
    n = 1 << log2n
    rng = np.random.default_rng(0)
    scalars = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    scalars[:, 31] &= 0x0f
    gens = ref_oracle.ristretto_generators(n)
    t0 = time.perf_counter()
    ref_oracle.commit(0, [(scalars, False)], gens)
    dt = time.perf_counter() - t0
    return {
        "value": n / dt,
        "unit": "scalar-point ops/s",
        "cores": 1,
        "kind": "reference",
        "sample": f"1 column x 2^{log2n} rows of the same workload (252-bit scalars, built-in "
                  f"generators), {dt:.2f} s on 1 of {os.cpu_count()} host cores; the reference cpu "
                  "backend is single-threaded and its ops/s falls with n",
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm

    name = "curve25519_msm_n2^20_252bit"
    curve_id, log2n, nbytes, top_mask, gen_bytes = WORKLOADS[name]
    if args.log2n is not None:
        log2n = args.log2n
    n = 1 << log2n

    lib = api.load()
    assert api.init(api.SXT_GPU_BACKEND, 0) == 0
    stream = torch.cuda.current_stream()
    sh = ctypes.c_void_p(stream.cuda_stream)

    # synthetic inputs, resident in HBM: per-rank scalar column, shared generator set
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    scalars = torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device=dev, generator=g)
    scalars[:, nbytes - 1] &= top_mask
    generators = torch.empty((n, gen_bytes), dtype=torch.uint8, device=dev)
    lib.bzamd_ristretto255_generators_device(ctypes.c_void_p(generators.data_ptr()), 0, n, sh)
    out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    gathered = torch.zeros((world, 32), dtype=torch.uint8, device=dev)
    desc = (api.sxt_sequence_descriptor * 1)()
    desc[0] = api.sxt_sequence_descriptor(nbytes, n, scalars.data_ptr(), 0)
    torch.cuda.synchronize()

    def step():
        lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 1, desc,
                             ctypes.c_void_p(generators.data_ptr()), sh)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lib.bzamd_stage_timing_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stage_ms = (ctypes.c_double * 6)()
    calls = lib.bzamd_stage_timing_collect(stage_ms)

    # informational second leg (rank 0 of a single-GPU run): the same step with the generators
    # registered once as a resident set (bzamd_generators_*), i.e. without the
    # per-call conversion of caller generators.  Never used for `value`.
    resident_ms = None
    if world == 1:
        handle = lib.bzamd_generators_new_device(curve_id, ctypes.c_void_p(generators.data_ptr()), n, sh)
        for _ in range(args.warmup):
            lib.bzamd_msm_device_resident(ctypes.c_void_p(out.data_ptr()), 1, desc, handle, sh)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            lib.bzamd_msm_device_resident(ctypes.c_void_p(out.data_ptr()), 1, desc, handle, sh)
        torch.cuda.synchronize()
        resident_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        lib.bzamd_generators_free(handle)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        total_ops = world * n * args.steps
        result = {
            "metric": "MSM throughput (scalar-point ops/sec)",
            "value": total_ops / elapsed,
            "unit": "scalar-point ops/s",
            "commitments_per_s": world * args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (9 x 29-bit GF(2^255-19), v_mad_u64_u32 products)",
            "data": "synthetic: uniform random 252-bit scalars, built-in ristretto generators",
            "config": {"workload": name if args.log2n is None else f"curve25519_msm_n2^{log2n}_252bit",
                       "columns_per_gpu": 1, "rows": n, "parallelism": f"columns x{world}"},
        }
        if resident_ms is not None:
            result["resident_generators_ms_per_step"] = resident_ms
        if calls > 0:
            per_call = {STAGES[i]: stage_ms[i] / calls for i in range(6)}
            result["stage_ms"] = {k: round(v, 4) for k, v in per_call.items()}
            # dominant kernel: k_accumulate.  Algorithmic bytes per launch = SURVEY 8(d) per-unit
            # figure (scalar + generator bytes per scalar-point op) x ops per launch.
            alg_bytes = n * (nbytes + gen_bytes)
            dur_s = per_call["accumulate"] * 1e-3
            achieved = alg_bytes / dur_s / 1e9
            # HBM-side bytes per launch from the rocprofv3 PMC passes of the same command
            # (tools/prof/run_pmc.sh -> profiles/roofline_traffic.json); null until collected
            traffic = None
            valu_busy = None
            tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            if os.path.exists(tpath) and args.log2n is None:
                with open(tpath) as fh:
                    pmc = json.load(fh)
                traffic = pmc.get("k_accumulate_bytes_per_launch")
                valu_busy = pmc.get("valu_busy")
            result["roofline"] = {
                "kernel": "k_accumulate<ed25519>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": per_call["accumulate"],
                # the binding bound of this kernel is integer issue, not HBM:
                # fraction of cycles the SIMDs' VALU was issuing, from the same PMC run
                "valu_busy": valu_busy,
            }
            # one addition per non-zero digit: 252-bit scalars populate ceil(252 / 16) = 16 windows
            additions = n * ((8 * nbytes - (8 - top_mask.bit_length()) + 15) // 16)
            wave_mads = additions * MADS_PER_ADDITION / 64
            result["roofline"]["alu"] = {
                "instruction": "v_mad_u64_u32",
                "wave_instructions_per_launch": wave_mads,
                "achieved_per_s": wave_mads / dur_s,
                "peak_per_s": PEAK_WAVE_MADS_PER_S,
                "frac": wave_mads / dur_s / PEAK_WAVE_MADS_PER_S,
            }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.cpu_log2n)
            if cb is not None:
                result["cpu_baseline"] = cb
        print(json.dumps(result), flush=True)

    api.reset_for_testing()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
