#!/usr/bin/env python3
"""Hot-path benchmark: variable-base MSM / Pedersen commitments on MI355X through the C ABI.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: on every rank one
commitment column of 2^20 uniform 252-bit scalars -- the `std::mt19937{rank}` byte stream of the
reference benchmark (benchmark/multi_commitment/benchmark.m.cc:141-156), top nibble masked --
against 2^20 caller-supplied ristretto generators compute_base_element(i) (BASELINE.json
configs[1]).  Inputs are resident in HBM before the timed region.  With N ranks the job is N
independent columns (columns shard) followed by one RCCL all-gather of the N 32-byte
commitments -- weak scaling.

The LAST stdout line (rank 0) is ONE compact strict-JSON record (< 4 KB, `compact_record`):
metric = scalar-point ops / s over the whole job, plus
  roofline      dominant kernel k_accumulate: algorithmic bytes per launch / its HIP-event duration
                against 8 TB/s; `traffic` / `valu_busy` from the committed PMC passes; `alu_frac` =
                the integer-ALU side (the binding bound) against the v_mad_u64_u32 issue rate
                measured on THIS box in THIS run, `issue_frac` = the same for the loop's whole
                instruction mix (profiles/isa_counts.json); `box_*`: which kind of box ran, its sustained step
                and lone call -- flat scalars only
  cpu_baseline  N = 1 only: the reference's own CPU backend (oracle/_ref) on THE SAME scalars; the
                timed GPU commitment must equal its output or the bench aborts (`verified`)
The full record (every leg, nested) goes to gpurun_out/bench_detail.json (--detail-file).  With
--detail the untimed legs that only feed that file run too: BASELINE configs 1, 3, 4, 5 at their
stated shapes on one GPU (each with its roofline, a bounded cpu_baseline sample and a full-size
parity check), the 50 Hz power / clock trace, the host-buffer (PCIe-inclusive) figures.  N > 1:
config 4's 256 columns sharded over the ranks and config 2 row-split run by default.
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
sys.path.insert(0, os.path.join(ROOT, "tools"))
from blitzar_amd import api  # noqa: E402
import baseline_workloads as wl  # noqa: E402

START_WALL = time.time()
STAGES = ["prepare_addends", "recode", "bucket_sort", "accumulate", "reduce", "combine"]
ACC_KERNEL = {0: "k_accumulate<bz::ed25519_msm>", 1: "k_accumulate<bz::bls12_381_msm>",
              2: "k_accumulate<bz::bn254_msm>", 3: "k_accumulate<bz::grumpkin_msm>"}
ACCUMULATE_ONLY = 1 << 3  # stage mask of bzamd_stage_timing_begin_masked
HBM_PEAK_GBS = 8000.0
# integer-ALU side of k_accumulate (the honest binding bound).  v_mad_u64_u32 per
# bucket addition = the count in the kernel's ISA (field products x 99: 81 limb products, 16 that
# fold the eight high columns, as 64-bit sums, onto the low ones, 2 for the last carry).  Peak: the instruction issues once per 4 shader cycles per SIMD
# (profiles/round2_valu_rates.txt: 4.5 against the 2.4 of a plain VALU op), 1024 SIMDs; the clock is
# the effective shader clock the same micro-benchmark measures under an all-SIMD integer load.
# (per bucket addition: profiles/isa_counts.json, the count in the kernel's ISA -- curve25519: 8 field
# products x (81 limb products + 17 that fold the eight high columns onto the low ones + 1))
SIMDS = 1024
MAD_ISSUE_CYCLES = 4.0
EFFECTIVE_CLOCK_HZ = 2.1e9  # refined from profiles/alu_calibration.json when present


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=None, help="override rows (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the reference-CPU leg (and with it the output verification)")
    ap.add_argument("--detail", action="store_true",
                    help="also run the untimed legs that only feed the detail file: configs 1/3/4/5 at "
                         "full shape, the 50 Hz device-state trace, the host-buffer (PCIe) figures, the "
                         "one-process multi-device check (minutes; the default run is the headline, its "
                         "roofline and the reference-CPU baseline)")
    ap.add_argument("--detail-file", default=None,
                    help="where the full record goes (default gpurun_out/bench_detail.json beside "
                         "bench.py); the LAST stdout line is the compact record")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 1/3/4/5 legs")
    ap.add_argument("--skip-headline-check", action="store_true",
                    help="profiling runs: keep the oracle for the configs legs but skip the 16 s "
                         "reference-CPU run on the headline column (no `verified`, no cpu_baseline)")
    ap.add_argument("--config-steps", type=int, default=3)
    ap.add_argument("--config4-log2n", type=int, default=20,
                    help="rows of the sharded config-4 leg under --gpus N (debug / dry runs only: the "
                         "config is 2^20 rows)")
    ap.add_argument("--no-aux", action="store_true",
                    help="profiling runs: skip the device_state legs (3000 extra calls) and the "
                         "host_api child processes")
    ap.add_argument("--dry-run-one-gpu", action="store_true",
                    help="self-test of the N > 1 control flow on a one-GPU box: every rank uses "
                         "cuda:0 and the collectives go through gloo on host copies (never a "
                         "measurement)")
    return ap.parse_args()


class Collectives:
    """the three collectives of the bench: RCCL on device tensors, or (dry run) gloo on host copies"""

    def __init__(self, dist, host):
        self.dist = dist
        self.host = host

    def all_gather(self, dst, src):
        if not self.host:
            self.dist.all_gather_into_tensor(dst, src)
            return
        torch.cuda.synchronize()
        parts = [torch.empty_like(src, device="cpu") for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(parts, src.cpu())
        dst.copy_(torch.cat(parts, dim=0).to(dst.device))

    def all_reduce(self, t, op):
        if not self.host:
            self.dist.all_reduce(t, op=op)
            return
        c = t.cpu()
        self.dist.all_reduce(c, op=op)
        t.copy_(c.to(t.device))

    def broadcast(self, t, src):
        if not self.host:
            self.dist.broadcast(t, src)
            return
        c = t.cpu()
        self.dist.broadcast(c, src)
        t.copy_(c.to(t.device))

    def barrier(self):
        self.dist.barrier()


def vp(t):
    return ctypes.c_void_p(t.data_ptr())


def load_oracle():
    from oracle import ref_oracle  # the checker / CPU baseline; never the thing measured
    return ref_oracle if ref_oracle.available() else None


def alu_calibration():
    path = os.path.join(ROOT, "profiles", "alu_calibration.json")
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return {}


class StageClock:
    """HIP-event stage timing of the engine (bzamd_stage_timing_*), per call"""

    def __init__(self, lib, calls, mask=0x3f, sample_every=1):
        """`sample_every` > 1: one call in that many carries the event pairs (the others run
        without their stream bubbles); the per-call figures are means over the recorded calls"""
        self.lib = lib
        self.sample_every = sample_every
        if sample_every > 1:
            lib.bzamd_stage_timing_begin_sampled(calls, mask, sample_every)
        else:
            lib.bzamd_stage_timing_begin_masked(calls, mask)

    def collect(self, calls_expected):
        ms = (ctypes.c_double * 6)()
        batches = self.lib.bzamd_stage_timing_collect(ms)
        recorded = batches if self.sample_every > 1 else calls_expected
        per_call = {STAGES[i]: ms[i] / max(recorded, 1) for i in range(6)}
        return per_call, batches


def timed_calls(lib, fn, steps, warmup, stream):
    """seconds per call of a sequence of `steps` calls in the library's throughput mode (the last
    stage of a call beside the front of the next, bzamd_pipeline_next), flushed inside the timed
    region; stage times from the same calls"""
    for _ in range(max(warmup, 2)):  # in the same mode: the engine sizes its workspace for it
        lib.bzamd_pipeline_next()
        fn()
    lib.bzamd_pipeline_flush(stream)
    torch.cuda.synchronize()
    clock = StageClock(lib, steps * 64)
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.bzamd_pipeline_next()
        fn()
    lib.bzamd_pipeline_flush(stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    stages, _ = clock.collect(steps)
    return dt, stages


def lone_calls(fn, calls=3, lib=None):
    """ms of a lone call with plain stream semantics (no throughput mode): each call is followed by
    a device synchronisation; the minimum and the mean over `calls` calls after one untimed call.
    With `lib`: one more lone call under the engine's stage clock -- a lone call runs its six
    stages one after the other on the caller's stream, so these spans DO add up"""
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(calls):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
    out = {"min": min(ms), "mean": sum(ms) / len(ms), "calls": calls}
    if lib is not None:
        clock = StageClock(lib, 64)
        fn()
        torch.cuda.synchronize()
        stages, _ = clock.collect(1)
        out["stages"] = {k: round(v, 4) for k, v in stages.items()}
    return out


STAGE_NOTE = ("stage times of calls IN A SEQUENCE: HIP-event spans on the stream each stage runs on; "
              "the tail stages (reduce, combine) of a call run beside the next call at the lowest "
              "queue priority, so the spans overlap and do NOT add up to ms_per_call")


def profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        with open(path) as fh:
            return json.load(fh)
    return {}


IN_RUN_ALU = {}  # filled by probe_alu(): the v_mad_u64_u32 issue rate of THIS box, measured in this run


def probe_alu(lib, target_ms=50.0):
    """v_mad_u64_u32 issue rate of this device now (bzamd_probe_mad_rate: all SIMDs at 6 waves for
    ~50 ms at 6 waves per SIMD, the fastest launch of its second half); profiles/alu_calibration.json stays the fallback"""
    out = (ctypes.c_double * 4)()
    if lib.bzamd_probe_mad_rate(target_ms, out) != 0 or out[0] <= 0:
        return None
    return {"wave_instructions_per_s": out[0], "effective_clock_hz": out[1],
            "cycles_per_wave_instruction": out[2], "load_ms": out[3]}


def roofline_of(kernel, alg_bytes, accumulate_ms, additions=None, use_pmc=True):
    """HBM roofline of the dominant kernel (algorithmic bytes over its HIP-event duration), plus --
    from the committed rocprofv3 PMC passes of this command (profiles/roofline_traffic.json) and the
    ISA counts of the kernel's loop (profiles/isa_counts.json) -- the HBM-side traffic per launch,
    the fraction of cycles the SIMDs issued VALU work, and the integer-ALU side: v_mad_u64_u32
    wave-instructions per second against the measured issue rate of that instruction."""
    achieved = alg_bytes / (accumulate_ms * 1e-3) / 1e9 if accumulate_ms > 0 else 0.0
    roof = {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_call": alg_bytes, "kernel_ms_per_call": accumulate_ms}
    pmc_all = profile_json("roofline_traffic.json") if use_pmc else {}
    pmc = pmc_all.get("kernels", {}).get(kernel)
    if pmc:
        roof["traffic"] = pmc.get("bytes_per_launch")
        roof["traffic_source"] = pmc_all.get("source")
        roof["valu_busy"] = pmc.get("valu_busy")
        roof["valu_wave_instructions_per_launch"] = pmc.get("sq_insts_valu_per_launch")
    isa = profile_json("isa_counts.json").get("kernels", {}).get(kernel)
    if isa and additions and accumulate_ms > 0:
        mads = isa["mads_per_addition"]
        wave_mads = additions * mads / 64
        dur_s = accumulate_ms * 1e-3
        if IN_RUN_ALU.get("before"):
            # this box, this run: the mean of the probes right before the warm-up and right after
            # the timed region
            probes = [IN_RUN_ALU[k] for k in ("before", "after") if IN_RUN_ALU.get(k)]
            peak = sum(p["wave_instructions_per_s"] for p in probes) / len(probes)
            clock_hz = sum(p["effective_clock_hz"] for p in probes) / len(probes)
            issue = sum(p["cycles_per_wave_instruction"] for p in probes) / len(probes)
            peak_source = ("in-run: bzamd_probe_mad_rate on this device (all SIMDs at 6 waves, ~50 ms), "
                           "mean of the probes before the warm-up and after the timed region")
        else:
            cal = alu_calibration()
            clock_hz = cal.get("effective_clock_hz", EFFECTIVE_CLOCK_HZ)
            issue = cal.get("mad_u64_u32_cycles", MAD_ISSUE_CYCLES)
            peak = SIMDS * clock_hz / issue
            peak_source = "fallback: profiles/alu_calibration.json (another box, round 2)"
        roof["alu"] = {"instruction": "v_mad_u64_u32", "per_addition": mads,
                       "valu_instructions_per_addition_isa": isa["loop_valu"],
                       "bucket_additions_per_launch": additions,
                       "wave_instructions_per_launch": wave_mads,
                       "achieved_per_s": wave_mads / dur_s, "peak_per_s": peak,
                       "issue_cycles": issue, "effective_clock_hz": clock_hz,
                       "frac": wave_mads / dur_s / peak,
                       "ps_per_addition": dur_s / additions * 1e12,
                       "peak_source": peak_source,
                       "source": "instruction counts: profiles/isa_counts.json (tools/prof/isa_count.py)"}
        # the issue bound of the loop's OWN instruction mix (every VALU instruction at the measured
        # issue cost of its class, relative to the probed multiply-add): what the kernel could reach
        # with the arithmetic it has
        if isa.get("issue_cycles_per_addition") and isa.get("mad_issue_cycles_per_addition"):
            mix = isa["issue_cycles_per_addition"] / isa["mad_issue_cycles_per_addition"]
            roof["alu"]["issue_frac"] = roof["alu"]["frac"] * mix
            roof["alu"]["issue_cycles_per_wave_addition_isa"] = isa["issue_cycles_per_addition"]
            roof["issue_frac"] = roof["alu"]["issue_frac"]
        # the driver's record keeps the scalars of `roofline`, not its nested objects
        roof["alu_frac"] = roof["alu"]["frac"]
        roof["alu_peak_wave_mads_per_s"] = peak
        roof["alu_effective_clock_hz"] = clock_hz
        roof["alu_peak_in_run"] = bool(IN_RUN_ALU.get("before"))
        # (a kernel cannot issue more mads than the probe: if it seems to, the probe was disturbed)
        roof["alu_probe_suspect"] = bool(roof["alu"]["frac"] > 1.0)
        roof["alu_mads_per_addition"] = mads
        if pmc and pmc.get("sq_insts_valu_per_launch"):
            roof["alu"]["valu_instructions_per_addition_pmc"] = (
                pmc["sq_insts_valu_per_launch"] * 64 / additions)
    return roof


#--------------------------------------------------------------------------------------------------
# configs 1, 3, 4, 5 (N = 1)
#--------------------------------------------------------------------------------------------------
def config1(oracle):
    """curve25519, 1 column x 2^16 rows, SXT_CPU_BACKEND (plumbing, no GPU): a subprocess, the
    backend is chosen once per process"""
    import subprocess
    code = (
        "import sys, time, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tools')!r})\n"
        "from blitzar_amd import api\n"
        "import baseline_workloads as wl\n"
        "assert api.init(api.SXT_CPU_BACKEND, 1 << 16) == 0\n"
        "s = wl.mt19937_scalars(1, 1 << 16, 32)[0]\n"
        "api.compute_pedersen_commitments(0, [(s[:256], False)])\n"
        "t0 = time.perf_counter(); out = api.compute_pedersen_commitments(0, [(s, False)])\n"
        "print(json.dumps({'s': time.perf_counter() - t0, 'out': out[0].tolist()}))\n")
    def run(threads):
        env = dict(os.environ)
        if threads is not None:
            env["BLITZAR_AMD_HOST_THREADS"] = str(threads)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                           timeout=600, check=True, env=env)
        return json.loads(r.stdout.strip().splitlines()[-1])

    got = run(None)
    single = run(1)
    assert single["out"] == got["out"], "config 1: the thread count changed the commitment"
    n = 1 << 16
    entry = {"config": "1: curve25519 Pedersen commitment, 1 column x 2^16 rows, cpu backend",
             "backend": "SXT_CPU_BACKEND (host code of this library: the windows of a column on "
                        f"host threads, here at most {os.cpu_count()}; `single_thread_ms` = the "
                        "serial loop, what the reference's cpu backend is)",
             "ms_per_call": got["s"] * 1e3, "single_thread_ms": single["s"] * 1e3,
             "scalar_point_ops_per_s": n / got["s"],
             "data": "mt19937{0} bytes, 32-byte scalars, built-in generators",
             "roofline": None}
    if oracle is not None:
        s = wl.mt19937_scalars(1, n, 32)[0]
        gens = oracle.ristretto_generators(n)
        t0 = time.perf_counter()
        want = oracle.commit(0, [(s, False)], gens)
        dt = time.perf_counter() - t0
        assert want[0].tolist() == got["out"], "config 1: host backend disagrees with the reference"
        entry["verified"] = "bit-exact vs the reference CPU backend on the same scalars"
        entry["cpu_baseline"] = {"value": n / dt, "unit": "scalar-point ops/s", "cores": 1,
                                 "kind": "reference",
                                 "sample": f"the full config: 1 column x 2^16 rows, {dt:.2f} s"}
    return entry


def variable_base_config(lib, oracle, cid, name, log2n, columns, scalars, steps, dev, stream,
                         cpu_sample_log2n):
    n = 1 << log2n
    assert oracle is not None, "configs 3-5 build their generator sets from the reference's base point"
    base, gens = wl.dlog_generators(lib, oracle, cid, n, dev, stream)
    out = torch.zeros((columns, api.CURVE_LAYOUT[cid][1]), dtype=torch.uint8, device=dev)
    desc = (api.sxt_sequence_descriptor * columns)()
    for c in range(columns):
        desc[c] = api.sxt_sequence_descriptor(32, n, scalars[c].data_ptr(), 0)

    def step():
        lib.bzamd_msm_device(cid, vp(out), columns, desc, vp(gens), stream)

    dt, stages = timed_calls(lib, step, steps, 1, stream)
    got = out.cpu().numpy()
    lone = lone_calls(step, lib=lib)
    bad = []
    for c in range(columns):
        sums = wl.weighted_byte_sums(scalars[c])
        want = wl.expected_canonical(oracle, cid, base, wl.weighted_scalar_sum(sums, 0, 32))
        if not np.array_equal(got[c], want[:got.shape[1]]):
            bad.append(c)
    assert not bad, f"{name}: columns {bad} differ from the reference"
    ops = columns * n
    stride = api.CURVE_LAYOUT[cid][0]
    alg_bytes = ops * 32 + n * stride + columns * api.CURVE_LAYOUT[cid][1]
    entry = {"config": name, "rows": n, "columns": columns, "ms_per_call": dt * 1e3,
             "lone_call_ms": round(lone["min"], 4), "lone_call_ms_mean": round(lone["mean"], 4),
             "lone_call_stage_ms": lone["stages"],
             "scalar_point_ops_per_s": ops / dt, "commitments_per_s": columns / dt,
             "stage_ms_per_call": {k: round(v, 4) for k, v in stages.items()},
             "stage_ms_note": STAGE_NOTE,
             "verified": f"all {columns} outputs bit-exact vs (sum a_i (i+1) mod r) G computed and "
                         "encoded by the reference's curve code",
             # one bucket addition per non-zero digit: 252-bit scalars populate 16 windows of 16 bits
             "roofline": roofline_of(ACC_KERNEL[cid], alg_bytes, stages["accumulate"],
                                     additions=ops * 16)}
    # CPU sample: the reference backend on one column of 2^cpu_sample_log2n rows of this workload
    m = 1 << cpu_sample_log2n
    g_host = gens[:m].cpu().numpy()
    s_host = scalars[0][:m].cpu().numpy()
    t0 = time.perf_counter()
    want = oracle.commit(cid, [(s_host, False)], g_host)
    cdt = time.perf_counter() - t0
    sums = wl.weighted_byte_sums(scalars[0][:m])
    chk = wl.expected_canonical(oracle, cid, base, wl.weighted_scalar_sum(sums, 0, 32))
    assert np.array_equal(want[0], chk[:want.shape[1]])
    entry["cpu_baseline"] = {
        "value": m / cdt, "unit": "scalar-point ops/s", "cores": 1, "kind": "reference",
        "sample": f"1 column x 2^{cpu_sample_log2n} rows of this workload, {cdt:.2f} s on 1 core; "
                  "columns are independent and the reference's ops/s falls with n, so the full "
                  "config is at best this rate"}
    return entry


def config5(lib, oracle, steps, dev, stream, outputs=1024, log2n=18):
    cid, n = 3, 1 << log2n
    base, gens = wl.dlog_generators(lib, oracle, cid, n, dev, stream)
    gens_host = gens.cpu().numpy()
    t0 = time.perf_counter()
    handle = api.MultiexpHandle(cid, oracle.affine_to_projective(cid, gens_host))
    handle_s = time.perf_counter() - t0
    bit_table = wl.config5_bit_table(outputs)
    row_bytes = (int(bit_table.sum()) + 7) // 8
    # the mt19937{0} byte stream of the reference benchmarks, row-major over the packed rows
    # (3.3e9 draws: tools/mt19937 on all host threads)
    scalars = torch.from_numpy(wl.mt19937_bytes(n * row_bytes).reshape(n, row_bytes)).to(dev)
    offs = (np.concatenate([[0], np.cumsum(bit_table)[:-1]]) // 8).astype(np.int64)
    wide = torch.from_numpy(offs[bit_table == 256] + 31).to(dev)
    scalars[:, wide] &= 0x0f
    psize = api.CURVE_LAYOUT[cid][2]
    res = torch.zeros((outputs, psize), dtype=torch.uint8, device=dev)

    def step():
        lib.bzamd_fixed_packed_multiexponentiation_device(
            vp(res), handle._h, bit_table.ctypes.data_as(ctypes.c_void_p), None, outputs, n,
            vp(scalars), stream)

    dt, stages = timed_calls(lib, step, steps, 1, stream)
    got = res.cpu().numpy()
    lone = lone_calls(step, lib=lib)
    sums = wl.weighted_byte_sums(scalars)
    bad = []
    for k in range(outputs):
        want = wl.expected_canonical(oracle, cid, base, wl.weighted_scalar_sum(
            sums, int(offs[k]), int(bit_table[k]) // 8))
        have = np.ascontiguousarray(oracle.canonical(cid, got[k].view(np.uint64))).view(np.uint8)
        if not np.array_equal(have.reshape(-1), want):
            bad.append(k)
    assert not bad, f"config 5: outputs {bad} differ from the reference"
    handle.close()
    ops = outputs * n
    alg_bytes = n * row_bytes + outputs * psize
    entry = {"config": f"5: grumpkin packed fixed-base, {outputs} outputs x 2^{log2n} rows, "
                       "8/32/256-bit fields", "rows": n, "outputs": outputs,
             "bits_per_row": int(bit_table.sum()), "ms_per_call": dt * 1e3,
             "lone_call_ms": round(lone["min"], 4), "lone_call_ms_mean": round(lone["mean"], 4),
             "lone_call_stage_ms": lone["stages"],
             "row_output_ops_per_s": ops / dt, "outputs_per_s": outputs / dt,
             "handle_creation_s": handle_s,
             "data": "mt19937{0} bytes (tools/mt19937: the one serial stream produced on all host "
                     "threads by jump-ahead), the 256-bit fields masked to 252 bits; generators: "
                     "SURVEY 8(d)'s chain g_i = g_{i-1} + g_0 built by the reference's own code",
             "stage_ms_per_call": {k: round(v, 4) for k, v in stages.items()},
             "stage_ms_note": STAGE_NOTE,
             "verified": f"all {outputs} outputs bit-exact vs (sum a_i (i+1) mod r) G computed and "
                         "encoded by the reference's curve code",
             # bucket additions: one per signed 16-bit window of every field (1 / 3 / 16 for 8 / 32 /
             # 256 bits; the 256-bit fields hold 252-bit values) and row
             "roofline": roofline_of(ACC_KERNEL[cid], alg_bytes, stages["accumulate"],
                                     additions=n * int(sum((min(int(b), 252) + 1 + 15) // 16
                                                           for b in bit_table)))}
    # CPU sample: the reference's OWN fixed-base host path -- its partition-table
    # accessor and mtxpp2::multiexponentiate, compiled in place into oracle/_ref (round 4:
    # oracle/ref/ref_fixed_base.cc over host stand-ins for the CUDA runtime) -- at window width 8 (the
    # w = 16 table of the full config is 64 GiB) on a 2^14-row subsample of the first 96 outputs
    # (32 each of 8, 32 and 256 bits); its result is checked against the known-discrete-log values
    m, sub_outputs = 1 << 14, 96
    sub_bits = [int(bit_table[k]) for k in range(sub_outputs)]
    sub_rows = scalars[:m, :int(offs[sub_outputs])].cpu().numpy().copy()
    proj = oracle.affine_to_projective(cid, gens_host[:m])
    t0 = time.perf_counter()
    ref_handle = oracle.FixedHandle(cid, proj, 8)
    table_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    sub = ref_handle.packed_multiexponentiation(sub_bits, m, sub_rows)
    fdt = time.perf_counter() - t0
    ref_handle.close()
    sub_sums = wl.weighted_byte_sums(scalars[:m])
    for k in range(sub_outputs):
        want = wl.expected_canonical(oracle, cid, base, wl.weighted_scalar_sum(
            sub_sums, int(offs[k]), sub_bits[k] // 8))
        have = np.ascontiguousarray(oracle.canonical(cid, sub[k])).view(np.uint8).reshape(-1)
        assert np.array_equal(have, want), "config 5: the reference's fixed-base host path is off"
    entry["cpu_baseline"] = {
        "value": sub_outputs * m / fdt, "unit": "row-output ops/s", "cores": 1, "kind": "reference",
        "sample": f"{sub_outputs} outputs (32 each of 8, 32, 256 bits) x 2^14 rows through the "
                  f"reference's own mtxpp2::multiexponentiate (host path, compiled in place) over its "
                  f"partition table at window width 8 (built in {table_s:.1f} s by its "
                  f"make_in_memory_partition_table_accessor; the full config's default w = 16 table "
                  f"would be 64 GiB): {fdt:.1f} s on 1 core"}
    return entry


def run_configs(lib, oracle, args, dev, stream):
    entries = [config1(oracle)]
    if oracle is None:
        return entries
    s3 = torch.from_numpy(wl.mt19937_scalars(1, 1 << 22, 32, top_mask=0x0f)).to(dev)
    # (a single column in throughput mode: its last call's tails, 2.3 ms, drain inside the timed
    # region -- over at least 10 calls)
    e3 = variable_base_config(lib, oracle, 1, "3: bls12-381 G1 MSM, n = 2^22, 252-bit scalars", 22,
                              1, s3, max(args.config_steps, 10), dev, stream, 15)
    e3["data"] = ("mt19937{0} bytes, top nibble masked; generators: SURVEY 8(d)'s chain g_0 = "
                  "generate_random_element(rng{1, 2}), g_i = g_{i-1} + g_0, built by the reference's "
                  "own code")
    entries.append(e3)
    del s3
    # one mt19937{0} stream filling the 256 columns column-major (multi_commitment/benchmark.m.cc:
    # 141-156), 2^33 draws: tools/mt19937 on all host threads
    s4 = torch.from_numpy(wl.mt19937_scalars(256, 1 << 20, 32, top_mask=0x0f)).to(dev)
    e4 = variable_base_config(lib, oracle, 2, "4: bn254 G1 multi-commitment, 256 columns x 2^20 "
                              "rows (the whole config on ONE GPU)", 20, 256, s4, args.config_steps,
                              dev, stream, 15)
    e4["data"] = ("mt19937{0} bytes, column-major, top nibble masked (tools/mt19937: the one serial "
                  "stream on all host threads by jump-ahead); generators: SURVEY 8(d)'s chain "
                  "g_i = g_{i-1} + g_0 built by the reference's own code")
    entries.append(e4)
    del s4
    torch.cuda.empty_cache()
    entries.append(config5(lib, oracle, args.config_steps, dev, stream))
    return entries


def strong_scaling_config2(lib, args, dev, stream, rank, world, dist, coll, generators, n,
                           rank0_commitment):
    """N > 1: the HEADLINE column (mt19937{0}, 2^20 rows) cut into N row ranges (SURVEY 8(e) way 2):
    rank r commits rows [r n / N, (r + 1) n / N) against the matching generator slice to a projective
    partial (bzamd_msm_device_projective), ONE all-gather of the ranks' K x 160 bytes closes the
    sequence and every rank folds + encodes the K commitments (bzamd_fold_encode_device).  Group
    addition is exact, so the commitment must be rank 0's own commitment of the whole column, which
    the reference CPU backend verified.  Strong scaling: total work fixed as N grows.
    Reference: sxt/multiexp/curve/multiexponentiation.h:176-198 (chunks over devices, host combine)."""
    lo, hi = rank * n // world, (rank + 1) * n // world
    rows = hi - lo
    part = wl.mt19937_scalars(1, n, 32, top_mask=0x0f, seed=0)[0][lo:hi]
    scalars = torch.from_numpy(np.ascontiguousarray(part)).to(dev)
    gens = generators[lo:hi]
    steps = max(args.steps, 1)
    partials = torch.zeros((steps, 160), dtype=torch.uint8, device=dev)
    gathered = torch.zeros((world * steps, 160), dtype=torch.uint8, device=dev)
    commitments = torch.zeros((steps, 32), dtype=torch.uint8, device=dev)
    desc = (api.sxt_sequence_descriptor * 1)()
    desc[0] = api.sxt_sequence_descriptor(32, rows, scalars.data_ptr(), 0)

    def sequence(count):
        for k in range(count):
            lib.bzamd_pipeline_next()
            lib.bzamd_msm_device_projective(0, vp(partials[k:k + 1]), 1, desc, vp(gens), stream)
        lib.bzamd_pipeline_flush(stream)
        coll.all_gather(gathered, partials)
        # gathered = [rank][step][160]: exactly the partials[r * num_outputs + k] layout of the fold
        lib.bzamd_fold_encode_device(0, vp(commitments), vp(gathered), world, steps, stream)

    sequence(max(args.warmup, 2))
    torch.cuda.synchronize()
    coll.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sequence(steps)
    torch.cuda.synchronize()
    coll.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    coll.all_reduce(t, dist.ReduceOp.MAX)
    dt = float(t.item()) / steps
    got = commitments.cpu().numpy()
    want = torch.from_numpy(rank0_commitment.copy()).to(dev)
    coll.broadcast(want, 0)
    want = want.cpu().numpy()
    ok = bool((got == want[0]).all())
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    coll.all_reduce(flag, dist.ReduceOp.MIN)
    assert int(flag.item()) == 1, "row-split headline column: a rank's commitment differs"
    return {"config": f"2: curve25519 MSM, ONE column of 2^{n.bit_length() - 1} rows cut into {world} "
                      "row ranges (projective partials, all-gather, fold + encode on every rank)",
            "scaling": "strong", "rows_per_gpu": rows, "steps": steps, "ms_per_step": dt * 1e3,
            "scalar_point_ops_per_s": n / dt, "commitments_per_s": 1.0 / dt,
            "all_gather_bytes": int(gathered.numel()),
            "verified": "every rank's K commitments equal rank 0's commitment of the whole column "
                        "(itself bit-exact vs the reference CPU backend when the oracle is present)"}


def sharded_config4(lib, oracle, args, dev, stream, rank, world, dist, coll):
    """N > 1: config 4's 256 columns sharded over the ranks (strong scaling), one all-gather of
    the 72-byte commitments"""
    cid, n, columns = 2, 1 << args.config4_log2n, 256
    per = columns // world
    begin = rank * per
    # this rank's columns of the ONE mt19937{0} stream of the config (jump-ahead to column `begin`)
    scalars = torch.from_numpy(wl.mt19937_scalars(per, n, 32, top_mask=0x0f,
                                                  first_column=begin)).to(dev)
    base, gens = wl.dlog_generators(lib, oracle, cid, n, dev, stream)
    out = torch.zeros((per, 72), dtype=torch.uint8, device=dev)
    gathered = torch.zeros((world * per, 72), dtype=torch.uint8, device=dev)
    desc = (api.sxt_sequence_descriptor * per)()
    for c in range(per):
        desc[c] = api.sxt_sequence_descriptor(32, n, scalars[c].data_ptr(), 0)

    def step():
        lib.bzamd_msm_device(cid, vp(out), per, desc, vp(gens), stream)
        coll.all_gather(gathered, out)

    step()
    torch.cuda.synchronize()
    coll.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.config_steps):
        step()
    torch.cuda.synchronize()
    coll.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    coll.all_reduce(t, dist.ReduceOp.MAX)
    dt = float(t.item()) / args.config_steps
    # every rank checks its own shard inside the gathered result
    got = gathered.cpu().numpy()[begin:begin + per]
    ok = True
    for c in range(per):
        sums = wl.weighted_byte_sums(scalars[c])
        want = wl.expected_canonical(oracle, cid, base, wl.weighted_scalar_sum(sums, 0, 32))
        ok = ok and np.array_equal(got[c], want[:72])
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    coll.all_reduce(flag, dist.ReduceOp.MIN)
    assert int(flag.item()) == 1, "sharded config 4: a commitment differs from the reference"
    ops = world * per * n
    return {"config": f"4: bn254 G1, {world * per} columns x 2^{args.config4_log2n} rows sharded over {world} GPUs "
                      "(RCCL all-gather of the commitments)", "scaling": "strong",
            "columns_per_gpu": per, "ms_per_call": dt * 1e3, "scalar_point_ops_per_s": ops / dt,
            "commitments_per_s": world * per / dt,
            "verified": "every rank's commitments bit-exact vs the reference's curve code"}


def in_process_multi_device():
    import subprocess
    exe = os.path.join(ROOT, "tools", "pipeline_bench", "_build", "multi_device_check")
    if not os.path.exists(exe):
        return {"error": "tools/pipeline_bench/_build/multi_device_check is not built"}
    env = {k: v for k, v in os.environ.items()
           if k not in ("BLITZAR_AMD_NUM_DEVICES", "BLITZAR_AMD_FORCE_SHARDS")}
    try:
        r = subprocess.run([exe, "--log2n", "18", "--columns", "16", "--steps", "3"], env=env,
                           capture_output=True, text=True, timeout=240)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            return {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-600:]}
        out = json.loads(lines[-1])
        out["rc"] = r.returncode
        out["what"] = ("tools/pipeline_bench/multi_device_check.cc: one process, every visible "
                       "device; results compared with the same work on device 0 alone")
        return out
    except Exception as exc:  # a hang or crash there must not cost the bench line
        return {"error": repr(exc)[:300]}


class SmiTrace:
    """tools/prof/smi_trace.py in a process of its own: a >= 10 Hz (asked: 50 Hz) trace of socket
    power, the shader clock of every XCD and the part's energy accumulator through the amdsmi
    library, sliced afterwards by the wall-clock windows of the bench's legs"""

    def __init__(self):
        import subprocess
        import tempfile
        self.path = os.path.join(tempfile.gettempdir(), f"bzamd_smi_trace_{os.getpid()}.jsonl")
        self.proc = None
        self.windows = {}
        try:
            self.proc = subprocess.Popen(
                [sys.executable, os.path.join(ROOT, "tools", "prof", "smi_trace.py"), "--hz", "50",
                 "--out", self.path, "--max-seconds", "900"],
                stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None

    def window(self, name, t0, t1):
        self.windows[name] = (t0, t1)

    def finish(self):
        sys.path.insert(0, os.path.join(ROOT, "tools", "prof"))
        import smi_trace
        if self.proc is not None:
            try:
                self.proc.stdin.close()
                self.proc.wait(timeout=10)
            except Exception:
                self.proc.kill()
        static, samples, errors = smi_trace.load(self.path)
        try:
            os.remove(self.path)
        except OSError:
            pass
        out = {"static": static, "samples": len(samples)}
        if errors:
            out["errors"] = errors[:3]
        for name, (t0, t1) in self.windows.items():
            out[name] = smi_trace.summarize(samples, t0, t1)
        return out


def device_state(lib, sequence_call, lone_call, stream, trace):
    """which box this is and what the part does under the bench's two kinds of load: a sequence of
    the timed region's calls (every SIMD busy) and lone calls (40 % of a lone call are
    single-wavefront tails), each kept up for seconds and traced at 50 Hz by `trace` (SmiTrace);
    idle before.  Untimed, after everything that is measured.  The sequence leg is also the long-run
    figure of the step: `sustained_ms_per_step` over its last 2500 calls."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "prof"))
    import device_state as smi
    import threading
    out = {"static_cli": smi.static_info(),
           # the library's own probe (which of the pool's two kinds of boxes this is)
           "instruction_fetch_beyond_the_icache": {1: "half speed (the slower kind of box)",
                                                   0: "full speed (the faster kind of box)"}.get(
               lib.bzamd_slow_instruction_fetch(), "unknown")}
    torch.cuda.synchronize()
    t0 = time.time()
    time.sleep(0.5)
    trace.window("idle", t0, time.time())
    # ~3 s of calls in throughput mode, enqueued ahead of the device (0.1 ms of host per call)
    calls, skip = 3000, 500
    begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    for k in range(calls):
        if k == skip:
            begin.record()
        lib.bzamd_pipeline_next()
        sequence_call()
    lib.bzamd_pipeline_flush(stream)
    end.record()
    torch.cuda.synchronize()
    t1 = time.time()
    trace.window("under_sequence_load", t0 + 0.5, t1)  # past the clock ramp
    out["sequence_leg"] = {"calls": calls, "ms_per_step_last_2500": begin.elapsed_time(end) / (calls - skip)}
    stop = threading.Event()
    count = [0]

    def lone_loop():
        while not stop.is_set():
            lone_call()
            torch.cuda.synchronize()
            count[0] += 1

    t = threading.Thread(target=lone_loop)
    t0 = time.time()
    t.start()
    time.sleep(2.5)
    stop.set()
    t.join()
    t1 = time.time()
    trace.window("under_lone_call_load", t0 + 0.5, t1)
    out["lone_leg"] = {"calls": count[0], "ms_per_call_incl_host_sync": 1e3 * (t1 - t0) / max(count[0], 1)}
    out["how"] = ("amdsmi library (amdsmi_get_gpu_metrics_info) sampled at 50 Hz by tools/prof/"
                  "smi_trace.py in a process of its own; `energy_counter_mean_w` = the part's energy "
                  "accumulator over the window's wall time, independent of the sampling; sequence leg: "
                  f"{calls} calls of the timed shape in throughput mode, the first 0.5 s cut off; lone "
                  "leg: calls with a device synchronisation after each for 2.5 s, the first 0.5 s cut off")
    return out


def host_api(oracle=None, generators=None):
    """What a drop-in caller sees: the blocking sxt_* entry points with HOST buffers, PCIe-inclusive,
    in child processes (their own sxt_init, no torch): the native warm driver
    (tools/pipeline_bench/hostapi_bench.cc) and the clone of the reference's own benchmark CLI
    (tools/multi_commitment: `multi_commitment gpu 1048576 10 {1,10} 32`, whose mean includes the
    first, cold sample -- benchmark/multi_commitment/benchmark.m.cc:204-236).  Never `value`."""
    import subprocess
    out = {"what": "sxt_curve25519_compute_pedersen_commitments[_with_generators], host buffers, "
                   "2^20 rows x 32 bytes per column; every call uploads its scalars (and the caller's "
                   "160-byte generators) over PCIe and blocks until the commitments are in host memory"}
    exe = os.path.join(ROOT, "tools", "pipeline_bench", "_build", "hostapi_bench")
    try:
        r = subprocess.run([exe, "--samples", "10", "--warmup", "2"], capture_output=True, text=True,
                           timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["warm"] = json.loads(lines[-1]) if lines else {"error": f"rc {r.returncode}",
                                                           "stderr_tail": r.stderr[-400:]}
    except Exception as exc:  # a failure here must not cost the bench line
        out["warm"] = {"error": repr(exc)[:300]}
    cli = os.path.join(ROOT, "tools", "multi_commitment", "_build", "multi_commitment")

    def run_cli(n, columns, nbytes):
        entry = {"command": f"multi_commitment gpu {n} 10 {columns} {nbytes} 0"}
        try:
            r = subprocess.run([cli, "gpu", str(n), "10", str(columns), str(nbytes), "0"],
                               capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, BLITZAR_AMD_CLI_WARM="1"))
            for ln in r.stdout.splitlines():
                if ln.startswith("compute duration (s)"):
                    entry["mean_ms_incl_cold_first_sample"] = 1e3 * float(ln.split(":")[1])
                if ln.startswith("warm compute duration (s)"):
                    entry["warm_mean_ms"] = 1e3 * float(ln.split(":")[1])
                if ln.startswith("throughput (exponentiations / s)"):
                    entry["exponentiations_per_s"] = float(ln.split(":")[1])
            if "warm_mean_ms" in entry:
                entry["exponentiations_per_s_warm"] = n * columns / (entry["warm_mean_ms"] * 1e-3)
            entry["rc"] = r.returncode
        except Exception as exc:
            entry["error"] = repr(exc)[:300]
        return entry

    out["reference_cli_clone"] = [run_cli(1048576, columns, 32) for columns in (1, 10)]
    # the reference's own benchmark grid (benchmark/scripts/run_benchmarks.py:24-39): n x commitments x
    # element bytes, 10 samples each, through the same CLI clone; beside every (n, bytes) the reference
    # CPU backend on ONE column of that shape (columns are independent)
    grid = []
    cpu = {}
    for columns in (1, 10):
        for nbytes in (1, 32):
            for n in (10000, 100000, 1000000):
                e = run_cli(n, columns, nbytes)
                if oracle is not None and (n < 1000000 or nbytes == 1):
                    if (n, nbytes) not in cpu:
                        rng = np.random.default_rng(n + nbytes)
                        col = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
                        if nbytes == 32:
                            col[:, 31] &= 0x0f
                        gens = (generators[:n] if generators is not None and len(generators) >= n
                                else oracle.ristretto_generators(n))
                        t0 = time.perf_counter()
                        oracle.commit(0, [(col, False)], gens)
                        cpu[(n, nbytes)] = n / (time.perf_counter() - t0)
                    e["reference_cpu_ops_per_s_1_core"] = cpu[(n, nbytes)]
                grid.append(e)
    out["reference_benchmark_grid"] = grid
    out["reference_benchmark_grid_note"] = (
        "run_benchmarks.py's sweep: n in {1e4, 1e5, 1e6} x {1, 10} commitments x {1, 32} bytes; host "
        "buffers, caller generators uploaded per call; `warm_mean_ms` leaves the first sample out "
        "(BLITZAR_AMD_CLI_WARM, an extension of the clone); the reference CPU rate of 1e6 x 32 bytes is "
        "the headline's own cpu_baseline")
    return out


def box_record(roof, state, legs, lone_call_ms, lib):
    """`roofline.box` + the same as flat scalars (the driver's record keeps the scalars of `roofline`
    whole and drops nested objects): which box ran, which kind it is, its clocks and socket power
    under the sustained leg, the long-run step and the lone call"""
    tr = state.get("trace", {})
    seq = tr.get("under_sequence_load", {})
    lone = tr.get("under_lone_call_load", {})
    static = dict(tr.get("static", {}))
    if not static.get("asic_serial"):
        static.update(state.get("static_cli", {}))
    fetch = {1: "slow-fetch", 0: "fast-fetch"}.get(lib.bzamd_slow_instruction_fetch(), "unknown")
    box = {"asic_serial": static.get("asic_serial"),
           "fetch_kind": fetch,
           "sclk_mhz_under_sequence": seq.get("sclk_mhz"),
           "socket_power_w": seq.get("socket_power_w"),
           "socket_power_w_energy_counter": seq.get("energy_counter_mean_w"),
           "power_limited_share": seq.get("power_limited_share"),
           "trace_hz": seq.get("hz"), "trace_samples": seq.get("samples"),
           "sustained_ms_per_step": (state.get("sequence_leg", {}).get("ms_per_step_last_2500")
                                     or legs.get("sustained_ms")),
           "sustained_ms_per_step_before_warmup": legs.get("sustained_ms"),
           "lone_call_ms": lone_call_ms,
           "lone_leg": {"sclk_mhz": lone.get("sclk_mhz"), "socket_power_w": lone.get("socket_power_w"),
                        "socket_power_w_energy_counter": lone.get("energy_counter_mean_w")},
           "idle": tr.get("idle"),
           "power_cap_w": static.get("power_cap_w")}
    roof["box"] = box
    if box["asic_serial"] is not None:
        roof["box_asic_serial"] = str(box["asic_serial"])
    roof["box_fetch_kind"] = fetch
    roof["box_sustained_ms_per_step"] = box["sustained_ms_per_step"]
    roof["box_lone_call_ms"] = lone_call_ms
    if isinstance(seq.get("sclk_mhz"), dict):
        roof["box_sclk_mhz_under_sequence"] = seq["sclk_mhz"]["mean"]
    if isinstance(seq.get("socket_power_w"), dict):
        roof["box_power_w_min"] = seq["socket_power_w"]["min"]
        roof["box_power_w_mean"] = seq["socket_power_w"]["mean"]
        roof["box_power_w_max"] = seq["socket_power_w"]["max"]
    if seq.get("energy_counter_mean_w") is not None:
        roof["box_power_w_energy_counter"] = seq["energy_counter_mean_w"]
    for name in ("uclk_mhz", "socclk_mhz", "socclks_mhz_mean", "temperature_hotspot_c", "temperature_mem_c"):
        if seq.get(name) is not None:
            box[name + "_under_sequence"] = seq[name]
            roof["box_" + name + "_under_sequence"] = seq[name]
    if isinstance(lone.get("socket_power_w"), dict):
        roof["box_power_w_lone_calls_mean"] = lone["socket_power_w"]["mean"]
    roof["box_trace_hz"] = seq.get("hz")


#--------------------------------------------------------------------------------------------------
#--------------------------------------------------------------------------------------------------
# the record: ONE compact strict-JSON line, last on stdout; everything else in a sidecar file
#--------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                 "algorithmic_bytes_per_launch", "kernel_ms", "valu_busy", "alu_frac", "issue_frac")
BOX_KEYS = ("box_fetch_kind", "box_asic_serial", "box_sustained_ms_per_step", "box_lone_call_ms",
            "box_sclk_mhz_under_sequence", "box_power_w_mean")
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config")


def _finite(v):
    """strict JSON has no NaN / Infinity: they become null"""
    if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
        return None
    if isinstance(v, dict):
        return {k: _finite(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_finite(x) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return _finite(float(v))
    return v


def _round(v, digits=6):
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    return v


def compact_record(result):
    """the driver's line: the contract's members, `roofline` and `cpu_baseline` as flat objects of
    scalars, `verified`, and a handful of scalars beside them -- never a nested leg"""
    out = {k: result[k] for k in TOP_KEYS if k in result}
    out["dtype"] = "u32"  # the arithmetic type; the limb form is in DESIGN.md section 5
    out["data"] = "synthetic"
    roof = result.get("roofline")
    if roof:
        out["roofline"] = {k: _round(roof.get(k)) for k in ROOFLINE_KEYS}
        out["roofline"].update({k: _round(roof[k]) for k in BOX_KEYS if roof.get(k) is not None})
    cpu = result.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {"value": _round(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"],
                               "kind": cpu["kind"], "sample": cpu["sample"][:200]}
    out["verified"] = bool(result.get("verified"))
    for k in ("commitments_per_s", "single_call_ms", "sustained_ms_per_step", "effective_warmup_calls",
              "resident_generators_ms_per_step"):
        if result.get(k) is not None:
            out[k] = _round(result[k])
    if "stage_ms" in result:
        out["stage_ms"] = result["stage_ms"]
    if "configs" in result:
        out["configs_ms_per_call"] = {c["config"].split(":")[0]: _round(c["ms_per_call"], 4)
                                      for c in result["configs"] if "ms_per_call" in c}
    for name in ("strong_scaling", "strong_scaling_config2"):
        leg = result.get(name)
        if isinstance(leg, dict):
            out[name] = {k: _round(v) for k, v in leg.items()
                         if isinstance(v, (int, float, bool)) or (isinstance(v, str) and len(v) < 80)}
            if "verified" in leg:
                out[name]["verified"] = bool(leg["verified"])
    if "distributed" in result:
        d = result["distributed"]
        out["distributed"] = {k: d[k] for k in ("backend", "rccl_world_size", "distinct_devices",
                                                "distinct_commitments_gathered") if k in d}
    if result.get("detail_file"):
        out["detail_file"] = result["detail_file"]
    out = {k: _round(v) for k, v in _finite(out).items()}
    line = json.dumps(out, allow_nan=False, separators=(", ", ": "))
    # (whatever a later edit adds: the line stays under the limit; the optional members go first)
    for k in ("stage_ms", "strong_scaling_config2", "strong_scaling", "configs_ms_per_call",
              "distributed", "detail_file"):
        if len(line) < COMPACT_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, allow_nan=False, separators=(", ", ": "))
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(result, args):
    """full record -> the detail file (and stderr); compact record -> the LAST line on stdout"""
    path = args.detail_file
    if path is None:
        path = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(_finite(result), fh, indent=1, allow_nan=False)
            fh.write("\n")
        result["detail_file"] = os.path.relpath(path, ROOT)
    except OSError as exc:
        print(f"[bench] could not write {path}: {exc}", file=sys.stderr, flush=True)
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact_record(result), flush=True)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if args.dry_run_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    coll = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        # a rendezvous or a collective that never completes must end the run with an error, not hold
        # the node for the backend's default of ten minutes and more
        limit = datetime.timedelta(seconds=240)
        if args.dry_run_one_gpu:
            dist.init_process_group(backend="gloo", timeout=limit)
        else:
            # nccl == RCCL on ROCm
            dist.init_process_group(backend="nccl", device_id=dev, timeout=limit)
        coll = Collectives(dist, args.dry_run_one_gpu)
    # one process per GPU: the library's own multi-device sharding stays off, this process drives
    # the device torch selected
    os.environ["BLITZAR_AMD_NUM_DEVICES"] = "1"

    curve_id, nbytes, top_mask, gen_bytes = api.SXT_CURVE_RISTRETTO255, 32, 0x0f, 160
    log2n = 20 if args.log2n is None else args.log2n
    n = 1 << log2n
    name = f"curve25519_msm_n2^{log2n}_252bit"

    lib = api.load()
    assert api.init(api.SXT_GPU_BACKEND, 0) == 0
    # the untimed single-GPU legs that only feed the detail file run under --detail
    aux_legs = args.detail and not args.no_aux
    config_legs = args.detail and not args.no_configs
    trace = SmiTrace() if (rank == 0 and world == 1 and aux_legs) else None
    # a stream of the bench's own (callers on the NULL stream work too, the engine then uses plain
    # non-blocking tail streams: include/blitzar_amd.h, bzamd_pipeline_next)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    oracle = None if args.no_cpu_baseline else load_oracle()

    # synthetic inputs, resident in HBM: per-rank scalar column (mt19937{rank}), shared generators
    scalars_host = wl.mt19937_scalars(1, n, nbytes, top_mask=top_mask, seed=rank)[0]
    scalars = torch.from_numpy(scalars_host).to(dev)
    generators = torch.empty((n, gen_bytes), dtype=torch.uint8, device=dev)
    lib.bzamd_ristretto255_generators_device(vp(generators), 0, n, stream)
    out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    desc = (api.sxt_sequence_descriptor * 1)()
    desc[0] = api.sxt_sequence_descriptor(nbytes, n, scalars.data_ptr(), 0)
    torch.cuda.synchronize()

    # The steps run in the library's throughput mode (bzamd_pipeline_next, include/blitzar_amd.h):
    # the last stage of step k -- one workgroup walking the column's chain of doublings -- runs on
    # the engine's tail stream beside the generator conversion, recoding and sorting of step k + 1.
    # Every step writes its commitment to its own row of `outs`; a flush and -- with more than one
    # rank -- ONE all-gather of the ranks' K x 32 bytes close the sequence inside the timed region
    # (fewer, larger collectives: the commitments of a job are needed when the job is done).
    max_steps = max(args.steps, args.warmup, 1)
    outs = torch.zeros((max_steps, 32), dtype=torch.uint8, device=dev)
    gathered = torch.zeros((world * max_steps, 32), dtype=torch.uint8, device=dev)

    def step(k):
        lib.bzamd_pipeline_next()
        lib.bzamd_msm_device(curve_id, vp(outs[k:k + 1]), 1, desc, vp(generators), stream)

    def finish():
        lib.bzamd_pipeline_flush(stream)
        if world > 1:
            coll.all_gather(gathered, outs)

    # Untimed legs: the six stage times and the latency of a lone call with plain stream semantics (no
    # throughput mode), and the resident-generators leg (every rank runs it, rank 0 of a single-GPU
    # run reports it: the same step with the generators registered once as a resident set,
    # bzamd_generators_*; never used for `value`).
    stage_steps = min(args.steps, 50)
    legs = {}

    def lone_passes():
        clock = StageClock(lib, stage_steps)
        for _ in range(stage_steps):
            lib.bzamd_msm_device(curve_id, vp(out), 1, desc, vp(generators), stream)
        torch.cuda.synchronize()
        legs["per_call"], _ = clock.collect(stage_steps)
        t_single = time.perf_counter()
        for _ in range(stage_steps):
            lib.bzamd_msm_device(curve_id, vp(out), 1, desc, vp(generators), stream)
        torch.cuda.synchronize()
        legs["single_call_ms"] = 1e3 * (time.perf_counter() - t_single) / stage_steps
        legs["lone_output"] = out.cpu().numpy().copy()

    handle = lib.bzamd_generators_new_device(curve_id, vp(generators), n, stream)
    out2 = torch.zeros((1, 32), dtype=torch.uint8, device=dev)

    def resident_lone_pass():
        clock2 = StageClock(lib, stage_steps)
        for _ in range(stage_steps):
            lib.bzamd_msm_device_resident(vp(out2), 1, desc, handle, stream)
        torch.cuda.synchronize()
        legs["resident_stages"], _ = clock2.collect(stage_steps)

    def resident_sequence():
        for _ in range(max(args.warmup, 2)):
            lib.bzamd_pipeline_next()
            lib.bzamd_msm_device_resident(vp(out2), 1, desc, handle, stream)
        lib.bzamd_pipeline_flush(stream)
        torch.cuda.synchronize()
        resident_steps = max(args.steps, 50)
        clock2 = StageClock(lib, resident_steps, ACCUMULATE_ONLY)
        t1 = time.perf_counter()
        for _ in range(resident_steps):
            lib.bzamd_pipeline_next()
            lib.bzamd_msm_device_resident(vp(out2), 1, desc, handle, stream)
        lib.bzamd_pipeline_flush(stream)
        torch.cuda.synchronize()
        legs["resident_ms"] = 1e3 * (time.perf_counter() - t1) / resident_steps
        legs["untimed_calls_before_clock"] = (legs.get("untimed_calls_before_clock", 0)
                                              + max(args.warmup, 2) + resident_steps)
        legs["resident_acc"], _ = clock2.collect(resident_steps)
        legs["resident_output"] = out2.cpu().numpy().copy()

    def long_sequence():
        # the timed region's own step, >= 50 times in one sequence: the sustained rate, reported
        # beside the K-step figure (`sustained_ms_per_step`); its first calls size the workspace
        # of this call shape (hipMalloc: tens of ms of idle device)
        for k in range(2):
            step(0)
        finish()
        torch.cuda.synchronize()
        calls = max(args.steps, 50)
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def sustained():
            # one sequence; the clock (two events on the caller's stream) starts behind its 30th call:
            # the device comes out of an idle gap at reduced clocks
            for k in range(30 + calls):
                if k == 30:
                    begin.record()
                step(k % max_steps)
            finish()
            end.record()
            torch.cuda.synchronize()
            legs["untimed_calls_before_clock"] = legs.get("untimed_calls_before_clock", 0) + 30 + calls
            return begin.elapsed_time(end) / calls

        sustained()  # (the first sequence after the workspace was sized)
        legs["sustained_ms"] = sustained()
        legs["sustained_calls"] = calls
        legs["untimed_calls_before_clock"] = legs.get("untimed_calls_before_clock", 0) + 2

    # Order.  The device idles at ~100 MHz and comes back slowly: after ANY idle gap (20 ms are
    # enough) the first calls of a sequence take 1.14, 1.11, 1.06, 1.03, 1.00, 0.98 ms ... and the
    # sustained 0.96 only after ~30 ms of load (tools/prof/clock_course.py; sustained means
    # sustained: 3000 calls in sequence stay at 0.963).  A 20-step timed region is 20 ms long, so
    # entered from an idle or lightly loaded device it measures the ramp, not the engine.  Hence:
    # the legs that are long sequences in throughput mode run right before the W warmup steps --
    # the last of them with the timed region's own call shape, so that nothing is allocated between
    # it and the clock -- and the lone-call legs (low load: 40 % of a lone call is tails) after the
    # timed region.
    resident_sequence()
    IN_RUN_ALU["before"] = probe_alu(lib)
    long_sequence()

    for k in range(args.warmup):
        step(k)
    if args.warmup:
        finish()
    torch.cuda.synchronize()
    if world > 1:
        coll.barrier()
    torch.cuda.synchronize()
    # the timed region carries HIP events around the dominant kernel only (the roofline's live
    # duration), and around one launch in four: every recorded stage costs an event pair = two
    # stream bubbles per call (~1.5 % of a step when every call carries them); the other five
    # stages are measured by a separate, untimed pass below
    acc_sample = 4 if args.steps >= 8 else 1
    clock = StageClock(lib, args.steps, ACCUMULATE_ONLY, sample_every=acc_sample)
    wall0 = time.time()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    finish()
    torch.cuda.synchronize()
    if world > 1:
        coll.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if trace is not None:
        trace.window("timed_region", wall0, time.time())
    timed_stages, calls = clock.collect(args.steps)
    IN_RUN_ALU["after"] = probe_alu(lib)
    all_outputs = outs[:args.steps].cpu().numpy()
    timed_output = all_outputs[-1:].copy()
    assert (all_outputs == timed_output).all(), "the steps of the sequence disagree with each other"
    lone_passes()
    resident_lone_pass()
    per_call, single_call_ms, lone_output = legs["per_call"], legs["single_call_ms"], legs["lone_output"]
    resident_ms, resident_stages = legs["resident_ms"], legs["resident_stages"]
    resident_stages["accumulate"] = legs["resident_acc"]["accumulate"]
    per_call["accumulate"] = timed_stages["accumulate"]
    assert np.array_equal(lone_output, timed_output), "lone call disagrees with the sequence"
    assert np.array_equal(legs["resident_output"], timed_output), "resident path disagrees"
    lib.bzamd_generators_free(handle)  # (a hipFree: after the timed region)
    dist_info = None
    if world > 1:
        everyone = gathered.cpu().numpy().reshape(world, max_steps, 32)[:, :args.steps]
        assert np.array_equal(everyone[rank], all_outputs), "all-gather returned something else"
        # what the collective really spanned: every rank commits a DIFFERENT column (mt19937{rank}),
        # so the gathered buffer holds one distinct commitment per participating rank
        props = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(),
              "device_name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
              "commitment": bytes(timed_output[0]).hex()}
        peers = [None] * world
        dist.all_gather_object(peers, me)
        dist_info = {
            "backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""),
            "rccl_world_size": dist.get_world_size(),
            "ranks": peers,
            "all_gather_bytes": int(gathered.numel()),
            "all_gather_check": "every rank found its own K commitments in the gathered buffer; "
                                f"{len({bytes(everyone[r, -1]).hex() for r in range(world)})} "
                                f"distinct commitments from {world} ranks",
            "distinct_commitments_gathered": len({bytes(everyone[r, -1]).hex()
                                                  for r in range(world)}),
            "distinct_devices": len({(p["device_index"], p["pci_bus_id"]) for p in peers}),
        }
        assert args.dry_run_one_gpu or dist_info["distinct_commitments_gathered"] == world
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        coll.all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # every rank checks the commitment it timed against the reference CPU backend on the same
    # scalars (the ranks run their 16 s of oracle in parallel); rank 0 / N = 1 reports it as the
    # CPU baseline
    cpu = None
    verified = None
    headline_gens = None
    if oracle is not None and not args.skip_headline_check:
        gens_host = oracle.ristretto_generators(n)
        headline_gens = gens_host
        t2 = time.perf_counter()
        want = oracle.commit(0, [(scalars_host, False)], gens_host)
        cdt = time.perf_counter() - t2
        assert np.array_equal(want, timed_output), \
            f"rank {rank}: the timed GPU commitment differs from the reference CPU backend"
        verified = "timed output bit-exact vs the reference CPU backend on the same scalars"
        cpu = {"value": n / cdt, "unit": "scalar-point ops/s", "cores": 1, "kind": "reference",
               "sample": f"the full workload of one step: 1 column x 2^{log2n} rows of the same "
                         f"mt19937 scalars and generators, {cdt:.2f} s on 1 of {os.cpu_count()} "
                         "host cores (the reference cpu backend is single-threaded; N independent "
                         "processes scale it by the core count)",
               "all_cores_estimate": n / cdt * (os.cpu_count() or 1)}

    # N > 1: the BASELINE metric itself under strong scaling -- the headline column row-split over
    # the ranks -- beside the (trivially linear) weak-scaling `value`
    strong2 = None
    if world > 1:
        strong2 = strong_scaling_config2(lib, args, dev, stream, rank, world, dist, coll, generators,
                                         n, timed_output)
    sharded = None
    if world > 1 and oracle is not None and not args.no_configs and 256 % world == 0:
        sharded = sharded_config4(lib, oracle, args, dev, stream, rank, world, dist, coll)

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
            # calls of the timed shape (and of the resident-generators shape) issued right before
            # the W warmup steps to get the device out of its idle clocks (see "Order" above): the
            # timed region is a SUSTAINED-clock figure
            "effective_warmup_calls": legs["untimed_calls_before_clock"] + args.warmup,
            "ms_per_step": ms_per_step,
            "single_call_ms": single_call_ms,
            "sustained_ms_per_step": legs["sustained_ms"],
            "sustained_steps": legs["sustained_calls"],
            "mode": "throughput mode of the library (bzamd_pipeline_next / bzamd_pipeline_flush): the "
                    "last stage of step k (one workgroup per column) runs beside the front of step "
                    "k + 1; all K commitments are complete, and the last one verified, inside the "
                    "timed region; `single_call_ms` = a lone call with plain stream semantics; "
                    "`sustained_ms_per_step` = the same step over the last `sustained_steps` calls "
                    "of a longer sequence that runs right before the warmup steps (two events on "
                    "the caller's stream, flush included)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (9 x 29-bit GF(2^255-19), v_mad_u64_u32 products)",
            "data": "synthetic: std::mt19937{rank} bytes, uniform 252-bit scalars, generators "
                    "compute_base_element(i) supplied by the caller on every call"
                    + (" [DRY RUN: all ranks on one GPU, gloo collectives -- not a measurement]"
                       if args.dry_run_one_gpu else ""),
            "config": {"workload": name, "columns_per_gpu": 1, "rows": n,
                       "parallelism": f"columns x{world}"},
        }
        if verified:
            result["verified"] = verified
        if resident_ms is not None and world == 1:
            # the same column against a generator set registered once (bzamd_generators_*): Z = 1
            # addends and, for sets of 2^14 generators or more, window tables (2^(16 w) g_i
            # resident: one bucket set for all windows, no Horner chain)
            result["resident_generators_ms_per_step"] = resident_ms
            result["resident_generators_stage_ms"] = {k: round(v, 4)
                                                      for k, v in resident_stages.items()}
        if calls > 0:
            result["stage_ms"] = {k: round(v, 4) for k, v in per_call.items()}
            result["stage_ms_source"] = ("accumulate: HIP events inside the timed region; the other "
                                         f"stages: a separate untimed pass of {stage_steps} steps "
                                         "(every recorded stage costs two stream bubbles per call)")
            alg_bytes = n * (nbytes + gen_bytes)
            # one addition per non-zero digit: 252-bit scalars populate ceil(252 / 16) = 16 windows
            additions = n * ((8 * nbytes - (8 - top_mask.bit_length()) + 15) // 16)
            # (the PMC passes were collected at the default shape: 2^20 rows)
            roof = roofline_of(ACC_KERNEL[0], alg_bytes, per_call["accumulate"], additions=additions,
                               use_pmc=args.log2n is None)
            roof["algorithmic_bytes_per_launch"] = alg_bytes
            roof["kernel_ms"] = per_call["accumulate"]
            roof["kernel_ms_launches_timed"] = int(calls)  # (HIP events around one launch in four)
            roof["alu_probe"] = {k: v for k, v in IN_RUN_ALU.items() if v}
            result["roofline"] = roof
        if cpu is not None and world == 1:
            result["cpu_baseline"] = cpu
        if world == 1 and args.log2n is None and trace is not None:
            try:
                state = device_state(
                    lib, lambda: lib.bzamd_msm_device(curve_id, vp(outs[0:1]), 1, desc,
                                                      vp(generators), stream),
                    lambda: lib.bzamd_msm_device(curve_id, vp(out), 1, desc, vp(generators), stream),
                    stream, trace)
                state["trace"] = trace.finish()
                result["device_state"] = state
                if "roofline" in result:
                    box_record(result["roofline"], state, legs, single_call_ms, lib)
            except Exception as exc:  # never at the price of the line
                result["device_state"] = {"error": repr(exc)[:300]}
        elif "roofline" in result:
            # no trace (the default run): the box's kind, its sustained step and its lone call
            box_record(result["roofline"], {}, legs, single_call_ms, lib)
            del result["roofline"]["box"]
        if world == 1 and config_legs and args.log2n is None:
            result["configs"] = run_configs(lib, oracle, args, dev, stream)
        if sharded is not None:
            result["configs"] = [sharded]
            # the informative multi-GPU number: a FIXED job (config 4's 256 columns) cut over the ranks
            result["strong_scaling"] = {k: sharded[k] for k in
                                        ("config", "columns_per_gpu", "ms_per_call",
                                         "scalar_point_ops_per_s", "commitments_per_s", "verified")}
        if strong2 is not None:
            result["strong_scaling_config2"] = strong2
        if dist_info is not None:
            result["distributed"] = dist_info
        if args.dry_run_one_gpu:
            free_b, total_b = torch.cuda.mem_get_info(dev)
            result["dry_run"] = {"ranks_on_one_gpu": world, "device_bytes_in_use": int(total_b - free_b),
                                 "wall_s_since_start": time.time() - START_WALL}

    if trace is not None and trace.proc is not None and trace.proc.poll() is None:
        trace.finish()  # (device_state did not run: --log2n)
    api.reset_for_testing()
    if world > 1:
        coll.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # one PROCESS driving every visible device through the C ABI (bzamd_msm_multi_device: RCCL
        # all-gather inside the library; the sharded blocking sxt_* entry points), checked against
        # the same work on device 0 alone -- in a child process with a timeout, after this process
        # has released its GPU and the process group: a failure there cannot touch the line above
        # (more than one visible device only: a lone device would exchange with itself, and mapping
        # the 570 MB librccl for that costs up to a minute on a fresh box)
        if config_legs and not args.dry_run_one_gpu and torch.cuda.device_count() > 1:
            result["in_process_multi_device"] = in_process_multi_device()
        if world == 1 and config_legs and aux_legs and args.log2n is None:
            result["host_api"] = host_api(oracle, headline_gens)
        emit(result, args)


if __name__ == "__main__":
    main()
