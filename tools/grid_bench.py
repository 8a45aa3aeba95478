#!/usr/bin/env python3
"""The regimes the reference's users run, measured on one MI355X:

  A  the reference's own benchmark grid (benchmark/scripts/run_benchmarks.py:24-39): n in {1e4, 1e5,
     1e6} x {1, 10} commitments x {1, 32} bytes, through the clone of its CLI (tools/multi_commitment:
     the drop-in sxt_* entry point, HOST buffers, caller generators uploaded on every call, 10
     samples, the mean including the cold first one as the reference reports it + the warm mean);
  B  the bucket_method2 regime (sxt/multiexp/bucket_method2/multiexponentiation.h:48-121: 256 <= n <=
     4096, many outputs): n in {256, 1024, 4096} x 1024 columns, curve25519 and bn254,
     device-resident operands (tools/pipeline_bench: in sequence and lone);
  C  short single columns, 2^12 .. 2^18 rows, device-resident.

Beside every point: the reference CPU backend (oracle/_ref, one core) on a bounded sample of the
same shape.  Writes one JSON document and prints a markdown table.

    python tools/grid_bench.py --out gpurun_out/<tag>/grid.json [--skip-cpu]
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
CLI = os.path.join(ROOT, "tools", "multi_commitment", "_build", "multi_commitment")
PIPE = os.path.join(ROOT, "tools", "pipeline_bench", "_build", "pipeline_bench")


def run_cli(n, commitments, nbytes):
    env = dict(os.environ, BLITZAR_AMD_CLI_WARM="1")
    r = subprocess.run([CLI, "gpu", str(n), "10", str(commitments), str(nbytes), "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    out = {"n": n, "commitments": commitments, "element_nbytes": nbytes, "rc": r.returncode}
    for ln in r.stdout.splitlines():
        if ln.startswith("compute duration (s)"):
            out["mean_ms_incl_cold"] = 1e3 * float(ln.split(":")[1])
        if ln.startswith("warm compute duration (s)"):
            out["warm_mean_ms"] = 1e3 * float(ln.split(":")[1])
        if ln.startswith("throughput (exponentiations / s)"):
            out["exponentiations_per_s_incl_cold"] = float(ln.split(":")[1])
    if "warm_mean_ms" in out:
        out["exponentiations_per_s_warm"] = n * commitments / (out["warm_mean_ms"] * 1e-3)
    return out


def run_pipe(curve, log2n, columns, steps, nbytes=32):
    r = subprocess.run([PIPE, "--curve", str(curve), "--log2n", str(log2n), "--columns", str(columns),
                        "--steps", str(steps), "--warmup", "3", "--nbytes", str(nbytes)],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": f"rc {r.returncode}", "stderr": r.stderr[-300:]}
    d = json.loads(lines[-1])
    ops = columns << log2n
    return {"curve": curve, "rows": 1 << log2n, "columns": columns, "element_nbytes": nbytes,
            "ms_in_sequence": d["ms_per_step"], "ms_lone": d["lone_ms"],
            "ops_per_s_in_sequence": ops / (d["ms_per_step"] * 1e-3),
            "ops_per_s_lone": ops / (d["lone_ms"] * 1e-3),
            "lone_stage_ms": d["lone_stage_ms"], "outputs_agree": d["outputs_agree"]}


def cpu_rate(oracle, cid, gens, n, nbytes, columns, budget_s=6.0):
    """reference CPU backend, one core: scalar-point ops/s on `columns` columns of n rows (bounded)"""
    rng = np.random.default_rng(n + nbytes)
    cols = [(rng.integers(0, 256, (n, nbytes), dtype=np.uint8), False) for _ in range(columns)]
    if nbytes == 32:
        for c, _ in cols:
            c[:, 31] &= 0x0f
    t0 = time.perf_counter()
    oracle.commit(cid, cols, gens[:n])
    dt = time.perf_counter() - t0
    return {"ops_per_s": columns * n / dt, "sample": f"{columns} column(s) x {n} rows, {dt:.2f} s on 1 core"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--skip-cpu", action="store_true")
    args = ap.parse_args()
    doc = {"reference_grid": [], "short_many_columns": [], "short_single_columns": []}
    oracle = None
    if not args.skip_cpu:
        from oracle import ref_oracle
        oracle = ref_oracle if ref_oracle.available() else None
    gens25519 = oracle.ristretto_generators(1000000) if oracle else None

    # A: the reference's grid, in its own order
    cpu_cache = {}
    for commitments in (1, 10):
        for nbytes in (1, 32):
            for n in (10000, 100000, 1000000):
                e = run_cli(n, commitments, nbytes)
                if oracle is not None:
                    # columns are independent and identical in shape: one column is the sample
                    if (n, nbytes) not in cpu_cache:
                        cpu_cache[(n, nbytes)] = cpu_rate(oracle, 0, gens25519, n, nbytes, 1)
                    e["reference_cpu"] = cpu_cache[(n, nbytes)]
                doc["reference_grid"].append(e)
                print(json.dumps(e), flush=True)

    # B: 256 <= n <= 4096 x 1024 columns, device-resident
    bn_gens = None
    if oracle is not None:
        from tests import util
        bn_gens = util.generators_for(2, 4096)
    for curve in (0, 2):
        for log2n in (8, 10, 12):
            e = run_pipe(curve, log2n, 1024, 30)
            if oracle is not None and "error" not in e:
                g = gens25519 if curve == 0 else bn_gens
                e["reference_cpu"] = cpu_rate(oracle, curve, g, 1 << log2n, 32, 16 if log2n < 12 else 4)
            doc["short_many_columns"].append(e)
            print(json.dumps(e), flush=True)

    # C: short single columns
    for log2n in (12, 14, 16, 18):
        e = run_pipe(0, log2n, 1, 200)
        doc["short_single_columns"].append(e)
        print(json.dumps(e), flush=True)

    with open(args.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("\n| n | commitments | bytes | ms (mean of 10 incl. cold) | ms warm | exp/s warm | reference cpu, 1 core |")
    print("|---|---|---|---|---|---|---|")
    for e in doc["reference_grid"]:
        cpu = e.get("reference_cpu", {}).get("ops_per_s", 0.0)
        print(f"| {e['n']} | {e['commitments']} | {e['element_nbytes']} | {e.get('mean_ms_incl_cold', 0):.3f} | "
              f"{e.get('warm_mean_ms', 0):.3f} | {e.get('exponentiations_per_s_warm', 0):.3g} | {cpu:.3g} |")
    print("\n| curve | rows | columns | ms in sequence | ms lone | ops/s lone | reference cpu, 1 core |")
    print("|---|---|---|---|---|---|---|")
    for e in doc["short_many_columns"] + doc["short_single_columns"]:
        if "error" in e:
            continue
        cpu = e.get("reference_cpu", {}).get("ops_per_s", 0.0)
        print(f"| {e['curve']} | {e['rows']} | {e['columns']} | {e['ms_in_sequence']:.4f} | {e['ms_lone']:.4f} | "
              f"{e['ops_per_s_lone']:.3g} | {cpu:.3g} |")


if __name__ == "__main__":
    main()
