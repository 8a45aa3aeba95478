#!/usr/bin/env python3
"""Summarise tools/prof/run_pmc.sh output (rocprofv3 csv) as markdown: per-kernel durations from
the --kernel-trace --stats pass and per-kernel averages of the PMC counters of the other passes.

    python profiles/summarize_pmc.py gpurun_out/prof_<tag> > profiles/<round>_<tag>.md
"""
import collections
import csv
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = name.replace("void ", "")
    m = re.match(r"(?:bz::)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9:]+)(<[^>]*>)?", name)
    s = (m.group(1) + (m.group(2) or "")) if m else name
    return s[:60]


def main():
    root = sys.argv[1]
    print(f"# rocprofv3 summary of `{root}`\n")
    stats = os.path.join(root, "trace", "r_kernel_stats.csv")
    if os.path.exists(stats):
        print("## --kernel-trace --stats\n\n| kernel | calls | avg_us | total_us | % |\n|---|---|---|---|---|")
        for r in csv.DictReader(open(stats)):
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                  f"{float(r['TotalDurationNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for sub in sorted(os.listdir(root)):
        f = os.path.join(root, sub, "r_counter_collection.csv")
        if not os.path.exists(f):
            continue
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per_dispatch[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"],
                       r["LDS_Block_Size"])
        for (k, _, c), v in per_dispatch.items():
            counters[k][c].append(v)
    if counters:
        names = sorted({c for k in counters for c in counters[k]})
        print("\n## PMC counters (average per dispatch)\n")
        print("| kernel | grid | wg | vgpr | sgpr | lds | " + " | ".join(names) + " |")
        print("|---|---|---|---|---|---|" + "---|" * len(names))
        for k in sorted(counters):
            if not k.startswith("k_"):
                continue
            vals = [f"{sum(counters[k][c]) / len(counters[k][c]):.4g}" if c in counters[k] else ""
                    for c in names]
            print(f"| {k} | " + " | ".join(meta[k]) + " | " + " | ".join(vals) + " |")
        def avg(k, c):
            v = counters[k].get(c)
            return sum(v) / len(v) if v else None

        # What FETCH_SIZE has to be multiplied by, per access pattern (profiles/fetch_calibration.json,
        # tools/ubench/gather_fetch.hip under rocprofv3 --pmc FETCH_SIZE, round 5): the counter tallies
        # a 128-byte request at 64 bytes -- wide streaming reads AND whole-line 128-byte row gathers
        # (curve25519 addends) read x 2 -- but 64- and 32-byte requests at face value: the 64-byte rows
        # of bn254 / grumpkin and the 96-byte rows of bls12-381 (64 + 32) read x 1.
        import json as _json
        cal_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fetch_calibration.json")
        patterns = _json.load(open(cal_path))["patterns"] if os.path.exists(cal_path) else {}

        def pattern_factor(row_bytes, default):
            # the table beyond the Infinity Cache: no L2 hits hiding requests from the counter
            e = patterns.get(f"k_gather<{row_bytes}, 31>")
            return round(e["factor"], 2) if e and e.get("factor") else default

        def read_factor(k):
            if k.startswith("k_accumulate<bz::bn254") or k.startswith("k_accumulate<bz::grumpkin"):
                return pattern_factor(64, 1.0)
            if k.startswith("k_accumulate<bz::bls12_381"):
                return pattern_factor(96, 1.0)
            if k.startswith("k_accumulate<bz::ed25519"):
                return pattern_factor(128, 2.0)
            return 2.0

        def kernel_entry(k):
            f, w = avg(k, "FETCH_SIZE"), avg(k, "WRITE_SIZE")
            e = {"fetch_kib": f, "write_kib": w, "launches_averaged": len(counters[k].get("FETCH_SIZE", []))}
            if f is not None and w is not None:
                e["fetch_factor"] = read_factor(k)
                e["bytes_per_launch"] = (e["fetch_factor"] * f + w) * 1024
                e["raw_bytes_per_launch"] = (f + w) * 1024
            act, cyc = avg(k, "SQ_ACTIVE_INST_VALU"), avg(k, "GRBM_GUI_ACTIVE")
            if act is not None and cyc:
                # quad-cycles of VALU issue summed over 1024 SIMDs / (cycles per XCD x SIMDs);
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                e["valu_busy"] = 4 * act / (1024 * cyc / 8)
            for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
                      "SQ_WAIT_ANY", "SQ_INST_CYCLES_VMEM", "GRBM_GUI_ACTIVE"):
                if avg(k, c) is not None:
                    e[c.lower() + "_per_launch"] = avg(k, c)
            return e

        acc = sorted(k for k in counters if k.startswith("k_accumulate"))
        head = [k for k in acc if k == "k_accumulate<bz::ed25519_msm>"] or acc
        if head and "FETCH_SIZE" in counters[head[0]] and "WRITE_SIZE" in counters[head[0]]:
            import json
            h = kernel_entry(head[0])
            # gfx950 correction of the guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies a
            # 128-byte request of 16-byte-per-lane loads as 64 bytes -> x 2.  Calibrated in the
            # same run on a kernel with known bytes: k_prepare_addends reads n x 160 B and writes
            # n x 128 B (n = 2^20: 167.8 MB / 134.2 MB).
            cal = {}
            prep = [k for k in counters if k.startswith("k_prepare_addends_staged<bz::ed25519_msm")]
            if prep and "FETCH_SIZE" in counters[prep[0]]:
                pf, pw = avg(prep[0], "FETCH_SIZE"), avg(prep[0], "WRITE_SIZE")
                cal = {"kernel": prep[0], "known_read_bytes": (1 << 20) * 160,
                       "fetch_size_bytes_raw": pf * 1024, "known_write_bytes": (1 << 20) * 128,
                       "write_size_bytes_raw": pw * 1024,
                       "read_factor": (1 << 20) * 160 / (pf * 1024),
                       "write_factor": (1 << 20) * 128 / (pw * 1024)}
            out = {"kernel": head[0], "fetch_kib": h["fetch_kib"], "write_kib": h["write_kib"],
                   "valu_busy": h.get("valu_busy"),
                   "k_accumulate_bytes_per_launch": h["bytes_per_launch"],
                   "raw_bytes_per_launch": h["raw_bytes_per_launch"],
                   "calibration": cal,
                   # every k_accumulate instantiation of the run (bench.py's configs read theirs)
                   "kernels": {k: kernel_entry(k) for k in acc},
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                             "tools/prof/run_pmc.sh) of `python bench.py` on MI355X, summarised "
                             "by profiles/summarize_pmc.py from " + root,
                   "note": "HBM-side bytes per launch = fetch_factor x FETCH_SIZE + WRITE_SIZE (KiB x "
                           "1024); fetch_factor per access pattern from profiles/fetch_calibration.json "
                           "(128-byte requests are tallied at 64 bytes: x 2 for streaming reads and "
                           "128-byte row gathers; 64- and 96-byte row gathers x 1), cross-checked in the "
                           "same run on k_prepare_addends whose bytes are known (see "
                           "`calibration`).  The bucket method gathers every 128-byte addend once "
                           "per window: 16 x 2^20 x 128 B = 2.15 GB plus 0.2 GB of indices and "
                           "bucket writes against 0.20 GB algorithmic; the addend table (128 MiB) "
                           "sits in the 256 MiB Infinity Cache, whose hits these counters include"}
            if len(sys.argv) > 2:
                with open(sys.argv[2], "w") as fh:
                    json.dump(out, fh, indent=1)
        print("\nFETCH_SIZE / WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE counts wide "
              "streaming reads at half their bytes (MI355X_MICROARCH.md, HBM section).")


if __name__ == "__main__":
    main()
