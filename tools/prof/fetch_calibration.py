#!/usr/bin/env python3
"""FETCH_SIZE calibration on the gather pattern (tools/ubench/gather_fetch.hip):

    python tools/prof/fetch_calibration.py <rocprofv3 --pmc FETCH_SIZE output dir> <ubench stdout> \
        [--out profiles/fetch_calibration.json]

factor = known bytes of a launch / (FETCH_SIZE of the launch x 1024): what a kernel's FETCH_SIZE has
to be multiplied by to read as bytes, per access pattern.  profiles/summarize_pmc.py applies the
128-/96-/64-byte gather factors to k_accumulate<curve> and the streaming factor to everything else."""
import collections
import csv
import json
import os
import sys

csv.field_size_limit(1 << 30)


def main():
    root, ubench = sys.argv[1], sys.argv[2]
    out = "profiles/fetch_calibration.json"
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    known = {e["kernel"]: e for e in json.load(open(ubench))}
    per_dispatch = collections.defaultdict(float)
    for dirpath, _, files in os.walk(root):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(path)):
                    if r["Counter_Name"] != "FETCH_SIZE":
                        continue
                    name = r["Kernel_Name"].replace("void ", "").split("(")[0].strip()
                    per_dispatch[(name, r["Dispatch_Id"])] += float(r["Counter_Value"])
            elif f.endswith(".db"):  # rocprofv3's default output: a rocpd sqlite database
                import sqlite3
                db = sqlite3.connect(path)
                for name, dispatch, value in db.execute(
                        "select kernel_name, dispatch_id, value from counters_collection "
                        "where counter_name = 'FETCH_SIZE'"):
                    name = name.replace("void ", "").split("(")[0].strip()
                    per_dispatch[(name, dispatch)] += float(value)
    by_kernel = collections.defaultdict(list)
    for (name, _), v in per_dispatch.items():
        by_kernel[name].append(v)
    result = {"source": f"rocprofv3 --pmc FETCH_SIZE of tools/ubench/bin/gather_fetch ({root})",
              "unit": "factor = known bytes / (FETCH_SIZE x 1024)", "patterns": {}}
    for name, e in known.items():
        vals = by_kernel.get(name)
        if not vals:
            continue
        # the first launch of a gather pattern warms the caches: drop it where there are more
        use = vals[1:] if len(vals) > 1 else vals
        fetch_bytes = sum(use) / len(use) * 1024
        entry = dict(e)
        entry["fetch_size_bytes_per_launch"] = fetch_bytes
        entry["factor"] = e["known_bytes_per_launch"] / fetch_bytes if fetch_bytes else None
        result["patterns"][name] = entry
    json.dump(result, open(out, "w"), indent=1)
    print(json.dumps({k: round(v["factor"], 3) for k, v in result["patterns"].items() if v["factor"]}))


if __name__ == "__main__":
    main()
