#!/bin/bash
# round 4, last GPU session: rocprofv3 kernel statistics of LONE full-width calls (2^22 rows, one
# column, the four curves): the tails as they run when nothing runs beside them
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python $REPO/tools/prof/horner_phases.py --log2n 22 --full-width-only > $OUT/run.txt 2>&1
echo "rc=$?"; grep "lone call" $OUT/run.txt
python - $OUT/trace <<'PY'
import csv, glob, os, sys
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
print("| kernel | calls | avg_us | min_us | max_us |")
print("|---|---|---|---|---|")
for r in rows:
    n = r["Name"].replace("void ", "")
    if any(k in n for k in ("k_horner", "k_reduce", "k_accumulate", "k_group", "k_recode", "k_prepare")):
        print(f"| {n[:60]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} |")
PY
find $OUT/trace -name "*kernel_trace.csv" -size +2M -delete
