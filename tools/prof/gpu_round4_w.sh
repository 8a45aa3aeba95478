#!/bin/bash
# instruction-cache counters of the lone tails (k_reduce, k_horner) on whichever kind of box this is
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4w/pmc_$(date +%s)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc ${PMC_SET:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU} --output-format csv -d $OUT/pmc -o r -- python $REPO/tools/prof/horner_phases.py --log2n 16 --full-width-only > $OUT/run.txt 2>&1
echo "rc=$?"; grep "slow_instruction_fetch\|lone call" $OUT/run.txt; tail -3 $OUT/run.txt | cut -c1-300
python - $OUT/pmc <<'PY'
import csv, glob, os, sys, collections
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("bz::", "")
    if "k_horner" in n or "k_reduce" in n:
        acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
PY
