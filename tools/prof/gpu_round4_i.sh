#!/bin/bash
# round 4, ninth GPU session: what the top window's streamed groups cost pass 2 (masked 252-bit against
# full 256-bit scalars), wall time of the bench line, PMC of the tail kernels on this box's kind
set -u
OUT=gpurun_out/r4i
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
tools/ubench/bin/tail_latency 2>&1 | tail -12 | head -4
AB=tools/prof/ab_pipeline.sh
PB=tools/pipeline_bench/_build/pipeline_bench
for rep in 1 2; do
  echo "== masked (252-bit scalars)"; $PB --steps 200 | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//'
  echo "== --no-mask (256-bit scalars)"; $PB --steps 200 --no-mask | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//'
done 2>&1 | tee $OUT/ab_top_window.log
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_mask -o t -- $REPO/$PB --steps 12 --warmup 3 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_nomask -o t -- $REPO/$PB --steps 12 --warmup 3 --no-mask > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_INST_LDS --output-format csv -d $REPO/$OUT/pmc_tails -o r -- $REPO/$PB --steps 6 --warmup 2 > $REPO/$OUT/pmc_tails.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, collections
for tag in ("trace_mask", "trace_nomask"):
    d = collections.defaultdict(list)
    for f in glob.glob(f"{sys.argv[1]}/{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].replace("void ", "").replace("bz::", "")[:36]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(tag, {k: round(sorted(v)[len(v) // 4], 1) for k, v in d.items() if k.startswith("k_group") or k.startswith("k_recode")}, "(lower-quartile us: the lone calls)")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_tails/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].replace("void ", "").replace("bz::", "")[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
cs = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_IFETCH", "SQ_WAIT_INST_LDS")
print(f"{'kernel':32s} " + " ".join(f"{c[3:]:>15s}" for c in cs))
for name, c in sorted(rows.items()):
    if name.startswith(("k_reduce", "k_horner", "k_accumulate")):
        print(f"{name:32s} " + " ".join(f"{sum(c[x]) / max(len(c[x]), 1):15.4e}" if x in c else f"{'-':>15s}" for x in cs))
PY
rm -rf $OUT/trace_mask $OUT/trace_nomask $OUT/pmc_tails
/usr/bin/time -v python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; grep -E "Elapsed|Maximum resident" $OUT/bench.err
