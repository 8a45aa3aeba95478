#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box (run through gpurun from the repo root):
#   pass 0: --kernel-trace --stats            -> per-kernel durations
#   pass 1: --pmc SQ_* (8 slots) + GRBM       -> issue / stall breakdown
#   pass 2: --pmc FETCH_SIZE                  -> HBM read bytes (gfx950: x2 for wide streaming reads,
#   pass 3: --pmc WRITE_SIZE                     MI355X_MICROARCH.md section HBM)
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa tracing).
# usage: tools/prof/run_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --detail-file $OUT/bench_detail_pmc.json $*"
# the duration pass runs bench.py at its default length: the numbers then agree with BENCH_*.json
TRACE_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --detail-file $OUT/bench_detail_trace.json $*"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- python "$REPO/bench.py" $TRACE_ARGS > "$OUT/trace.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o r -- python "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o r -- python "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o r -- python "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
cd "$REPO"
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
python profiles/summarize_pmc.py "$OUT" "$OUT/roofline_traffic.json" > "$OUT/summary.md" 2> "$OUT/summary.err"
find "$OUT" -name "*.csv" | head -20
for f in $(find "$OUT" -name "*counter_collection.csv" | head -3); do echo "== $f"; head -3 "$f"; done
for f in $(find "$OUT" -name "*kernel_stats.csv" | head -1); do echo "== $f"; head -12 "$f"; done
du -sh "$OUT"
