#!/bin/bash
# rocprofv3 passes for bench.py INCLUDING the configs 3 / 4 / 5 legs (the Weierstrass kernels) on the
# GPU box (run through gpurun from the repo root):
#   pass 0: --kernel-trace --stats            -> per-kernel durations
#   pass 1: --pmc SQ_* (8 slots) + GRBM       -> issue / stall breakdown
#   pass 2: --pmc FETCH_SIZE                  -> HBM read bytes (gfx950: x2 for wide streaming reads,
#   pass 3: --pmc WRITE_SIZE                     MI355X_MICROARCH.md section HBM)
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa tracing).
# The headline's 16 s reference-CPU check is skipped (--skip-headline-check); the configs keep
# their own full-size output checks.
# (--detail: the configs legs only run under it since round 6; the headline-only passes of the
# driver's own command are tools/prof/run_pmc.sh)
# usage: tools/prof/run_pmc_configs.sh <tag> [bench args...]
set -u
TAG=${1:-configs}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --config-steps 1 --skip-headline-check --detail --no-aux --detail-file $OUT/bench_detail_pmc.json $*"
TRACE_ARGS="--steps 20 --warmup 5 --config-steps 3 --skip-headline-check --detail --no-aux --detail-file $OUT/bench_detail_trace.json $*"
timeout -k 10 330 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- python "$REPO/bench.py" $TRACE_ARGS > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
timeout -k 10 330 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o r -- python "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
timeout -k 10 330 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o r -- python "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
timeout -k 10 330 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o r -- python "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
cd "$REPO"
# keep what travels back small: the per-dispatch counter csv and the stats, not the kernel traces
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
python profiles/summarize_pmc.py "$OUT" "$OUT/roofline_traffic.json" > "$OUT/summary.md" 2> "$OUT/summary.err"
head -40 "$OUT/summary.md"
du -sh "$OUT"
