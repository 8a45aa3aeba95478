#!/bin/bash
# kernel timeline of the sequence with and without the accumulate LDS pad (two resident wavefronts per SIMD)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
REPO=$(pwd); export TMPDIR=/tmp
for pad in 0 55000; do
  OUT=$REPO/gpurun_out/${1:-pad}/trace_$pad; rm -rf "$OUT"; mkdir -p "$OUT"
  (cd /tmp && BLITZAR_AMD_ACC_LDS_PAD=$pad timeout 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- \
     "$REPO/tools/pipeline_bench/_build/pipeline_bench" --steps 40 --warmup 10 > "$OUT/log.txt" 2>&1)
  f=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
  echo "== pad $pad"; tail -1 "$OUT/log.txt"
  python3 tools/prof/timeline.py "$f" 30 10 30
  find "$OUT" -name "*.csv" -delete
done
