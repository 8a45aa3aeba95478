#!/bin/bash
# one gpurun call: micro-benchmarks, GPU test suite, bench.py (run from the repo root on the box)
# usage: tools/prof/gpu_round.sh <tag> [pytest args...]
set -u
TAG=${1:-r}; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout 120 tools/ubench/_build/valu_rates > "$OUT/valu_rates.txt" 2>&1
timeout 120 tools/ubench/_build/field_costs > "$OUT/field_costs.txt" 2>&1
timeout 900 python -m pytest tests -m gpu -q --durations=20 "$@" > "$OUT/pytest_gpu.txt" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gpu.txt"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" >> "$OUT/bench.err"
tail -5 "$OUT/pytest_gpu.txt"; tail -3 "$OUT/bench.err"; head -c 600 "$OUT/bench.json"
