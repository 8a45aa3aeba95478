#!/bin/bash
set -u
OUT=gpurun_out/r3c15
mkdir -p $OUT
for rep in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','single_call_ms','stage_ms','resident_generators_ms_per_step')})"; done
