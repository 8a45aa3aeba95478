#!/bin/bash
set -u
OUT=gpurun_out/r3c4
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_k200.log --steps 200 --warmup 10 -- BLITZAR_AMD_OVERLAP_FRONT=0 - BLITZAR_AMD_DEDICATED_QUEUES=0 BLITZAR_AMD_OVERLAP_FRONT=0,BLITZAR_AMD_DEDICATED_QUEUES=0
tools/prof/ab_pipeline.sh $OUT/ab_k20.log --steps 20 --warmup 5 -- BLITZAR_AMD_OVERLAP_FRONT=0 -
grep -E "^==|ms_per_step" $OUT/ab_k20.log $OUT/ab_k200.log | sed -E 's/"sequence_stage_ms.*//'
for v in "A=1" "BLITZAR_AMD_OVERLAP_FRONT=0" "BLITZAR_AMD_DEDICATED_QUEUES=0"; do
  echo "== bench.py $v"
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('ms_per_step','single_call_ms','stage_ms','resident_generators_ms_per_step')})"
done
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -25 $OUT/pytest_gpu.txt
