#!/bin/bash
set -u
OUT=gpurun_out/r3c5
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_front_fusion.log --steps 200 --warmup 10 -- - BLITZAR_AMD_FAST_RECODE=0 BLITZAR_AMD_FUSE_PREPARE=0 BLITZAR_AMD_FAST_RECODE=0,BLITZAR_AMD_FUSE_PREPARE=0
grep -E "^==|ms_per_step" $OUT/ab_front_fusion.log | sed -E 's/"hash.*//'
for c in 1 2; do tools/pipeline_bench/_build/pipeline_bench --curve $c --log2n 20 --steps 20 | sed -E 's/"hash.*//'; BLITZAR_AMD_FAST_RECODE=0 BLITZAR_AMD_FUSE_PREPARE=0 tools/pipeline_bench/_build/pipeline_bench --curve $c --log2n 20 --steps 20 | sed -E 's/"hash.*//'; done
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -12 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','single_call_ms','stage_ms','resident_generators_ms_per_step','verified')})"
