#!/bin/bash
# round 3, GPU call 1: A/B of the cross-call front overlap, the GPU test suite, bench.py, and the
# rocprofv3 / PMC passes of the configs legs
set -u
OUT=gpurun_out/r3c1
mkdir -p $OUT
rocm-smi --showclocks 2>/dev/null | head -20 > $OUT/clocks.txt
tools/prof/ab_pipeline.sh $OUT/ab_front.log --steps 200 --warmup 10 -- \
  BLITZAR_AMD_OVERLAP_FRONT=0 \
  - \
  BLITZAR_AMD_FRONT_CUS=16 \
  BLITZAR_AMD_FRONT_CUS=24 \
  BLITZAR_AMD_FRONT_CUS=48 \
  BLITZAR_AMD_FRONT_CUS=64 \
  BLITZAR_AMD_FRONT_CUS=0 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=1 \
  BLITZAR_AMD_FRONT_CUS=32,BLITZAR_AMD_ACC_MASKED=0 \
  BLITZAR_AMD_TAIL_STREAMS=1
grep -E "^==|ms_per_step|rc=" $OUT/ab_front.log | sed -E 's/"sequence_stage_ms.*//' | head -80
tools/pipeline_bench/_build/pipeline_bench --steps 20 --warmup 5 > $OUT/pb_20.json 2>&1
tools/pipeline_bench/_build/pipeline_bench --steps 200 --null-stream > $OUT/pb_null.json 2>&1
tools/pipeline_bench/_build/pipeline_bench --steps 50 --curve 1 --log2n 22 > $OUT/pb_bls.json 2>&1
BLITZAR_AMD_OVERLAP_FRONT=0 tools/pipeline_bench/_build/pipeline_bench --steps 50 --curve 1 --log2n 22 > $OUT/pb_bls_nofront.json 2>&1
cat $OUT/pb_20.json $OUT/pb_null.json $OUT/pb_bls.json $OUT/pb_bls_nofront.json | sed -E 's/"sequence_stage_ms.*//'
timeout 900 python -m pytest tests -m gpu -q -x --durations=10 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -15 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err; head -c 1500 $OUT/bench.json; echo
tools/prof/run_pmc_configs.sh r3c1
