#!/bin/bash
# round 4, seventh GPU session: compact tails, second version (ballot-skipped additions, rotated walk)
set -u
OUT=gpurun_out/r4g
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
tools/ubench/bin/tail_latency 2>&1 | tail -12 > $OUT/tail_latency.txt; head -7 $OUT/tail_latency.txt
AB=tools/prof/ab_pipeline.sh
$AB $OUT/ab_compact_config2.log --steps 200 -- - BLITZAR_AMD_COMPACT_TAILS=1 BLITZAR_AMD_COMPACT_TAILS=2 BLITZAR_AMD_COMPACT_TAILS=3
$AB $OUT/ab_compact_2_16.log --log2n 16 --steps 200 -- - BLITZAR_AMD_COMPACT_TAILS=1 BLITZAR_AMD_COMPACT_TAILS=2
$AB $OUT/ab_compact_config3.log --curve 1 --log2n 22 --steps 20 --warmup 3 -- - BLITZAR_AMD_COMPACT_TAILS=1 BLITZAR_AMD_COMPACT_TAILS=2
$AB $OUT/ab_compact_bn254_1col.log --curve 2 --log2n 20 --columns 1 --steps 50 --warmup 3 -- - BLITZAR_AMD_COMPACT_TAILS=1 BLITZAR_AMD_COMPACT_TAILS=2
$AB $OUT/ab_compact_bn254_16col.log --curve 2 --log2n 20 --columns 16 --steps 6 --warmup 2 -- - BLITZAR_AMD_COMPACT_TAILS=1
grep -h -E "^==|ms_per_step" $OUT/ab_*.log | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//; s/"host_enqueue_ms": [0-9.]*, //'
BLITZAR_AMD_COMPACT_TAILS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q > $OUT/pytest_gpu_compact.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_compact.txt
tail -4 $OUT/pytest_gpu_compact.txt
