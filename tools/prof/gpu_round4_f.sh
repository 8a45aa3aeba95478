#!/bin/bash
# round 4, sixth GPU session: the tail kernels with few addition sites (code that fits the I-cache)
set -u
OUT=gpurun_out/r4f
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
tools/ubench/bin/tail_latency 2>&1 | tail -12 > $OUT/tail_latency.txt; cat $OUT/tail_latency.txt
AB=tools/prof/ab_pipeline.sh
$AB $OUT/ab_compact_config2.log --steps 200 -- - BLITZAR_AMD_COMPACT_TAILS=1
$AB $OUT/ab_compact_resident.log --steps 200 --resident -- - BLITZAR_AMD_COMPACT_TAILS=1
$AB $OUT/ab_compact_2_16.log --log2n 16 --steps 200 -- - BLITZAR_AMD_COMPACT_TAILS=1
$AB $OUT/ab_compact_config3.log --curve 1 --log2n 22 --steps 20 --warmup 3 -- - BLITZAR_AMD_COMPACT_TAILS=1
$AB $OUT/ab_compact_bn254_1col.log --curve 2 --log2n 20 --columns 1 --steps 50 --warmup 3 -- - BLITZAR_AMD_COMPACT_TAILS=1
$AB $OUT/ab_compact_bn254_16col.log --curve 2 --log2n 20 --columns 16 --steps 6 --warmup 2 -- - BLITZAR_AMD_COMPACT_TAILS=1
$AB $OUT/ab_compact_grumpkin_64col.log --curve 3 --log2n 18 --columns 64 --steps 6 --warmup 2 -- - BLITZAR_AMD_COMPACT_TAILS=1
$AB $OUT/ab_compact_skew.log --log2n 20 --steps 50 --skew -- - BLITZAR_AMD_COMPACT_TAILS=1
grep -h -E "^==|ms_per_step" $OUT/ab_*.log | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//; s/"host_enqueue_ms": [0-9.]*, //'
BLITZAR_AMD_COMPACT_TAILS=1 timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_compact.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_compact.txt
tail -6 $OUT/pytest_gpu_compact.txt
