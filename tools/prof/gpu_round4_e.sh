#!/bin/bash
# round 4, fifth GPU session: streamed-group threshold of pass 2 (the top window of 252-bit scalars
# puts 65536 records into each of 16 groups: one workgroup streams each in 11 rounds), the row
# pipeline with three staging regions, the whole -m gpu suite with the new arrangement variants
set -u
OUT=gpurun_out/r4e
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
AB=tools/prof/ab_pipeline.sh
$AB $OUT/ab_stream_config2.log --steps 200 -- - BLITZAR_AMD_SORT_STREAM_FACTOR=8 BLITZAR_AMD_SORT_STREAM_FACTOR=4 BLITZAR_AMD_SORT_STREAM_FACTOR=2 BLITZAR_AMD_SORT_STREAM_FACTOR=1 BLITZAR_AMD_SORT_STREAM_FACTOR=4,BLITZAR_AMD_FUSE_BIG=0
$AB $OUT/ab_stream_2_16.log --log2n 16 --steps 200 -- - BLITZAR_AMD_SORT_STREAM_FACTOR=4 BLITZAR_AMD_SORT_STREAM_FACTOR=1
$AB $OUT/ab_stream_2_18.log --log2n 18 --steps 200 -- - BLITZAR_AMD_SORT_STREAM_FACTOR=4 BLITZAR_AMD_SORT_STREAM_FACTOR=1
$AB $OUT/ab_stream_config3.log --curve 1 --log2n 22 --steps 20 --warmup 3 -- - BLITZAR_AMD_SORT_STREAM_FACTOR=4 BLITZAR_AMD_SORT_STREAM_FACTOR=1
$AB $OUT/ab_stream_bn254_16col.log --curve 2 --log2n 20 --columns 16 --steps 6 --warmup 2 -- - BLITZAR_AMD_SORT_STREAM_FACTOR=4 BLITZAR_AMD_SORT_STREAM_FACTOR=1
grep -h -E "^==|ms_per_step" $OUT/ab_*.log | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//'
timeout 300 tools/pipeline_bench/_build/hostapi_bench > $OUT/hostapi_bench.json 2>&1; cat $OUT/hostapi_bench.json
tools/ubench/bin/tail_latency > $OUT/tail_latency.txt 2>&1; tail -22 $OUT/tail_latency.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -10 $OUT/pytest_gpu.txt
