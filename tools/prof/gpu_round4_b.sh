#!/bin/bash
# round 4, second GPU session: the fused pass 1b without fences (A/B again), the new lock / golden /
# re-entrancy tests, copy rates of the box, the drop-in host path, the tail-latency ubench and the SMI
# tools' JSON output (for bench.py's device_state block).
set -u
OUT=gpurun_out/r4b
mkdir -p $OUT
( amd-smi static --asic --json 2>&1 | head -40; amd-smi metric --clock --power --json 2>&1 | head -120 ) > $OUT/amdsmi_json.txt
timeout 900 python -m pytest tests/test_multi_device.py tests/test_sumcheck.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
AB=tools/prof/ab_pipeline.sh
OLD=BLITZAR_AMD_FUSE_BIG=0,BLITZAR_AMD_FUSE_OFFSETS=0,BLITZAR_AMD_RANK_ONCE=0
$AB $OUT/ab_config2.log --steps 200 -- - BLITZAR_AMD_FUSE_BIG=1 BLITZAR_AMD_FUSE_BIG=0 BLITZAR_AMD_FUSE_OFFSETS=0 BLITZAR_AMD_RANK_ONCE=0 $OLD
$AB $OUT/ab_config2_resident.log --steps 200 --resident -- - $OLD
$AB $OUT/ab_config3.log --curve 1 --log2n 22 --steps 20 --warmup 3 -- - $OLD
$AB $OUT/ab_bn254_16col.log --curve 2 --log2n 20 --columns 16 --steps 6 --warmup 2 -- - $OLD
$AB $OUT/ab_2_16.log --log2n 16 --steps 200 -- - $OLD
grep -h -E "^==|ms_per_step" $OUT/ab_*.log | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//'
tools/ubench/bin/h2d_rates > $OUT/h2d_rates.txt 2>&1; cat $OUT/h2d_rates.txt
tools/ubench/bin/tail_latency > $OUT/tail_latency.txt 2>&1; tail -12 $OUT/tail_latency.txt
timeout 300 tools/pipeline_bench/_build/hostapi_bench > $OUT/hostapi_bench.json 2>&1; cat $OUT/hostapi_bench.json
timeout 300 tools/multi_commitment/_build/multi_commitment gpu 1048576 10 1 32 0 > $OUT/multi_commitment_1.txt 2>&1; grep -E "duration|throughput" $OUT/multi_commitment_1.txt
timeout 300 tools/multi_commitment/_build/multi_commitment gpu 1048576 10 10 32 0 > $OUT/multi_commitment_10.txt 2>&1; grep -E "duration|throughput" $OUT/multi_commitment_10.txt
ls /sys/class/drm/ 2>&1 | head; ls /sys/class/drm/card*/device/ 2>/dev/null | head -80 > $OUT/sysfs_device.txt
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace -o t -- $REPO/tools/pipeline_bench/_build/pipeline_bench --steps 40 --warmup 10 > $REPO/$OUT/trace_log.txt 2>&1
cd $REPO
CSV=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/prof/timeline.py $CSV 30 14 30 > $OUT/timeline_sequence.txt 2>&1
python tools/prof/timeline.py $CSV 200 14 14 > $OUT/timeline_lone.txt 2>&1
rm -rf $OUT/trace
head -36 $OUT/timeline_sequence.txt; tail -16 $OUT/timeline_lone.txt
