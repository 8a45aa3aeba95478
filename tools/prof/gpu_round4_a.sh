#!/bin/bash
# round 4, first GPU session: the -m gpu suite on the new front (fused pass 1b, oversized groups
# inside pass 2's launch, one LDS atomic per record), then the A/B of every knob through the native
# driver, a kernel timeline of a sequence, and what the box's SMI tools report.
# usage (repo root, on the GPU box): tools/prof/gpu_round4_a.sh
set -u
OUT=gpurun_out/r4a
mkdir -p $OUT
( rocm-smi --showclocks --showpower --showcomputepartition --showmemorypartition --showperflevel 2>&1 | head -60 ) > $OUT/smi_idle.txt
( amd-smi static --asic --board 2>&1 | head -40; amd-smi metric --clock --power 2>&1 | head -60 ) > $OUT/amdsmi_idle.txt
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -14 $OUT/pytest_gpu.txt
AB=tools/prof/ab_pipeline.sh
$AB $OUT/ab_config2.log --steps 200 -- - BLITZAR_AMD_FUSE_BIG=1 BLITZAR_AMD_FUSE_BIG=0 BLITZAR_AMD_FUSE_OFFSETS=0 BLITZAR_AMD_RANK_ONCE=0 BLITZAR_AMD_FUSE_BIG=0,BLITZAR_AMD_FUSE_OFFSETS=0,BLITZAR_AMD_RANK_ONCE=0
$AB $OUT/ab_config2_resident.log --steps 200 --resident -- - BLITZAR_AMD_FUSE_BIG=0,BLITZAR_AMD_FUSE_OFFSETS=0,BLITZAR_AMD_RANK_ONCE=0
$AB $OUT/ab_config3.log --curve 1 --log2n 22 --steps 20 --warmup 3 -- - BLITZAR_AMD_FUSE_BIG=0,BLITZAR_AMD_FUSE_OFFSETS=0,BLITZAR_AMD_RANK_ONCE=0
$AB $OUT/ab_bn254_16col.log --curve 2 --log2n 20 --columns 16 --steps 6 --warmup 2 -- - BLITZAR_AMD_FUSE_BIG=0,BLITZAR_AMD_FUSE_OFFSETS=0,BLITZAR_AMD_RANK_ONCE=0
$AB $OUT/ab_2_16.log --log2n 16 --steps 200 -- - BLITZAR_AMD_FUSE_BIG=0,BLITZAR_AMD_FUSE_OFFSETS=0,BLITZAR_AMD_RANK_ONCE=0
grep -h -E "^==|ms_per_step" $OUT/ab_*.log | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//' 
# kernel timeline of a sequence and of lone calls (native driver under rocprofv3)
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace -o t -- $REPO/tools/pipeline_bench/_build/pipeline_bench --steps 40 --warmup 10 > $REPO/$OUT/trace_log.txt 2>&1
cd $REPO
CSV=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/prof/timeline.py $CSV 30 14 30 > $OUT/timeline_sequence.txt 2>&1
python tools/prof/timeline.py $CSV 200 14 14 > $OUT/timeline_lone.txt 2>&1
rm -rf $OUT/trace
head -50 $OUT/timeline_sequence.txt
( rocm-smi --showclocks --showpower 2>&1 | head -40 ) > $OUT/smi_after.txt
