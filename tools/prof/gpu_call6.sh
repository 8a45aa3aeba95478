#!/bin/bash
set -u
OUT=gpurun_out/r3c6
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_lane_geometry_in_sequence.log --steps 200 --warmup 10 -- - \
  BLITZAR_AMD_REDUCE_SEGMENT_LOG2=4 BLITZAR_AMD_REDUCE_SEGMENT_LOG2=5 BLITZAR_AMD_REDUCE_SEGMENT_LOG2=6 \
  BLITZAR_AMD_SEGMENT_LOG2=6 BLITZAR_AMD_SEGMENT_LOG2=6,BLITZAR_AMD_REDUCE_SEGMENT_LOG2=5 \
  BLITZAR_AMD_SEGMENT_LOG2=7,BLITZAR_AMD_REDUCE_SEGMENT_LOG2=5 BLITZAR_AMD_TAIL_STREAMS=1,BLITZAR_AMD_REDUCE_SEGMENT_LOG2=5
grep -E "^==|ms_per_step" $OUT/ab_lane_geometry_in_sequence.log | sed -E 's/"outputs_agree.*//'
