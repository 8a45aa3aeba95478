#!/bin/bash
set -u
OUT=gpurun_out/r3c11
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_group_entries.log --steps 200 --warmup 10 -- - BLITZAR_AMD_GROUP_ENTRIES=2048 BLITZAR_AMD_GROUP_ENTRIES=3072 BLITZAR_AMD_GROUP_ENTRIES=6144 BLITZAR_AMD_GROUP_ENTRIES=1024
grep -E "^==|ms_per_step" $OUT/ab_group_entries.log | sed -E 's/"outputs_agree.*//'
