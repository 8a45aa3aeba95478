#!/bin/bash
set -u
OUT=gpurun_out/r3c12
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_reduce_residency.log --steps 200 --warmup 10 -- - BLITZAR_AMD_REDUCE_WGS_PER_CU=1 BLITZAR_AMD_REDUCE_WGS_PER_CU=2 BLITZAR_AMD_REDUCE_WGS_PER_CU=1,BLITZAR_AMD_TAIL_STREAMS=1
tools/prof/ab_pipeline.sh $OUT/ab_reduce_residency_k20.log --steps 20 --warmup 5 -- - BLITZAR_AMD_REDUCE_WGS_PER_CU=1
grep -E "^==|ms_per_step" $OUT/ab_reduce_residency*.log | sed -E 's/"host_enqueue.*"acc_in/ "acc_in/; s/"outputs_agree.*//'
