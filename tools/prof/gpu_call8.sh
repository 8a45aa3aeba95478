#!/bin/bash
set -u
OUT=gpurun_out/r3c8
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_acc_prefetch.log --steps 200 --warmup 10 -- - -@blitzar_amd/lib/variants/np4 -@blitzar_amd/lib/variants/np5
tools/prof/ab_pipeline.sh $OUT/ab_acc_prefetch_resident.log --steps 200 --warmup 10 --resident -- - -@blitzar_amd/lib/variants/np4 -@blitzar_amd/lib/variants/np5
tools/prof/ab_pipeline.sh $OUT/ab_acc_prefetch_2_24.log --steps 10 --warmup 2 --log2n 24 -- - -@blitzar_amd/lib/variants/np4
grep -E "^==|ms_per_step" $OUT/ab_acc_prefetch*.log | sed -E 's/"outputs_agree.*//'
