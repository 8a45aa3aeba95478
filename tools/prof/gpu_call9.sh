#!/bin/bash
set -u
OUT=gpurun_out/r3c9
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_reduce_lds.log --steps 200 --warmup 10 -- -@blitzar_amd/lib/variants/old - -@blitzar_amd/lib/variants/r2 -@blitzar_amd/lib/variants/r3 -@blitzar_amd/lib/variants/r4
tools/prof/ab_pipeline.sh $OUT/ab_reduce_lds_k20.log --steps 20 --warmup 5 -- -@blitzar_amd/lib/variants/old - -@blitzar_amd/lib/variants/r3
grep -E "^==|ms_per_step" $OUT/ab_reduce_lds*.log | sed -E 's/"outputs_agree.*//'
grep -c '"outputs_agree": true' $OUT/ab_reduce_lds.log; grep -o '"hash": "[0-9a-f]*"' $OUT/ab_reduce_lds.log | sort | uniq -c
