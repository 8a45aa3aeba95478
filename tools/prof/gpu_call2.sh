#!/bin/bash
# round 3, GPU call 2: front beside the accumulation on shared CUs -- queue priority + occupancy caps
set -u
OUT=gpurun_out/r3c2
mkdir -p $OUT
tools/prof/ab_pipeline.sh $OUT/ab_front2.log --steps 200 --warmup 10 -- \
  BLITZAR_AMD_OVERLAP_FRONT=0 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=1,BLITZAR_AMD_FRONT_WAVES=0 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=1 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=1,BLITZAR_AMD_FRONT_WAVES=2 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=1,BLITZAR_AMD_FRONT_WAVES=8 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=0 \
  BLITZAR_AMD_OVERLAP_FRONT=0@blitzar_amd/lib/variants/accw2 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=1@blitzar_amd/lib/variants/accw2 \
  BLITZAR_AMD_FRONT_CUS=0,BLITZAR_AMD_FRONT_PRIORITY=1,GPU_MAX_HW_QUEUES=8
grep -E "^==|ms_per_step|rc=" $OUT/ab_front2.log | sed -E 's/"sequence_stage_ms.*//'
cd /tmp && export TMPDIR=/tmp
for v in "BLITZAR_AMD_FRONT_CUS=0 BLITZAR_AMD_FRONT_PRIORITY=1" "BLITZAR_AMD_FRONT_CUS=0 BLITZAR_AMD_FRONT_PRIORITY=0"; do
  tag=$(echo "$v" | tr -c 'A-Z0-9=' '_' | tail -c 30)
  env $v timeout 120 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_$tag -o r -- $GRAFT_REPO_ROOT/tools/pipeline_bench/_build/pipeline_bench --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$OUT/trace_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
for d in $OUT/trace_*/; do echo "== $d"; python tools/prof/timeline.py $(find $d -name "*kernel_trace.csv" | head -1) 20 2 30; done
