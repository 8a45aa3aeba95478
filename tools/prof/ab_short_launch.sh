#!/bin/bash
# lone short launches: entries per accumulation lane chosen by the planner against the 32 of rounds 1-5
P=tools/pipeline_bench/_build/pipeline_bench
for nb in 32 1; do for l in 12 14 16 17 18 20; do
  for spec in "BLITZAR_AMD_SEGMENT_LOG2=5" "X=1"; do
    echo -n "nbytes=$nb rows=2^$l $spec: "
    env $spec $P --log2n $l --nbytes $nb --steps 50 --warmup 5 | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('seq %.4f lone %.4f lone stages %s'%(d['ms_per_step'], d['lone_ms'], d['lone_stage_ms']))"
  done
done; done
