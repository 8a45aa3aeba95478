#!/bin/bash
# per-call window tables: the doubling chain with a wavefront per generator against a lane per generator
P=tools/pipeline_bench/_build/pipeline_bench
for curve in 0 2 1; do for l in 8 10 12; do
  for spec in "BLITZAR_AMD_CALL_TABLE_WAVE_CHAIN=0" "X=1"; do
    for rep in 1 2; do
    echo -n "curve=$curve rows=2^$l $spec: "
    env $spec $P --curve $curve --log2n $l --columns 1024 --steps 20 --warmup 3 | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('seq %.4f lone %.4f lone stages %s agree %s hash %s'%(d['ms_per_step'], d['lone_ms'], d['lone_stage_ms'], d['outputs_agree'], d['hash']))"
    done
  done
done; done
