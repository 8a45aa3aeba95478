#!/bin/bash
# round 4, eleventh GPU session: arrangements of the front beside the accumulation, native and under torch
set -u
OUT=gpurun_out/r4k
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
PB=tools/pipeline_bench/_build/pipeline_bench
for spec in "" "BLITZAR_AMD_OVERLAP_FRONT=1" "BLITZAR_AMD_OVERLAP_FRONT=1 BLITZAR_AMD_DEDICATED_QUEUES=0" "BLITZAR_AMD_OVERLAP_FRONT=1 BLITZAR_AMD_DEDICATED_QUEUES=0 BLITZAR_AMD_FRONT_PRIORITY=0" "BLITZAR_AMD_OVERLAP_FRONT=1 BLITZAR_AMD_TAIL_LOW_PRIORITY=0" "BLITZAR_AMD_OVERLAP_FRONT=1 BLITZAR_AMD_DEDICATED_QUEUES=0 BLITZAR_AMD_TAIL_LOW_PRIORITY=0"; do
  for kind in native torch hip; do
    for rep in 1 2; do
      if [ $kind = native ]; then
        r=$(env $spec $PB --steps 200 | sed -E 's/.*"ms_per_step": ([0-9.]+).*/\1/')
      else
        r=$(env $spec python tools/prof/front_overlap_under_torch.py $kind 200 2>/dev/null | tail -1 | sed -E 's/.*"ms_per_step": ([0-9.]+).*/\1/')
      fi
      echo "$kind | ${spec:-default} | $r"
    done
  done
done | tee $OUT/front_arrangements.txt
