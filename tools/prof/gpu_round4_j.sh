#!/bin/bash
# round 4, tenth GPU session: the front beside the accumulation once more (native, and in a PyTorch
# process with / without torch's stream pool); wall time of the bench line
set -u
OUT=gpurun_out/r4j
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
AB=tools/prof/ab_pipeline.sh
$AB $OUT/ab_overlap_front_native.log --steps 200 -- - BLITZAR_AMD_OVERLAP_FRONT=1 BLITZAR_AMD_OVERLAP_FRONT=1,BLITZAR_AMD_FRONT_PRIORITY=0
$AB $OUT/ab_overlap_front_native_resident.log --steps 200 --resident -- - BLITZAR_AMD_OVERLAP_FRONT=1
grep -h -E "^==|ms_per_step" $OUT/ab_*.log | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//; s/"host_enqueue_ms": [0-9.]*, //'
for rep in 1 2; do
  for kind in torch hip; do
    for ov in 0 1; do
      BLITZAR_AMD_OVERLAP_FRONT=$ov python tools/prof/front_overlap_under_torch.py $kind 200 2>/dev/null | tail -1
    done
  done
done | tee $OUT/front_overlap_under_torch.jsonl
time (python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err); tail -2 $OUT/bench.err
