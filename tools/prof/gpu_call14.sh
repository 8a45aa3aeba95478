#!/bin/bash
set -u
OUT=gpurun_out/r3c14
mkdir -p $OUT
for w in 5 50 200; do for rep in 1 2; do echo "== warmup $w steps 20"; tools/pipeline_bench/_build/pipeline_bench --steps 20 --warmup $w | sed -E 's/"host_enqueue.*"acc_in/ "acc_in/; s/"lone_stage.*//'; done; done
echo "== steps 200"; tools/pipeline_bench/_build/pipeline_bench --steps 200 --warmup 10 | sed -E 's/"host_enqueue.*"acc_in/ "acc_in/; s/"lone_stage.*//'
