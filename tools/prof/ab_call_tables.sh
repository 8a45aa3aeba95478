#!/bin/bash
# A/B of the per-call window tables on the bucket_method2 regime
P=tools/pipeline_bench/_build/pipeline_bench
run() { # label env curve log2n
  for rep in 1 2; do
  echo -n "$1 curve=$3 rows=2^$4: "
  env $2 $P --curve $3 --log2n $4 --columns 1024 --steps 20 --warmup 3 | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('seq %.3f lone %.3f host %.3f stages %s agree %s hash %s'%(d['ms_per_step'], d['lone_ms'], d['host_enqueue_ms'], d['lone_stage_ms'], d['outputs_agree'], d['hash']))"
  done
}
for curve in 0 2; do for l in 8 10 12; do
  run "off      " "BLITZAR_AMD_CALL_TABLES=0" $curve $l
  run "model    " "X=1" $curve $l
  run "nooverlap" "BLITZAR_AMD_CALL_TABLE_OVERLAP=0" $curve $l
done; done
for b in 10 11 12 13 14; do run "bits=$b  " "BLITZAR_AMD_CALL_TABLE_BITS=$b" 0 12; done
for b in 8 9 10 11 12; do run "bits=$b  " "BLITZAR_AMD_CALL_TABLE_BITS=$b" 0 10; done
for b in 10 11 12 13 14; do run "bits=$b  " "BLITZAR_AMD_CALL_TABLE_BITS=$b" 2 12; done
