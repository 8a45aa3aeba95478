#!/bin/bash
# A/B of builds on one MI355X box (box-to-box variance is ~10 %): the current library against
# variant builds under blitzar_amd/lib/ab/*.so, alternating, via BLITZAR_AMD_LIB.
# usage: tools/prof/ab_bench.sh <out file> [variant.so ...]
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/ab.log}; shift || true
mkdir -p $(dirname $OUT)
for rep in 1 2; do
for lib in blitzar_amd/lib/libblitzar_amd.so "$@"; do
  echo "== $lib" >> $OUT
  BLITZAR_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-configs 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*"resident_generators_ms_per_step": \([0-9.]*\), "stage_ms": \({[^}]*}\).*/ms \1 resident \2 \3/' >> $OUT
done
done
cat $OUT
