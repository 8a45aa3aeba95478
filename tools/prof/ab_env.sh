#!/bin/bash
# A/B of environment knobs / variant libraries on one box.  Each argument is "ENV=VAL,...@lib" ;
# usage: tools/prof/ab_env.sh <out file> <spec> [<spec> ...]   (spec: "[K=V,...]@[path to .so]")
cd $GRAFT_REPO_ROOT
OUT=$1; shift
mkdir -p $(dirname $OUT)
for rep in 1 2; do
for spec in "$@"; do
  envs=${spec%@*}; lib=${spec#*@}
  [ -z "$lib" ] && lib=blitzar_amd/lib/libblitzar_amd.so
  echo "== $spec" >> $OUT
  env $(echo $envs | tr ',' ' ') BLITZAR_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-configs 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*"resident_generators_ms_per_step": \([0-9.]*\), "stage_ms": \({[^}]*}\).*/ms \1 resident \2 \3/' >> $OUT
done
done
cat $OUT
