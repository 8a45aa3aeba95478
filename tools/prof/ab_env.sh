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
  env $(echo $envs | tr ',' ' ') BLITZAR_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms %.4f' % d['ms_per_step'], d.get('stage_ms'), '| resident %.4f' % d.get('resident_generators_ms_per_step', 0))" >> $OUT
done
done
cat $OUT
