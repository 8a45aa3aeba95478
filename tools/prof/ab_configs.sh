#!/bin/bash
# A/B of builds on the BASELINE configs legs of bench.py (one box): usage as tools/prof/ab_env.sh
cd $GRAFT_REPO_ROOT
OUT=$1; shift
mkdir -p $(dirname $OUT)
for spec in "$@"; do
  envs=${spec%@*}; lib=${spec#*@}
  [ -z "$lib" ] && lib=blitzar_amd/lib/libblitzar_amd.so
  echo "== $spec" >> $OUT
  env $(echo $envs | tr ',' ' ') BLITZAR_AMD_LIB=$PWD/$lib timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('cfg2 ms %.4f' % d['ms_per_step'], d['stage_ms'])
for c in d['configs'][1:]:
    print(c['config'][:40], 'ms %.2f' % c['ms_per_call'], c['stage_ms_per_call'])
" >> $OUT
done
cat $OUT
