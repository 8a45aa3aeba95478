#!/bin/bash
# A/B of environment knobs on the BASELINE configs legs of bench.py (config 2 shortened), one box:
# usage: tools/prof/ab_configs_env.sh <out file> "K=V,..." ["K=V,..." ...]   ("-" = no overrides)
cd $GRAFT_REPO_ROOT
OUT=$1; shift
mkdir -p $(dirname $OUT)
for spec in "$@"; do
  envs=$spec; [ "$spec" = "-" ] && envs=""
  echo "== $spec" >> $OUT
  env $(echo $envs | tr ',' ' ') timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('cfg2 ms %.4f' % d['ms_per_step'], d['stage_ms'])
for c in d['configs'][1:]:
    print(c['config'][:40], 'ms %.2f' % c['ms_per_call'], c['stage_ms_per_call'], c.get('verified','')[:20])
" >> $OUT
done
cat $OUT
