#!/bin/bash
# bench line as the driver runs it + the rocprofv3 / PMC passes of the same command (with configs)
set -u
OUT=gpurun_out/r3c7
mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c7/bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','single_call_ms','stage_ms','resident_generators_ms_per_step','verified')})
print(d['roofline'])
for c in d.get('configs',[]): print(c['config'][:40], c.get('ms_per_call'), c.get('cpu_baseline',{}).get('value'))
PY
tools/prof/run_pmc_configs.sh r3c7
tools/pipeline_bench/_build/pipeline_bench --steps 200 > $OUT/pb200.json; tools/pipeline_bench/_build/pipeline_bench --steps 20 --warmup 5 > $OUT/pb20.json; cat $OUT/pb200.json $OUT/pb20.json | sed -E 's/"outputs_agree.*//'
