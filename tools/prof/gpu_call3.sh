#!/bin/bash
set -u
OUT=gpurun_out/r3c3
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -30 $OUT/pytest_gpu.txt
tools/prof/ab_pipeline.sh $OUT/ab_k20.log --steps 20 --warmup 5 -- BLITZAR_AMD_OVERLAP_FRONT=0 -
tools/prof/ab_pipeline.sh $OUT/ab_k200.log --steps 200 --warmup 10 -- BLITZAR_AMD_OVERLAP_FRONT=0 -
grep -E "^==|ms_per_step" $OUT/ab_k20.log $OUT/ab_k200.log | sed -E 's/"sequence_stage_ms.*//'
tools/pipeline_bench/_build/multi_device_check > $OUT/md_check.json 2>&1; cat $OUT/md_check.json
BLITZAR_AMD_FORCE_SHARDS=8 tools/pipeline_bench/_build/multi_device_check --columns 256 --log2n 20 --steps 1 > $OUT/md_check_8x256.json 2>&1; cat $OUT/md_check_8x256.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c3/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','single_call_ms','stage_ms')})
print(d.get('in_process_multi_device'))
for c in d.get('configs',[]): print(c['config'][:40], c.get('ms_per_call'), (c.get('roofline') or {}).get('traffic'), ((c.get('roofline') or {}).get('alu') or {}).get('frac'))
PY
