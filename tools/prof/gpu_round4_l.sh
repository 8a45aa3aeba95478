#!/bin/bash
# round 4, twelfth GPU session: bench.py choosing the throughput mode's arrangement by measurement
set -u
OUT=gpurun_out/r4l
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
timeout 600 python -m pytest tests/test_multi_device.py tests/test_bench_multi_rank.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
for rep in 1 2; do
time (python bench.py --steps 20 --warmup 5 --no-configs > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err); tail -2 $OUT/bench_$rep.err
python - "$OUT/bench_$rep.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "single_call_ms", "sustained_ms_per_step", "stage_ms", "verified")})
print(json.dumps(d.get("pipeline_arrangement"))[:400])
PY
done
python bench.py --steps 20 --warmup 5 --no-configs --arrangement 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pinned 0:', d['ms_per_step'], d['sustained_ms_per_step'])"
