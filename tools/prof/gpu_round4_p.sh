#!/bin/bash
# round 4, sixteenth GPU session: the whole -m gpu suite and the bench line on the library with
# k_reduce<C, Scan> chosen per launch
set -u
OUT=gpurun_out/r4p
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -6 $OUT/pytest_gpu.txt
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line:", e); sys.exit(0)
print({k: d.get(k) for k in ("ms_per_step", "single_call_ms", "sustained_ms_per_step", "stage_ms")})
print(d.get("device_state", {}).get("slow_instruction_fetch"), json.dumps(d.get("host_api"))[:300])
for c in d.get("configs", [])[1:]:
    print(c["config"][:44], "ms %.3f" % c["ms_per_call"], "lone", c.get("lone_call_ms"), c.get("lone_call_stage_ms"), {k: c["stage_ms_per_call"][k] for k in ("reduce", "combine")})
PY
