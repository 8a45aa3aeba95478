#!/bin/bash
# one gpurun call at the end of round 3: the -m gpu suite, the bench line as the driver runs it, the
# rocprofv3 / PMC passes of the same command (with the configs legs), the native drivers
# usage (repo root, on the GPU box): tools/prof/gpu_round3.sh <tag>
set -u
TAG=${1:-r3final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -14 $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -2 $OUT/bench.err
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "single_call_ms", "stage_ms",
                             "resident_generators_ms_per_step", "verified")})
for c in d.get("configs", []):
    r = c.get("roofline") or {}
    print(c["config"][:44], round(c["ms_per_call"], 2), "frac", r.get("frac"), "traffic", r.get("traffic"),
          "valu_busy", r.get("valu_busy"), "alu", (r.get("alu") or {}).get("frac"))
PY
tools/prof/run_pmc_configs.sh $TAG
tools/pipeline_bench/_build/pipeline_bench --steps 200 > $OUT/pipeline_bench_200.json
tools/pipeline_bench/_build/pipeline_bench --steps 20 --warmup 50 > $OUT/pipeline_bench_20.json
cat $OUT/pipeline_bench_200.json $OUT/pipeline_bench_20.json | sed -E 's/"outputs_agree.*//'
