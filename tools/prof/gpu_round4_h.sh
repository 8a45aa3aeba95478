#!/bin/bash
# round 4, eighth GPU session: the probe-driven default (k_reduce_compact on slow-fetch boxes), the
# whole -m gpu suite, the bench line as the driver runs it, and the rocprofv3 / PMC passes of the same
# command with the configs legs -> profiles/round4_rocprofv3_summary.md, roofline_traffic.json
set -u
TAG=${1:-r4final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
tools/ubench/bin/tail_latency 2>&1 | tail -12 | head -7
AB=tools/prof/ab_pipeline.sh
$AB $OUT/ab_probe_config2.log --steps 200 -- - BLITZAR_AMD_COMPACT_REDUCE=0 BLITZAR_AMD_COMPACT_REDUCE=1
$AB $OUT/ab_probe_config3.log --curve 1 --log2n 22 --steps 20 --warmup 3 -- - BLITZAR_AMD_COMPACT_REDUCE=0 BLITZAR_AMD_COMPACT_REDUCE=1
grep -h -E "^==|ms_per_step" $OUT/ab_*.log | sed -E 's/"sequence_stage_ms.*"lone_stage_ms"/"lone_stage_ms"/; s/, "outputs_agree.*//; s/"host_enqueue_ms": [0-9.]*, //'
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -9 $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -2 $OUT/bench.err
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "single_call_ms", "sustained_ms_per_step", "stage_ms",
                             "resident_generators_ms_per_step", "verified")})
print("device_state", json.dumps(d.get("device_state"))[:700])
print("host_api", json.dumps((d.get("host_api") or {}).get("warm"))[:900])
for c in d.get("configs", []):
    r = c.get("roofline") or {}
    print(c["config"][:44], round(c["ms_per_call"], 2), "lone", c.get("lone_call_ms"), "frac", r.get("frac"), "cpu", (c.get("cpu_baseline") or {}).get("value"))
PY
tools/prof/run_pmc_configs.sh $TAG
tools/pipeline_bench/_build/pipeline_bench --steps 200 > $OUT/pipeline_bench_200.json
tools/pipeline_bench/_build/pipeline_bench --steps 20 --warmup 50 > $OUT/pipeline_bench_20.json
cat $OUT/pipeline_bench_200.json $OUT/pipeline_bench_20.json | sed -E 's/"outputs_agree.*//'
