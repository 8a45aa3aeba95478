#!/bin/bash
# round 4, fourteenth GPU session: k_horner of the Weierstrass curves after the lane-spread chain --
# window fold with quad-shared additions, Kaliski inversion with bulk shifts.  Parity, then k_horner
# alone by populated windows (tools/prof/horner_phases.py), then the bench line.
set -u
OUT=gpurun_out/r4n
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fullsize_golden.py tests/test_window_tables.py tests/test_inner_product.py tests/test_fixed_base_cpu.py -m gpu -x -q > $OUT/pytest_parity.txt 2>&1
echo "parity rc=$?" >> $OUT/pytest_parity.txt; tail -15 $OUT/pytest_parity.txt
timeout 250 python tools/prof/horner_phases.py > $OUT/phases_wave_v2.txt 2>&1; grep -h "per window" $OUT/phases_wave_v2.txt
timeout 500 python bench.py --steps 20 --warmup 5 --no-aux > $OUT/bench_wave_v2.json 2> $OUT/bench_wave_v2.err
echo "bench rc=$?"
python - "$OUT/bench_wave_v2.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line:", e); sys.exit(0)
print({k: d.get(k) for k in ("ms_per_step", "single_call_ms", "sustained_ms_per_step", "stage_ms")})
for c in d.get("configs", [])[1:]:
    print(c["config"][:44], "ms %.3f" % c["ms_per_call"], "lone", c.get("lone_call_ms"), c.get("lone_call_stage_ms"))
PY
