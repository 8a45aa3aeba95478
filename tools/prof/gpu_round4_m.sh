#!/bin/bash
# round 4, thirteenth GPU session: the lane-spread Horner chain of the Weierstrass curves
# (curve/sw_wave.h) -- parity first, then lone-call stage times against the quad-split form of
# rounds 2-3 (blitzar_amd/lib/variants/coop_horner, -DBZ_SW_WAVE_HORNER=0) on the same box.
set -u
OUT=gpurun_out/r4m
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1
# parity: everything that runs a Weierstrass MSM
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fullsize_golden.py tests/test_window_tables.py -m gpu -x -q > $OUT/pytest_parity.txt 2>&1
echo "parity rc=$?" >> $OUT/pytest_parity.txt; tail -15 $OUT/pytest_parity.txt
# the two-rank dry run of bench.py (hung in session 12 under the stream-layout walk, now opt-in)
timeout 400 python -m pytest tests/test_bench_multi_rank.py -m gpu -x -q > $OUT/pytest_two_ranks.txt 2>&1
echo "two ranks rc=$?" >> $OUT/pytest_two_ranks.txt; tail -5 $OUT/pytest_two_ranks.txt
for lib in blitzar_amd/lib/libblitzar_amd.so blitzar_amd/lib/variants/coop_horner/libblitzar_amd.so; do
  tag=$(echo $lib | grep -q variants && echo coop || echo wave)
  BLITZAR_AMD_LIB=$PWD/$lib timeout 500 python bench.py --steps 20 --warmup 5 --no-aux > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  echo "bench $tag rc=$?"
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line:", e); sys.exit(0)
print({k: d.get(k) for k in ("ms_per_step", "single_call_ms", "sustained_ms_per_step")})
for c in d.get("configs", [])[1:]:
    print(c["config"][:44], "ms %.3f" % c["ms_per_call"], "lone", c.get("lone_call_ms"), c.get("lone_call_stage_ms"))
PY
done
