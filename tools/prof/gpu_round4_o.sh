#!/bin/bash
# round 4, fifteenth GPU session: the tails of the Weierstrass curves, A/B on one box:
#   wave       the library as it is: lane-spread Horner chain, quad-shared additions in the window fold,
#              Kaliski inversion, k_reduce's lane weights as a suffix scan
#   noscan     the same with -DBZ_SW_REDUCE_SCAN=0 (per-lane double-and-add in k_reduce)
#   coop       the forms of round 3 (blitzar_amd/lib/variants/coop_horner: built before all of the above)
set -u
OUT=gpurun_out/r4o
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fullsize_golden.py tests/test_window_tables.py tests/test_baseline_configs.py -m gpu -x -q > $OUT/pytest_parity.txt 2>&1
echo "parity rc=$?" >> $OUT/pytest_parity.txt; tail -6 $OUT/pytest_parity.txt
for tag in wave noscan coop; do
  case $tag in
    wave) lib=blitzar_amd/lib/libblitzar_amd.so;;
    noscan) lib=blitzar_amd/lib/variants/no_reduce_scan/libblitzar_amd.so;;
    coop) lib=blitzar_amd/lib/variants/coop_horner/libblitzar_amd.so;;
  esac
  BLITZAR_AMD_LIB=$PWD/$lib timeout 250 python tools/prof/horner_phases.py > $OUT/phases_$tag.txt 2>&1
  echo "== $tag"; grep -h "per window" $OUT/phases_$tag.txt | grep -v curve25519
  python - "$OUT/phases_$tag.txt" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: v["32"]["reduce_ms"] for k, v in d.items()}, "reduce alone, 2^14 rows")
PY
  BLITZAR_AMD_LIB=$PWD/$lib timeout 500 python bench.py --steps 20 --warmup 5 --no-aux > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line:", e); sys.exit(0)
print({k: d.get(k) for k in ("ms_per_step", "single_call_ms", "sustained_ms_per_step")})
for c in d.get("configs", [])[1:]:
    print(c["config"][:44], "ms %.3f" % c["ms_per_call"], "lone", c.get("lone_call_ms"), c.get("lone_call_stage_ms"), {k: c["stage_ms_per_call"][k] for k in ("reduce", "combine")})
PY
done
