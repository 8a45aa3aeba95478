#!/bin/bash
# round 4, third GPU session: new tests (leases, re-entrant callback, row pipeline, two-rank dry run),
# the drop-in host path with the row pipeline, and the bench line as the driver runs it.
set -u
OUT=gpurun_out/r4c
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -12 $OUT/pytest_gpu.txt
timeout 300 tools/pipeline_bench/_build/hostapi_bench > $OUT/hostapi_bench.json 2>&1; cat $OUT/hostapi_bench.json
for r in 1 2 4 8; do
  python - $r <<'PY' >> $OUT/hostapi_chunks.txt 2>&1
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from blitzar_amd import api
r = int(sys.argv[1]); n = 1 << 20
api.init(api.SXT_GPU_BACKEND, n)
lib = api.load(); lib.bzamd_set_row_pipeline_chunks(r)
rng = np.random.default_rng(0)
g = api.get_generators(n, 0).view(np.uint8).reshape(n, 160)
for cols_n in (1, 10):
    cols = []
    for _ in range(cols_n):
        s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0f
        cols.append((s, False))
    for name, gens in (("caller", g), ("builtin", None)):
        for _ in range(2): api.compute_pedersen_commitments(0, cols, generators=gens)
        t0 = time.perf_counter()
        for _ in range(6): out = api.compute_pedersen_commitments(0, cols, generators=gens)
        print(f"chunks {r} cols {cols_n} {name}: {(time.perf_counter()-t0)/6*1e3:.3f} ms", out[0,:2].tolist())
PY
done
cat $OUT/hostapi_chunks.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "single_call_ms", "sustained_ms_per_step", "effective_warmup_calls", "stage_ms",
                             "resident_generators_ms_per_step", "verified")})
print("device_state", json.dumps(d.get("device_state"))[:1500])
print("host_api", json.dumps(d.get("host_api"))[:2500])
for c in d.get("configs", []):
    r = c.get("roofline") or {}
    print(c["config"][:44], round(c["ms_per_call"], 2), "lone", c.get("lone_call_ms"), "frac", r.get("frac"), "cpu", (c.get("cpu_baseline") or {}).get("value"), (c.get("cpu_baseline") or {}).get("kind"), c.get("data", "")[:60])
PY
