#!/bin/bash
# round 4, fourth GPU session: row pipeline with the uploader thread, instruction-fetch hypothesis
# for the two kinds of boxes (ubench + SQC_ICACHE counters of the tail kernels)
set -u
OUT=gpurun_out/r4d
mkdir -p $OUT
python tools/prof/device_state.py > $OUT/device_state.json 2>&1; cat $OUT/device_state.json
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_pipeline or several_passes or two_chunks" > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
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
for cols_n in (1, 2, 10):
    cols = []
    for _ in range(cols_n):
        s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0f
        cols.append((s, False))
    for name, gens in (("caller", g),):
        for _ in range(2): api.compute_pedersen_commitments(0, cols, generators=gens)
        t0 = time.perf_counter()
        for _ in range(6): out = api.compute_pedersen_commitments(0, cols, generators=gens)
        print(f"chunks {r} cols {cols_n} {name}: {(time.perf_counter()-t0)/6*1e3:.3f} ms", out[0,:2].tolist())
PY
done
grep -v amdgpu.ids $OUT/hostapi_chunks.txt
tools/ubench/bin/tail_latency > $OUT/tail_latency.txt 2>&1; cat $OUT/tail_latency.txt
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $REPO/$OUT/pmc_icache -o r -- $REPO/tools/pipeline_bench/_build/pipeline_bench --steps 6 --warmup 2 > $REPO/$OUT/pmc_icache.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("bz::", "")[:40]
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':42s} launches " + " ".join(f"{c:>16s}" for c in ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU")))
for name, cs in sorted(rows.items()):
    n = max(len(v) for v in cs.values())
    print(f"{name:42s} {n:8d} " + " ".join(f"{(sum(cs[c]) / max(len(cs[c]), 1)):16.3e}" if c in cs else f"{'-':>16s}" for c in ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU")))
PY
rm -rf $OUT/pmc_icache
