#!/bin/bash
# One cheap draw from the pool: which kind of box is this?  On a slow-fetch box (a wavefront waits ~23 %
# longer for misses; 4 of ~20 draws in round 5) run the slow-box measurements; on a fast-fetch box leave
# after ~10 s.   usage (through gpurun): bash tools/prof/hunt_slow_fetch.sh <tag>
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
TAG=${1:-hunt}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
KIND=$(python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from blitzar_amd import api
lib = api.load()
assert api.init(api.SXT_GPU_BACKEND, 0) == 0
print({1: "slow", 0: "fast"}.get(lib.bzamd_slow_instruction_fetch(), "unknown"))
PY
)
echo "fetch kind: $KIND" | tee "$OUT/kind.txt"
[ "$KIND" != "slow" ] && exit 0
P=tools/pipeline_bench/_build/pipeline_bench
{
  amd-smi static --asic --json 2>/dev/null | grep -i serial
  for spec in "X=1" "BLITZAR_AMD_COMPACT_REDUCE=0" "BLITZAR_AMD_COMPACT_REDUCE=1"; do
    for rep in 1 2; do echo "== config 2 $spec"; env $spec $P --steps 200 --warmup 10 | tail -1; done
  done
  echo "== 2^16 rows"; $P --log2n 16 --steps 100 --warmup 10 | tail -1
  echo "== 1024 x 4096"; $P --log2n 12 --columns 1024 --steps 20 --warmup 3 | tail -1
  echo "== bls12-381 2^22"; $P --curve 1 --log2n 22 --steps 10 --warmup 2 | tail -1
  tools/ubench/bin/tail_latency 2>/dev/null | tail -30
} > "$OUT/slow_box.log" 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-file "$OUT/bench_detail.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -40 "$OUT/slow_box.log"
