#!/bin/bash
# round 4, twentieth GPU session: buckets per k_reduce lane for lone calls (2^s, s = 1, 2 against the
# planner's 3): k_reduce alone and k_horner alone at 2^14 rows (tools/prof/horner_phases.py), and
# the lone calls of the bench's configs 2 / 3
set -u
OUT=gpurun_out/r4t
mkdir -p $OUT
for s in 0 2 1; do
  if [ $s = 0 ]; then export -n BLITZAR_AMD_REDUCE_SEGMENT_LOG2; unset BLITZAR_AMD_REDUCE_SEGMENT_LOG2; else export BLITZAR_AMD_REDUCE_SEGMENT_LOG2=$s; fi
  echo "== s=$s (0: the planner's choice)"
  timeout 250 python tools/prof/horner_phases.py --log2n 20 > $OUT/phases_s$s.txt 2>&1
  python - "$OUT/phases_s$s.txt" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: (v["32"]["reduce_ms"], v["32"]["combine_ms"]) for k, v in d.items()}, "(reduce, combine) alone, 2^20 rows, 32-byte scalars")
PY
done
