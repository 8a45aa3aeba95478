#!/bin/bash
# round 4, eighteenth GPU session: the row pipeline with two upload threads and whole columns behind
# the lead columns -- parity, the drop-in host path's four cases, timelines of two of them
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4r
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_pipeline or blocking or host or generators" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -4 $OUT/pytest.txt
for rep in 1 2; do
tools/pipeline_bench/_build/hostapi_bench --samples 10 --warmup 2 > $OUT/hostapi_bench_$rep.json 2>&1
python - $OUT/hostapi_bench_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for c in d["cases"]:
    print(c["columns"], c["generators"][:8], "mean %.3f min %.3f median %.3f" % (c["ms_mean"], c["ms_min"], c["ms_median"]))
print("agree:", d["caller_and_builtin_generators_agree"])
PY
done
cd /tmp && export TMPDIR=/tmp
for spec in "1 caller" "10 caller"; do
  tag=$(echo $spec | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_$tag -o r -- \
      $REPO/tools/pipeline_bench/_build/hostapi_bench --samples 3 --warmup 2 --only $spec > $OUT/run_$tag.txt 2>&1
  python $REPO/tools/prof/hostapi_timeline.py $OUT/trace_$tag > $OUT/timeline_$tag.txt 2>&1
  head -2 $OUT/timeline_$tag.txt
  rm -rf $OUT/trace_$tag
done
