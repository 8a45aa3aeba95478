#!/bin/bash
# round 4, seventeenth GPU session: timelines (kernels + memory copies) of ONE blocking call of the
# drop-in host path: 1 and 10 columns with caller generators, 10 columns with built-in generators
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r4q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "1 caller" "10 caller" "10 builtin"; do
  tag=$(echo $spec | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_$tag -o r -- \
      $REPO/tools/pipeline_bench/_build/hostapi_bench --samples 3 --warmup 2 --only $spec > $OUT/run_$tag.txt 2>&1
  echo "$spec rc=$?"; tail -1 $OUT/run_$tag.txt | cut -c1-300
  python $REPO/tools/prof/hostapi_timeline.py $OUT/trace_$tag > $OUT/timeline_$tag.txt 2>&1
  head -3 $OUT/timeline_$tag.txt
  find $OUT/trace_$tag -name "*.csv" -size +2M -delete
done
cd $REPO
# untraced reference of the same cases
tools/pipeline_bench/_build/hostapi_bench --samples 10 --warmup 2 > $OUT/hostapi_bench.json 2>&1; cut -c1-1200 $OUT/hostapi_bench.json
