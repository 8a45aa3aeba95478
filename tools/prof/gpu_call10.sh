#!/bin/bash
set -u
OUT=gpurun_out/r3c10
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -12 $OUT/pytest_gpu.txt
python tools/skew_bench.py > $OUT/skew.txt 2>&1; tail -12 $OUT/skew.txt
