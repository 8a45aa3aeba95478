#!/bin/bash
REPO=$(pwd)
OUT=$REPO/gpurun_out/trace_gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs > $OUT/log.txt 2>&1
cd $REPO
ls -R $OUT | head -20
python3 - <<'PY'
import csv, glob, os
root = 'gpurun_out/trace_gaps'
rows = []
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]))
for f in glob.glob(root + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
# print the last ~40 events (the last timed steps)
last = rows[-45:]
prev_end = last[0][0]
for s, e, n in last:
    print(f"gap {(s - prev_end)/1e3:8.1f} us  dur {(e - s)/1e3:8.1f} us  {n}")
    prev_end = e
PY
