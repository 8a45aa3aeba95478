#!/bin/bash
# kernel timeline of a few bench.py steps (rocprofv3 --kernel-trace): per queue, start offset, gap
# to the previous kernel of that queue and duration -- shows stream bubbles and what overlaps
REPO=$(pwd)
OUT=$REPO/gpurun_out/trace_gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-configs > $OUT/log.txt 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/trace_gaps/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'],
                     r['Kernel_Name'].replace('void ', '').replace('bz::', '')[:34]))
rows.sort()
# the timed region = the first long run of k_accumulate<ed25519_msm> launches: steps 6..9 of it
acc = [i for i, r in enumerate(rows) if r[3].startswith('k_accumulate<ed25519_msm')]
lo, hi = acc[3 + 5], acc[3 + 9]
t0 = rows[lo][0]
last_end = {}
for s, e, q, n in rows[lo - 8:hi]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    print(f"q{q} t={(s - t0)/1e3:9.1f} us  gap {gap:7.1f}  dur {(e - s)/1e3:7.1f}  {n}")
    last_end[q] = e
PY
