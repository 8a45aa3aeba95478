#!/usr/bin/env python3
"""Timeline of ONE blocking sxt_* call with host buffers: kernels and memory copies of a
`rocprofv3 --kernel-trace --memory-copy-trace` run of tools/pipeline_bench/hostapi_bench, merged and
printed relative to the first event of the LAST call in the trace (calls are separated by the host's
work between them: a gap of more than `gap_us` with nothing on the device).

    python tools/prof/hostapi_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [gap_us]
"""
import csv
import glob
import os
import sys


def rows(pattern, root):
    out = []
    for path in glob.glob(os.path.join(root, "**", pattern), recursive=True):
        out += list(csv.DictReader(open(path)))
    return out


def main():
    root = sys.argv[1]
    gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
    ev = []
    for r in rows("*kernel_trace.csv", root):
        name = r["Kernel_Name"].replace("void ", "").replace("bz::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q" + str(r.get("Queue_Id")), name[:52]))
    for r in rows("*memory_copy_trace.csv", root):
        what = r.get("Direction") or r.get("Name") or "copy"
        size = r.get("Bytes") or r.get("Size") or ""
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", f"{what} {size}"))
    ev.sort()
    # calls: split where the device idles for more than gap_us
    calls, cur, busy_until = [], [], None
    for e in ev:
        if busy_until is not None and e[0] - busy_until > gap_us * 1e3:
            calls.append(cur)
            cur = []
        cur.append(e)
        busy_until = e[1] if busy_until is None else max(busy_until, e[1])
    calls.append(cur)
    last = calls[-1]
    t0 = last[0][0]
    print(f"{len(calls)} groups of device work; the last one: {len(last)} events, "
          f"{(max(e[1] for e in last) - t0) / 1e6:.3f} ms from first start to last end")
    copy_busy = sum(e[1] - e[0] for e in last if e[2] == "C") / 1e6
    print(f"memory copies: {sum(1 for e in last if e[2] == 'C')} events, {copy_busy:.3f} ms busy")
    for e in last:
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f}  {e[2]:5s} {e[3]}")


if __name__ == "__main__":
    main()
