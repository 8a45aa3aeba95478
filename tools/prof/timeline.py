#!/usr/bin/env python3
"""Print the kernel timeline of a rocprofv3 --kernel-trace csv around the n-th k_accumulate launch
(start / end / duration in us relative to that launch, queue, kernel): which stages of consecutive
calls really overlap in the throughput mode.

    python tools/prof/timeline.py <r_kernel_trace.csv> [n] [kernels before] [kernels after]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    nth = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    before = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    after = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    ev = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void ", "").replace("bz::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:44], r.get("Queue_Id")))
    ev.sort()
    idx = [i for i, e in enumerate(ev) if e[2].startswith("k_accumulate")]
    i0 = idx[min(nth, len(idx) - 1)]
    t0 = ev[i0][0]
    for e in ev[max(0, i0 - before):i0 + after]:
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f} q={e[3]} {e[2]}")


if __name__ == "__main__":
    main()
