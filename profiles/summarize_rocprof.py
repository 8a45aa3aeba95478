#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) as markdown.

    python profiles/summarize_rocprof.py gpurun_out/prof1/r1_results.db "<command>" > profiles/x.md
"""
import sqlite3
import sys


def main():
    db, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    c = sqlite3.connect(db)
    print(f"# rocprofv3 --kernel-trace --stats\n\ncommand: `{cmd}`\n")
    print("| kernel | calls | total_us | avg_us | % | grid | wg | lds_B | scratch_B | vgpr | sgpr |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    rows = list(c.execute(
        "select name, count(*), sum(duration)/1000.0, avg(duration)/1000.0, max(grid_x), "
        "max(grid_y), max(workgroup_x), max(lds_size), max(scratch_size), max(vgpr_count), "
        "max(sgpr_count) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")[:70]
        print(f"| {name} | {r[1]} | {r[2]:.1f} | {r[3]:.1f} | {100 * r[2] / total:.1f} | "
              f"{r[4]}x{r[5]} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")


if __name__ == "__main__":
    main()
