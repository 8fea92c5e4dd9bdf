#!/usr/bin/env python3
"""Kernel timeline out of a rocprofv3 rocpd database (`*_results.db`): the repeating launch
sequence of the LAST iterations with each kernel's duration and the gap in front of it.

    python tools/rocpd_timeline.py <results.db> [n_last_kernels]
"""
import re
import sqlite3
import sys


def clean(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main():
    db = sqlite3.connect(sys.argv[1])
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    rows = rows[-n_last:]
    prev_end = None
    for name, start, end, gx, wx in rows:
        short = clean(name)[:70]
        gap = 0.0 if prev_end is None else (start - prev_end) / 1e3
        print(f"{gap:8.1f} us gap | {(end - start) / 1e3:8.1f} us  {short}  grid {gx}/{wx}")
        prev_end = end
    print("stats over the whole run (avg us, count):")
    for name, avg, cnt in db.execute(
            "select name, avg(end - start) / 1e3, count(*) from kernels group by name order by sum(end - start) desc"):
        print(f"  {avg:9.2f} {cnt:6d}  {clean(name)[:90]}", flush=False)


if __name__ == "__main__":
    main()
