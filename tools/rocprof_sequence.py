#!/usr/bin/env python3
"""Print the kernel sequence of the LAST `n` dispatches of a rocprofv3 (rocpd sqlite) kernel trace, in start order,
with each kernel's duration and the idle gap before it -- one replayed step of the HIP graph when n = dispatches/step.
    python tools/rocprof_sequence.py <results.db> <n> [skip_from_end]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:90]


def main():
    db, n = sys.argv[1], int(sys.argv[2])
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[len(rows) - n - skip: len(rows) - skip]
    prev_end = rows[0][1]
    busy = gap_tot = 0.0
    for i, (name, s, e) in enumerate(rows):
        gap = (s - prev_end) / 1e3
        print(f"{i:4d} {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {short(name)}")
        busy += (e - s) / 1e3
        gap_tot += max(gap, 0.0)
        prev_end = max(prev_end, e)
    print(f"# span {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us, kernel time {busy:.1f} us, idle gaps {gap_tot:.1f} us")


if __name__ == "__main__":
    main()
