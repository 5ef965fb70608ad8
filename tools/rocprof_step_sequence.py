#!/usr/bin/env python3
"""Find the replayed HIP-graph step in a rocprofv3 kernel trace (the shortest exactly repeating kernel-name period in the
middle of the trace) and print its kernels in order with their durations.
    python tools/rocprof_step_sequence.py <results.db> [--small]   (--small: only kernels under 8 us)"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:84]


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    names = [r[0] for r in rows]
    mid = len(rows) // 2
    # period = distance between consecutive occurrences of the RAREST kernel of the replayed region (a kernel launched once
    # per step), verified as an exact repetition -- the shortest repeating window is not the step once identical decoder /
    # caption layers repeat exactly inside it
    import collections
    lo, hi = len(rows) // 4, 3 * len(rows) // 4
    cnt = collections.Counter(names[lo:hi])
    N = None
    for nm, c in sorted(cnt.items(), key=lambda kv: kv[1]):
        if c < 3:
            continue
        idx = [i for i in range(lo, hi) if names[i] == nm]
        dist = {b - a for a, b in zip(idx, idx[1:])}
        if len(dist) == 1:
            n = dist.pop()
            m0 = idx[len(idx) // 2]
            if n >= 20 and names[m0:m0 + n] == names[m0 + n:m0 + 2 * n]:
                N, mid = n, m0
                break
    if N is None:
        N = next(n for n in range(20, 4000) if names[mid:mid + n] == names[mid + n:mid + 2 * n] == names[mid + 2 * n:mid + 3 * n])
    gaps = [(rows[i][1] - rows[i - 1][2], i) for i in range(mid, mid + N)]
    i0 = max(gaps)[1]
    small_only = "--small" in sys.argv
    busy = 0.0
    for k in range(N):
        n, s, e = rows[i0 + k]
        us = (e - s) / 1e3
        busy += us
        if not small_only or us < 8.0:
            print(f"{k:4d} {us:7.1f}  {short(n)}")
    print(f"# {N} dispatches per step, kernel time {busy:.1f} us, span {(rows[i0 + N - 1][2] - rows[i0][1]) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
