#!/usr/bin/env python3
"""Timeline of one LANED step (two graphs on two streams, pq3d_amd/overlap.py) from a rocprofv3 kernel trace: every kernel
of a step in the middle of the trace with its start offset, duration and the queue it ran on, plus per-queue busy time and
the overlap between the queues.
    python tools/rocprof_lane_timeline.py <results.db> [--brief]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:70]


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    rows = cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    # epoch bumps: the first kernel of graph A and of graph B of every step
    bumps = [i for i, r in enumerate(rows) if "lane_bump_kernel" in r[0]]
    if len(bumps) < 8:
        print("no laned steps in this trace (lane_bump_kernel not found); columns:", cols)
        return
    # steps are delimited by the bumps on the queue that carries the forward (the queue whose bump is followed by more kernels)
    qs = {}
    for i in bumps:
        qs.setdefault(rows[i][3], []).append(i)
    counts = {q: sum(1 for r in rows if r[3] == q) for q in qs}
    qmain = max(counts, key=counts.get)
    bm = qs[qmain]
    k = len(bm) // 2
    t0, t1 = rows[bm[k]][1], rows[bm[k + 1]][1]
    step = [r for r in rows if t0 <= r[1] < t1]
    brief = "--brief" in sys.argv
    busy = {}
    for n, s, e, q in step:
        busy.setdefault(q, []).append((s, e))
        if not brief:
            print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  {'main' if q == qmain else 'lane'}  {short(n)}")
    tot = {q: sum(e - s for s, e in v) / 1e3 for q, v in busy.items()}
    # overlap: time during which both queues have a kernel in flight
    ev = []
    for q, v in busy.items():
        for s, e in v:
            ev += [(s, 1, q), (e, -1, q)]
    ev.sort()
    act, last, both, anyb = {}, None, 0, 0
    for t, d, q in ev:
        if last is not None:
            n_act = sum(1 for x in act.values() if x > 0)
            if n_act >= 2:
                both += t - last
            if n_act >= 1:
                anyb += t - last
        act[q] = act.get(q, 0) + d
        last = t
    print(f"# step span {(t1 - t0) / 1e3:.1f} us; kernels: " + ", ".join(f"{'main' if q == qmain else 'lane'} {len(v)} ({tot[q]:.1f} us busy)"
                                                                             for q, v in busy.items())
          + f"; both queues busy {both / 1e3:.1f} us; any busy {anyb / 1e3:.1f} us")
    la = [v for q, v in busy.items() if q != qmain]
    if la:
        print(f"# lane: first kernel at +{(min(s for s, _ in la[0]) - t0) / 1e3:.1f} us, last ends at +{(max(e for _, e in la[0]) - t0) / 1e3:.1f} us; "
              f"main's last kernel ends at +{(max(e for _, e in busy[qmain]) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
