#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max duration.
    python tools/rocprof_summary.py <results.db> [steps]  ->  text table (commit under profiles/).
The header carries `steps N`: bench.py divides the per-kernel totals by it (roofline.in_graph)."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"pq3d_(gemm|attn|ln)_desc", "desc", n)
    return n[:100]


def main():
    db = sys.argv[1]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    # steps in the trace = how often a once-per-step kernel ran (the input encoders' one-launch kernels; else the most common call count)
    # (an explicit second argument overrides it).  NOT dispatches / dispatches-per-step: eager warm-up steps launch a few
    # kernels the replayed graph does not.
    import collections
    marks = [r[1] for r in rows if any(m in r[0] for m in ("fourier_pair_kernel", "pairwise_locs_kernel", "mask_not_kernel"))]
    cnt = collections.Counter(marks or [r[1] for r in rows if r[1] >= 3])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else (float(cnt.most_common(1)[0][0]) if cnt else None)
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    span = cur.execute("select min(start), max(end) from kernels").fetchone()
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from pq3d_amd.build import build_ids
        ids = build_ids()
        print(f"# build: src_sha256 {ids['src_sha256']} lib_sha256 {ids['lib_sha256']}")
    except Exception as e:  # noqa: BLE001
        print(f"# build: unknown ({type(e).__name__})")
    print(f"# {db}: {n} kernel dispatches, total kernel time {tot / 1e6:.2f} ms, trace span {(span[1] - span[0]) / 1e6:.1f} ms"
          + (f", steps {steps:.0f} (calls of a once-per-step kernel), per step: {n / steps:.1f} dispatches, "
             f"{tot / 1e6 / steps:.3f} ms kernel time" if steps else ""))
    print(f"{'kernel':100s} {'calls':>7s} {'total ms':>9s} {'avg us':>8s} {'min us':>8s} {'max us':>8s} {'%':>6s}")
    for r in rows:
        print(f"{short(r[0]):100s} {r[1]:7d} {r[2] / 1e6:9.2f} {r[3] / 1e3:8.1f} {r[4] / 1e3:8.1f} {r[5] / 1e3:8.1f} "
              f"{100 * r[2] / tot:6.1f}")


if __name__ == "__main__":
    main()
