#!/usr/bin/env python3
"""LDS bank-conflict share per kernel from one rocprofv3 pass
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python bench.py --no-graph ...
    python tools/pmc_lds_conflicts.py <results.db>
MI355X_MICROARCH.md: SQ_LDS_BANK_CONFLICT = extra LDS-array cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles (summed over the chip)."""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection "
                   "group by kernel_name, grid_size, counter_name").fetchall()
dur = {}
try:
    for name, g, n, a in cur.execute("select name, grid_size, count(*), avg(end-start) from kernels group by name, grid_size"):
        dur[(name, g)] = (n, a)
except sqlite3.Error:
    for name, n, a in cur.execute("select name, count(*), avg(end-start) from kernels group by name"):
        dur[(name, None)] = (n, a)
agg = collections.defaultdict(dict)
for name, g, cname, n, avg in rows:
    agg[(name, g)][cname] = (n, avg)
out = []
for (name, g), c in agg.items():
    act, conf = c.get("SQ_LDS_IDX_ACTIVE", (0, 0.0))[1], c.get("SQ_LDS_BANK_CONFLICT", (0, 0.0))[1]
    n, ns = dur.get((name, g)) or dur.get((name, None)) or (0, 0.0)
    if act > 0:
        out.append((conf, short(name), g, n, ns / 1e3, act, conf / act))
print(f"{'kernel':64s} {'grid':>8s} {'n':>4s} {'avg us':>8s} {'LDS cycles':>12s} {'conflict':>12s} {'share':>6s}")
for conf, name, g, n, us, act, sh in sorted(out, reverse=True):
    print(f"{name[:64]:64s} {g:8d} {n:4d} {us:8.1f} {act:12.0f} {conf:12.0f} {sh:6.2f}")
