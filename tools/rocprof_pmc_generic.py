#!/usr/bin/env python3
"""Average value of every collected counter per kernel from a rocprofv3 --pmc rocpd database.
    python tools/rocprof_pmc_generic.py <results.db> [kernel substring]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, n, v in sorted(rows):
    if sub in k:
        print(f"{k[:60]:60s} {c:28s} n={n:4d} avg={v:16.1f}")
