#!/usr/bin/env python3
"""MFMA utilisation per GPU kernel as JSON from one rocprofv3 pass
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python bench.py --no-graph ...
    python tools/pmc_mfma_json.py <results.db> > profiles/pmc_mfma_<round>_<config>.json
SQ_VALU_MFMA_BUSY_CYCLES sums, over every SIMD of the chip, the cycles its matrix pipe was busy (16 per
v_mfma_f32_16x16x32_bf16: the hoisted projection's 1 572 864 MFMAs read 25 165 824, profiles/rocprofv3_pmc_gemm_r01_hoisted.txt).
mfma_busy = busy cycles / (launch duration x 2.4 GHz x 1024 SIMDs) -- with the duration of the SAME (serialised, eager)
launch in this pass; bench.py re-divides the cycles by the in-graph duration of the family."""
import json
import re
import sqlite3
import sys

CLOCK_GHZ, SIMDS = 2.4, 1024


def _build_ids():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from pq3d_amd.build import build_ids
        return build_ids()
    except Exception as e:  # noqa: BLE001
        return {"error": type(e).__name__}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"[<(].*$", "", n)


def main():
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
    acc = {}
    for k, g, c, n, v in rows:
        e = acc.setdefault((k, g), {"launches": n})
        e[c] = v
    out = {}
    for (k, g), e in sorted(acc.items()):
        d = dur.get((k, g)) or dur.get((k, None))
        us = d[1] / 1e3 if d else None
        busy = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out.setdefault(short(k), []).append({
            "full_name": k[:160], "grid": g, "launches": e["launches"], "mfma_busy_cycles": round(busy, 1),
            "sq_busy_cycles": round(e.get("SQ_BUSY_CYCLES", 0.0), 1), "sq_wave_cycles": round(e.get("SQ_WAVE_CYCLES", 0.0), 1),
            "avg_us_this_pass": round(us, 2) if us else None,
            "mfma_busy": round(busy / (us * 1e3 * CLOCK_GHZ * SIMDS), 4) if us else None})
    print(json.dumps({"_build": _build_ids(), "_note": "per launch; SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs; mfma_busy = cycles / (duration x "
                               f"{CLOCK_GHZ} GHz x {SIMDS} SIMDs), duration of the same eager launch in this PMC pass",
                      "clock_ghz": CLOCK_GHZ, "simds": SIMDS, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
