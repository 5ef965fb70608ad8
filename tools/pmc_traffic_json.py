#!/usr/bin/env python3
"""HBM traffic per GPU kernel as JSON, from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in separate
runs of the same command; rocpd sqlite output):
    python tools/pmc_traffic_json.py fetch.db write.db > profiles/pmc_traffic_<round>_<config>.json
bench.py reads that file to fill ``roofline.traffic`` for the dominant entry point.  Counter units are KiB.
MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts a 128-byte read request of a wide coalesced stream as
64 B, so the true read traffic lies between FETCH_SIZE and 2 x FETCH_SIZE; both sums are kept."""
import json
import re
import sqlite3
import sys


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


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size, count(*), avg(value) from counters_collection "
                       "where counter_name=? group by kernel_name, grid_size", (counter,)).fetchall()
    return {(r[0], r[1]): (r[2], r[3]) for r in rows}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fc, fa = f.get(k, (0, 0.0))
        wc, wa = w.get(k, (0, 0.0))
        e = out.setdefault(short(k[0]), [])
        e.append({"full_name": k[0][:160], "grid": k[1], "launches": max(fc, wc), "fetch_kib": round(fa, 1), "write_kib": round(wa, 1)})
    # calibration launch (bench.py --pmc-calibration): copy_many_kernel streaming 100 MiB in and 100 MiB out with 16-byte
    # accesses per lane -- the largest copy_many row; the step's own gradient pack moves < 40 MB
    cal = None
    rows = [r for r in out.get("copy_many_kernel", []) if r["fetch_kib"] + r["write_kib"] > 60 * 1024]
    if rows:
        true_kib = 100 * 1024
        r = min(rows, key=lambda r: abs(r["write_kib"] - true_kib))     # WRITE_SIZE is exact: the row that wrote 100 MiB
        cal = {"kernel": "copy_many_kernel", "grid": r["grid"], "true_read_kib": true_kib, "true_write_kib": true_kib,
               "fetch_kib_raw": r["fetch_kib"], "write_kib_raw": r["write_kib"],
               "fetch_raw_over_true": round(r["fetch_kib"] / true_kib, 4), "write_raw_over_true": round(r["write_kib"] / true_kib, 4)}
    print(json.dumps({"_build": _build_ids(), "calibration": cal, "_note": "per launch, KiB; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py "
                               "--no-graph --headline-only` (tools/refresh_profiles.sh)", "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
