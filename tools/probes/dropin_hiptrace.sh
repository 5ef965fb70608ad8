# HIP runtime API calls per drop-in step (rocprofv3 --hip-runtime-trace): bash tools/probes/dropin_hiptrace.sh <config> <mode>
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ht
timeout 600 rocprofv3 --hip-runtime-trace --kernel-trace -d /tmp/ht -o t -- python $R/tools/probes/dropin_event_probe.py $1 $2 2>&1 | grep -E "^c[0-9]"
DB=$(find /tmp/ht -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if "region" in t.lower()]
print("tables:", cand[:8])
for t in ("regions", "regions_and_samples"):
    if t in tabs:
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        print(t, cols)
        namecol = "name" if "name" in cols else cols[1]
        for r in cur.execute(f"select {namecol}, count(*) from {t} group by 1 order by 2 desc limit 25"): print("  ", r)
        break
PY
