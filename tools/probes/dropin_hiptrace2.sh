# stream arguments of the per-parameter hipEventRecord / hipStreamWaitEvent calls: bash tools/probes/dropin_hiptrace2.sh <config> <mode>
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ht
timeout 600 rocprofv3 --hip-runtime-trace -d /tmp/ht -o t -- python $R/tools/probes/dropin_event_probe.py $1 $2 2>&1 | grep -E "^c[0-9]|^handles"
DB=$(find /tmp/ht -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(region_args)")]
print("region_args", cols)
rows = cur.execute("select r.name, a.name, a.value, r.tid from regions r join region_args a on a.id = r.id where r.name in ('hipStreamWaitEvent','hipEventRecord','hipGraphLaunch') ").fetchall() if "id" in cols else []
if not rows:
    # try event_id linkage
    key = "event_id" if "event_id" in cols else cols[0]
    rows = cur.execute(f"select r.name, a.name, a.value, r.tid from regions r join region_args a on a.{key} = r.event_id where r.name in ('hipStreamWaitEvent','hipEventRecord','hipGraphLaunch')").fetchall()
c = collections.Counter((n, an, v, t) for n, an, v, t in rows if an in ("stream", "hStream"))
for k, v in c.most_common(20): print("  ", k, v)
PY
