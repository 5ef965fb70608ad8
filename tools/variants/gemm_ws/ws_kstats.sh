# kernel-only durations (rocprofv3 kernel trace) of the projection kernels: nt128 (mode 0) against the weight-stationary modes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "24 8192" "24 16384" "32 32768"; do
for mode in 0 3; do
  rm -rf /tmp/k1
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/k1 -o k -- python $R/tools/probes/ws_prof.py $mode $shape > /dev/null 2>&1
  DB=$(find /tmp/k1 -name "*.db" | head -1)
  python - "$DB" "$mode" "$shape" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("kernels")] 
rows = cur.execute("select name, count(*), avg(end-start), min(end-start) from kernels where name like '%gemm_%' group by name").fetchall()
for n, c, a, m in rows: print(f"mode {sys.argv[2]} shape {sys.argv[3]}: {n[:48]:48s} n={c} avg {a/1e3:.1f} us min {m/1e3:.1f} us")
PY
done; done
