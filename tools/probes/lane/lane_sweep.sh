#!/bin/bash
# sweep the background lane's width and work selection at one config; prints ms/step and the lane timeline summary
CFG=${1:-c2}
for what in all dw kv; do for cus in 32 64 128 192 256; do
  PQ3D_BG_WHAT=$what PQ3D_BG_CUS=$cus timeout 300 python bench.py --config $CFG --headline-only --cpu-steps 0 --steps 40 --profile-steps 1 > /tmp/b.json 2>/tmp/b.err
  python - <<PY
import json
try:
    r=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    tl=r.get("background_lane",{}).get("timeline_us_last_step",[])
    g=[x for x in tl if x["kind"]=="gate"]; j=[x for x in tl if x["kind"]=="join"]
    print("$CFG what=$what cus=$cus ms=%.3f gates@%s lane_idle_before_gate=%s join: lane done %.0f main arrives %.0f" % (r["ms_per_step"], [round(x["published_us"]) for x in g], [round(x["wait_end_us"]-x["wait_begin_us"]) for x in g], j[-1]["published_us"] if j else -1, j[-1]["wait_begin_us"] if j else -1))
except Exception as e:
    print("$CFG what=$what cus=$cus FAILED", e, open("/tmp/b.err").read()[-300:])
PY
done; done
PQ3D_BG_CUS=0 timeout 300 python bench.py --config $CFG --headline-only --cpu-steps 0 --steps 40 --profile-steps 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG no lane ms=%.3f' % r['ms_per_step'])"
