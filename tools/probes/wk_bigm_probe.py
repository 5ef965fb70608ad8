"""Probe: the whole-K small-M GEMM kernels (gemm_wk.hip) on the BIG-M query-side products of the shipped stage-2 shape
(M = 128 x 80 = 10240 rows): option bit 8 (several rounds of workgroups) + a large max_m, against the 64 x 64 pipeline kernel.
    python tools/probes/wk_bigm_probe.py <config> <options> <max_m>"""
import json, os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pq3d_amd import _lib as L
cfg, opt, mm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L.lib().pq3d_gemm_set_wk(opt, mm)
import bench
sys.argv = ["bench.py", "--config", cfg, "--headline-only", "--cpu-steps", "0", "--steps", "15", "--warmup", "4", "--profile-steps", "1"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
r = json.loads(buf.getvalue().strip().splitlines()[-1])
print(cfg, "options", opt, "max_m", mm, "->", round(r["ms_per_step"], 3), "ms/step")
