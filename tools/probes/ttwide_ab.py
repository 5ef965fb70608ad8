"""A/B of the wide (256 x 128) tiles of pq3d_gemm_tt_multi inside the step: bench headline with the switch on / off.
usage: python tools/probes/ttwide_ab.py <config> <0|1>"""
import subprocess, sys, os, json
cfg, on = sys.argv[1], sys.argv[2]
code = f"""
import sys, runpy
sys.path.insert(0, '/root/repo')
from pq3d_amd import _lib
_lib.lib().pq3d_gemm_tt_multi_wide({on})
sys.argv = ['bench.py', '--config', '{cfg}', '--headline-only', '--cpu-steps', '0', '--steps', '20', '--warmup', '5']
runpy.run_path('/root/repo/bench.py', run_name='__main__')
"""
p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
try:
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(f"RESULT {cfg} wide={on} {r['ms_per_step']:.4f}")
except Exception as e:
    print("RESULT fail", e, p.stderr[-800:])
