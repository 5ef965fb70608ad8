"""bench.py with the whole-K GEMM option word set first (A/B of a runtime switch in one process each):
python tools/probes/bench_wkopt.py <options> <bench.py arguments...>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pq3d_amd import _lib
_lib.lib().pq3d_gemm_set_wk(int(sys.argv[1], 0), 2048)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
