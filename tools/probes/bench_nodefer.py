"""bench.py with the deferred weight-gradient queue of ops.grad_arena switched off (A/B of a Python-level switch):
python tools/probes/bench_nodefer.py <bench.py arguments...>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pq3d_amd import ops
ops._DW_DEFER = False
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
