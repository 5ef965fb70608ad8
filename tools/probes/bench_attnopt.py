"""bench.py with the attention kernel-selection word set first: python tools/probes/bench_attnopt.py <word> <bench.py arguments...>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pq3d_amd import _lib
_lib.lib().pq3d_attn_resident(int(sys.argv[1], 0))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
