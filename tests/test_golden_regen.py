"""Build container only (skipped where /root/reference is absent, e.g. on the GPU box): the committed generator script,
run against the reference as it lies under /root/reference, reproduces every committed fixture -- the fixtures are what
tests/golden/make_golden.py says they are, nothing else."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference/modules"), reason="the reference is only present in the build container")
def test_generator_reproduces_committed_fixtures(tmp_path):
    env = dict(os.environ, PQ3D_GOLDEN_OUT=str(tmp_path), OMP_NUM_THREADS="8")
    p = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    # F18 is not made by this generator: it holds outputs of the reference's own GPU kernels (oracle/run_ref_pointnet2.py on an
    # MI355X box; checked against the oracle by tests/test_pointnet2.py)
    names = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz") and f != "F18_pointnet2_ref.npz")
    assert names == sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    for f in names:
        a, b = np.load(os.path.join(GOLDEN, f)), np.load(os.path.join(tmp_path, f))
        assert set(a.files) == set(b.files), f
        for k in a.files:
            x, y = a[k], b[k]
            if x.dtype.kind == "f":      # bit-identical here; the tolerance only absorbs a different BLAS thread count
                fin = np.isfinite(x)
                assert x.shape == y.shape and np.array_equal(fin, np.isfinite(y)), (f, k)
                assert np.allclose(x[fin], y[fin], rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(x[fin]).max()) if fin.any() else 1.0)), (f, k)
            else:
                assert np.array_equal(x, y), (f, k)
