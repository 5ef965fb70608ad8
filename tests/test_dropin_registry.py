"""Build container only (skipped where /root/reference is absent, e.g. on the GPU box): INTEGRATION.md section 1 is
EXECUTED -- the HIP module classes are registered into the reference's own fvcore-style registries and the reference's own
Query3DUnified is built around them from a config whose module names were swapped, loads the all-reference model's
state_dict strictly and passes get_opt_params().  Runs tests/dropin_registry_check.py in a process of its own (the import
recipe seeds sys.modules with the reference's package names).  CPU only; no reference code is copied or shipped."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/modules"), reason="the reference is only present in the build container")
def test_hip_modules_register_into_the_reference_registries_and_build_its_model():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_registry_check.py")], cwd=ROOT,
                       env=dict(os.environ, OMP_NUM_THREADS="4"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "DROPIN-OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
