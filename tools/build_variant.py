#!/usr/bin/env python3
"""Build a second copy of the kernel library with extra compiler flags, for same-box A/B measurements:
    python tools/build_variant.py <name> [-DFLAG ...]   ->  pq3d_amd/libpq3d_hip_<name>.so
    PQ3D_LIB_PATH=pq3d_amd/libpq3d_hip_<name>.so python bench.py ...   (the variant; default library otherwise)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pq3d_amd import build as B  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
objdir = os.path.join(B.HERE, "build", "variant_" + name)
os.makedirs(objdir, exist_ok=True)
procs = []
for src in B.SOURCES:
    obj = os.path.join(objdir, src + ".o")
    procs.append((obj, subprocess.Popen(["/opt/rocm/bin/hipcc", *B.FLAGS, *B.EXTRA.get(src, []), *extra, "-x", "hip", "-c",
                                        os.path.join(B.CSRC, src), "-o", obj])))
for obj, p in procs:
    assert p.wait() == 0, obj
out = os.path.join(B.HERE, f"libpq3d_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *[o for o, _ in procs]])
print("built", out)
