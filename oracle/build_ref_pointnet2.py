"""TEST INFRASTRUCTURE (oracle side): builds the REFERENCE's own PointNet++ extension for gfx950 into oracle/_ref/.

The reference's only native code is modules/third_party/pointnet2/_ext_src (4 CUDA C files + their pybind wrappers,
SURVEY 8f-4).  They are plain CUDA C over ATen, so the image's PyTorch-ROCm toolchain builds them unmodified:
torch.utils.cpp_extension.load (which applies torch.utils.hipify, its standard translation step for CUDA extensions on
ROCm, and drives hipcc) is run on a TRANSIENT scratch copy of that directory in the system temp dir -- cpp_extension
translates in place and /root/reference must not be written to; the copy is deleted when the build ends, nothing of it
enters the repository or its history -- and the result is linked into
oracle/_ref/pq3d_ref_pointnet2.so, a Python extension exposing the reference's own entry points (furthest_point_sampling,
ball_query, group_points(+grad), gather_points(+grad), three_nn, three_interpolate(+grad)).

No stand-in headers or libraries are written: the extension needs only torch's headers and ROCm, both in the image.
The .so travels to the GPU box with the snapshot; oracle/run_ref_pointnet2.py executes it there on seeded inputs and
writes tests/golden/F18_pointnet2_ref.npz, which pins pq3d_amd/csrc/pointnet2.hip by EXECUTION of the reference's kernels.

    python oracle/build_ref_pointnet2.py          # needs /root/reference; a no-op message elsewhere
"""
from __future__ import annotations

import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/modules/third_party/pointnet2/_ext_src"
OUT_DIR = os.path.join(HERE, "_ref")
BUILD = os.path.join(OUT_DIR, "pointnet2_build")
NAME = "pq3d_ref_pointnet2"


def build(verbose: bool = False) -> str | None:
    target = os.path.join(OUT_DIR, NAME + ".so")
    if not os.path.isdir(REF_SRC):
        print(f"[oracle/_ref] {REF_SRC} not present: keeping the prebuilt {target}" if os.path.exists(target)
              else f"[oracle/_ref] {REF_SRC} not present and no prebuilt extension")
        return target if os.path.exists(target) else None
    newest = max(os.path.getmtime(p) for p in glob.glob(REF_SRC + "/*/*"))
    if os.path.exists(target) and os.path.getmtime(target) >= newest:
        return target
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    import tempfile
    from torch.utils.cpp_extension import load
    # torch.utils.cpp_extension translates CUDA extension sources IN PLACE (x.cu -> x.hip next to it), and /root/reference
    # must not be written to: the build therefore runs on a transient scratch copy of the extension's source directory
    # (system temp dir, removed afterwards); only the linked .so lands in oracle/_ref/
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="pq3d_ref_pointnet2_") as tmp:
        work = os.path.join(tmp, "_ext_src")
        shutil.copytree(REF_SRC, work)
        for root, _d, files in os.walk(work):   # the reference tree is read-only; the scratch copy must be writable
            os.chmod(root, 0o755)
            for f in files:
                os.chmod(os.path.join(root, f), 0o644)
        bdir = os.path.join(tmp, "build")
        os.makedirs(bdir)
        srcs = sorted(glob.glob(work + "/src/*.cpp") + glob.glob(work + "/src/*.cu"))
        load(name=NAME, sources=srcs, extra_include_paths=[os.path.join(work, "include")], build_directory=bdir,
             verbose=verbose, extra_cflags=["-O2"], extra_cuda_cflags=["-O2"], is_python_module=False)   # build only: the .so is
        # imported (= the reference's code executed) nowhere but in oracle/run_ref_pointnet2.py
        shutil.copy2(os.path.join(bdir, NAME + ".so"), target)
    print(f"[oracle/_ref] built {target}")
    return target


if __name__ == "__main__":
    sys.exit(0 if build(verbose="-v" in sys.argv) or not os.path.isdir(REF_SRC) else 1)
