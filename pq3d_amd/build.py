"""Build libpq3d_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pq3d_amd.build          # incremental: rebuilds only when a source is newer than the .so
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpq3d_hip.so")
SOURCES = ["api.cpp", "gemm.hip", "attention.hip", "norm.hip", "misc.hip", "optim.hip", "loss.hip", "pointnet2.hip", "gemm128.hip", "attn_resident.hip", "attn_small.hip", "gemm_wk.hip", "attn_sa.hip", "gemm_wktt.hip", "gemm_cv128.hip", "attn_ca.hip", "segment.hip", "t5glue.hip", "gemm_ttmulti.hip", "chain_ffn.hip", "chain_ca.hip", "chain_ffn_bwd.hip", "chain_sa_bwd.hip", "chain_mh.hip", "attn_x3.hip", "chain_probe.hip", "gemm_x3p.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-Wno-unused-result"]
# per-file extras.  attn_resident.hip: MFMA results feed the softmax VALU code directly; with the default AGPR form of the
# MFMAs the compiler shuttles every S / dP tile through v_accvgpr_read / _write (352 extra VALU-slot moves per 4 query
# pairs); the VGPR form removes them (gfx950 has one unified register file).
# attn_resident.hip: MFMA results in VGPRs (no accumulator-register copies around the softmax);  -fno-honor-nans drops the
# NaN-canonicalising v_max x,x that every fmaxf otherwise costs (32 of ~150 VALU instructions per 64-key block of the
# forward; the kernels never produce or test NaNs: masked scores are -inf, the running max starts at -1e30)
EXTRA = {"attn_resident.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
         "attn_x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
         "attn_sa.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         "attn_ca.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def build_ids() -> dict:
    """Identity of the build a measurement belongs to: sha256 (first 16 hex digits) of the shared library's bytes and of the sources it
    is built from (csrc/*, the public header, the compile flags).  tools/rocprof_summary.py / pmc_*_json.py write both into every
    profiles/ file; bench.py uses a committed profile only when its `src_sha256` equals the running build's (a stale file would
    otherwise silently yield a roofline fraction)."""
    import hashlib
    hs = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "pq3d_hip.h")]:
        path = os.path.join(CSRC, f)
        if os.path.isfile(path) and (f.endswith((".hip", ".h", ".cpp"))):
            hs.update(os.path.basename(f).encode()); hs.update(open(path, "rb").read())
    hs.update(repr((FLAGS, sorted(EXTRA.items()), SOURCES)).encode())
    lib = hashlib.sha256(open(LIB, "rb").read()).hexdigest()[:16] if os.path.exists(LIB) else None
    return {"src_sha256": hs.hexdigest()[:16], "lib_sha256": lib}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "pq3d_hip.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def _obj_stale(src: str, obj: str) -> bool:
    """An object is rebuilt when its own source, any header under csrc/ or the public header is newer than it."""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src), os.path.join(HERE, "..", "include", "pq3d_hip.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src + ".o")
        objs.append(obj)
        if not force and not _obj_stale(src, obj):
            continue
        cmd = [hipcc, *FLAGS, *EXTRA.get(src, []), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    if verbose:
        print(f"built {LIB} ({len(procs)} of {len(SOURCES)} objects recompiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
