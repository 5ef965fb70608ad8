"""CPU: the C-ABI library loads, exports every symbol include/pq3d_hip.h declares, the ctypes structs agree with
the header's layout, the host modules mirror the reference's state_dict, and the product path fails loudly
(no CPU / oracle fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

from pq3d_amd import _lib, build
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pq3d_hip.h")


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.lib()


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pq3d_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    declared = header_functions()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pq3d_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "ctypes binding and header disagree on the entry-point set"
    assert lib.pq3d_version() >= 1


def test_struct_layouts_match_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with the ctypes Structures."""
    csrc = tmp_path / "layout.c"
    csrc.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "pq3d_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_gemm_desc), offsetof(pq3d_gemm_desc, alpha), offsetof(pq3d_gemm_desc, A),
         offsetof(pq3d_gemm_desc, mask_out));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_attn_desc), offsetof(pq3d_attn_desc, scale), offsetof(pq3d_attn_desc, q),
         offsetof(pq3d_attn_desc, dbias));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_ln_desc), offsetof(pq3d_ln_desc, eps), offsetof(pq3d_ln_desc, x),
         offsetof(pq3d_ln_desc, dbeta));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_chain_ffn_desc), offsetof(pq3d_chain_ffn_desc, eps2), offsetof(pq3d_chain_ffn_desc, o_s),
         offsetof(pq3d_chain_ffn_desc, err));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_chain_ca_desc), offsetof(pq3d_chain_ca_desc, eps), offsetof(pq3d_chain_ca_desc, o),
         offsetof(pq3d_chain_ca_desc, err));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_chain_ffn_bwd_desc), offsetof(pq3d_chain_ffn_bwd_desc, F), offsetof(pq3d_chain_ffn_bwd_desc, dx),
         offsetof(pq3d_chain_ffn_bwd_desc, err));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_chain_sa_bwd_desc), offsetof(pq3d_chain_sa_bwd_desc, dqkv), offsetof(pq3d_chain_sa_bwd_desc, coef),
         offsetof(pq3d_chain_sa_bwd_desc, err));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_chain_mh_desc), offsetof(pq3d_chain_mh_desc, fill), offsetof(pq3d_chain_mh_desc, Wq),
         offsetof(pq3d_chain_mh_desc, err));
  printf("%zu %zu %zu %zu\n", sizeof(pq3d_chain_mh_bwd_desc), offsetof(pq3d_chain_mh_bwd_desc, dc), offsetof(pq3d_chain_mh_bwd_desc, dq),
         offsetof(pq3d_chain_mh_bwd_desc, gq));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(csrc), "-o", str(exe)])
    rows = [list(map(int, l.split())) for l in subprocess.check_output([str(exe)]).decode().splitlines()]
    G, A, Ln = _lib.GemmDesc, _lib.AttnDesc, _lib.LnDesc
    assert rows[0] == [ctypes.sizeof(G), G.alpha.offset, G.A.offset, G.mask_out.offset]
    assert rows[1] == [ctypes.sizeof(A), A.scale.offset, A.q.offset, A.dbias.offset]
    assert rows[2] == [ctypes.sizeof(Ln), Ln.eps.offset, Ln.x.offset, Ln.dbeta.offset]
    Ch = _lib.ChainFfnDesc
    assert rows[3] == [ctypes.sizeof(Ch), Ch.eps2.offset, Ch.o_s.offset, Ch.err.offset]
    Ca = _lib.ChainCaDesc
    assert rows[4] == [ctypes.sizeof(Ca), Ca.eps.offset, Ca.o.offset, Ca.err.offset]
    Cb = _lib.ChainFfnBwdDesc
    assert rows[5] == [ctypes.sizeof(Cb), Cb.F.offset, Cb.dx.offset, Cb.err.offset]
    Cs = _lib.ChainSaBwdDesc
    assert rows[6] == [ctypes.sizeof(Cs), Cs.dqkv.offset, Cs.coef.offset, Cs.err.offset]
    Cm = _lib.ChainMhDesc
    assert rows[7] == [ctypes.sizeof(Cm), Cm.fill.offset, Cm.Wq.offset, Cm.err.offset]
    Cn = _lib.ChainMhBwdDesc
    assert rows[8] == [ctypes.sizeof(Cn), Cn.dc.offset, Cn.dq.offset, Cn.gq.offset]


def test_argument_errors_are_reported(lib):
    d = _lib.GemmDesc()
    d.groups, d.batch, d.M, d.N, d.K = 999, 1, 4, 4, 4
    rc = lib.pq3d_gemm(ctypes.byref(d), None)
    assert rc == -1 and b"groups" in lib.pq3d_last_error()
    a = _lib.AttnDesc()
    a.B, a.H, a.Lq, a.Lk, a.dh = 1, 1, 4, 4, 48
    assert lib.pq3d_attn_fwd(ctypes.byref(a), None) == -1 and b"head dim" in lib.pq3d_last_error()


def test_comm_entry_points_without_a_gpu(lib):
    """SURVEY 8b's gradient-exchange exports (csrc/comm.hip): the host-only parts answer without a GPU -- the wire form's scratch size,
    argument errors before any RCCL call, and (librccl is bound with dlopen at first use) the rendezvous id."""
    assert lib.pq3d_allreduce_wire_scratch_bytes(8, 1000) == (3 * 8 + 1) * 128 * 2
    assert lib.pq3d_allreduce_wire_scratch_bytes(1, 0) == 0 and lib.pq3d_allreduce_wire_scratch_bytes(0, 5) == -1
    junk = ctypes.create_string_buffer(64)
    assert lib.pq3d_allreduce_grads(junk, None, 8, 0, 1, None) == -1 and b"not a communicator" in lib.pq3d_last_error()
    assert lib.pq3d_comm_destroy(junk) == -1
    h = ctypes.c_void_p()
    assert lib.pq3d_comm_init(2, 2, junk, ctypes.byref(h)) == -1 and b"rank" in lib.pq3d_last_error() and not h.value
    a, b = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
    rc = lib.pq3d_comm_unique_id(a)
    if rc == 0:      # (a host without librccl reports an error text instead)
        assert lib.pq3d_comm_unique_id(b) == 0 and a.raw != b.raw and any(a.raw)
    else:
        assert b"rccl" in lib.pq3d_last_error().lower()


def test_state_dict_keys_match_reference():
    """Every parameter name/shape of the reference (recorded as grad/<name> in the fixtures) exists in our modules."""
    for name in ("F2_c1_mask", "F4_c2_slice", "F5_dimloc6", "F5_offline_mask"):
        z, args = util.load_fixture(name)
        _cfg, model, _sd, _dd = util.model_case(args)
        ref = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
        ours = dict(model.named_parameters())
        assert sorted(ours) == ref, "parameter names differ from the reference's state_dict"
        for n in ref:
            assert tuple(ours[n].shape) == tuple(z[f"grad/{n}/shape"]), n
        assert "coord_encoder.pos_enc.gauss_B" in model.state_dict() or args.get("dim_loc", 3) > 3
        groups = model.get_opt_params()
        assert sum(len(g["params"]) for g in groups) == len(ours)


def test_no_cpu_fallback():
    """CPU tensors must raise (the product path never silently computes on the host or through the oracle)."""
    _z, args = util.load_fixture("F1_c1")
    _cfg, model, _sd, dd = util.model_case(args)
    with pytest.raises(_lib.Pq3dError):
        model(dd)
    for f in os.listdir(os.path.join(ROOT, "pq3d_amd")):
        if f.endswith(".py"):
            text = open(os.path.join(ROOT, "pq3d_amd", f)).read()
            assert "import oracle" not in text and "from oracle" not in text, f"{f} imports the oracle"
