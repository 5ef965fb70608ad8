"""CPU: the host logic of bench.py's roofline section -- SURVEY 8(d)'s closed-form step bytes, the (entry point, shape key) ->
GPU kernel rules that attach algorithmic work to the kernels of the in-graph trace, and the build-hash gate on committed profiles."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pq3d_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_step_bytes_is_survey_8d():
    # forward compulsory bytes 133.8 MB at c2 (221.4 MB with the mask head), 434.2 MB at c4, 678.3 MB at c5; STEP = 3 x forward
    c2 = dict(bench.CONFIGS["c2"])
    assert abs(bench.step_bytes(c2) / 3 / 1e6 - 133.8) < 0.1
    assert abs(bench.step_bytes(dict(bench.CONFIGS["c4"])) / 3 / 1e6 - 434.2) < 0.5
    assert abs(bench.step_bytes(dict(bench.CONFIGS["c5"])) / 3 / 1e6 - 678.3) < 0.5


def test_kernel_rules_cover_the_headline_shapes():
    k = bench.kernel_for
    assert k("pq3d_chain_ffn_fwd", "R800d256F2048") == "chain_ffn_fwd_kernel"
    assert k("pq3d_gemm_tt_multi", "ttmulti48") == "gemm_tt_multi_kernel"
    assert k("pq3d_attn_fwd", "B24H8Lq100Lk1024dh32ct1") == "attn_fwd_resident_kernel"
    assert k("pq3d_attn_fwd", "B24H8Lq100Lk1024dh32ct2") == "attn_fwd_x3_kernel"
    assert k("pq3d_attn_fwd", "B12H8Lq200Lk4096dh32ct1m3") == "attn_fwd_kernel"
    assert k("pq3d_attn_fwd", "B8H8Lq100Lk100dh32ct2") == "attn_sa_fwd_kernel"
    assert k("pq3d_attn_bwd", "B24H8Lq100Lk1024dh32ct1") == "attn_bwd_resident_kernel"
    assert k("pq3d_gemm", "M8192N256K256g24b1NNct1") == "gemm_nt128_kernel"
    assert k("pq3d_gemm", "M256N256K8192g24b1TTs2ct1") == "gemm_tt128_kernel"
    assert k("pq3d_gemm", "M800N256K256g3b1NNct2") == "gemm_wk_kernel"
    assert k("pq3d_gemm", "M8192N256K256g3b1NNct2") == "gemm_fast_kernel"


def test_committed_profiles_are_tied_to_the_build(tmp_path, monkeypatch):
    """A profile is used only when its stamped src_sha256 equals the running build's; others are listed as refused."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    good, bad = {"src_sha256": "aaaa", "lib_sha256": "x"}, {"src_sha256": "bbbb", "lib_sha256": "y"}
    (prof / "pmc_traffic_r06_c2.json").write_text(json.dumps({"_build": good, "kernels": {}, "calibration": None}))
    (prof / "pmc_mfma_r06_c2.json").write_text(json.dumps({"_build": bad, "kernels": {}}))
    (prof / "rocprofv3_kernel_stats_r06_fused_graph_c2.txt").write_text(
        "# build: src_sha256 aaaa lib_sha256 x\n# db: 10 kernel dispatches, steps 2 (calls)\nkernel calls total avg min max %\n"
        "foo_kernel<1>(desc)      4      0.08     20.0     19.0     21.0    100.0\n")
    (prof / "pmc_traffic_r05_c2.json").write_text(json.dumps({"kernels": {}}))   # unstamped (an older round): refused
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    pmc, mfma, stats, info = bench.committed_profiles("c2", bench.DEFAULT_COMPUTE, good)
    assert pmc is not None and mfma is None
    assert stats == {"foo_kernel": (2.0, 40.0)}
    assert any("pmc_mfma_r06_c2.json" in r for r in info["refused"])
    pmc2, _m, stats2, info2 = bench.committed_profiles("c2", bench.DEFAULT_COMPUTE, bad)
    assert pmc2 is None and stats2 == {} and len(info2["refused"]) >= 2
