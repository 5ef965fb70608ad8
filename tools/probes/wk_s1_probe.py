"""tools/probes/wk_probe.py at the shipped stage-1 decoder size (M = 4 x 120 = 480 rows, d = 768): in-graph latency of the
small-M GEMM launches under every tile plan of the whole-K kernel and under the 64 x 64 pipeline kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wk_probe as P
X3, BF = P.X3, P.BF
M, D, F = 480, 768, 2048
P.SHAPES = [
    ("qproj  g3 x3 a2", M, D, D, 3, 0, X3, "f32", False, True, 1, None, "bf16"),
    ("qkv    g3 x3 a2", M, D, D, 3, 0, X3, "f32", False, True, 1, None, "f32"),
    ("fc     g1 x3   ", M, D, D, 1, 0, X3, "f32", False, False, 1, None, "f32"),
    ("ffn1   N2048 x3", M, F, D, 1, 0, X3, "f32", False, False, 1, None, "f32"),
    ("ffn2   K512g4 x3", M, D, 512, 4, 0, X3, "f32", False, False, 1, None, "f32"),
    ("outprj g3 bf16A", M, D, D, 3, 0, BF, "bf16", False, False, 1, None, "f32"),
    ("dhp    N2048 T ", M, F, D, 1, 0, BF, "f32", True, False, 1, "relu", "bf16"),
    ("dx2    K2048 s4", M, D, F, 1, 0, BF, "bf16", True, False, 4, None, "f32"),
    ("dxn    g3 k3 T ", M, D, D, 3, 3, BF, "bf16", True, False, 1, "add", "f32"),
    ("dx     g3 T    ", M, D, D, 3, 0, BF, "f32", True, False, 1, None, "f32"),
    ("dx     g1 T    ", M, D, D, 1, 0, BF, "f32", True, False, 1, None, "f32"),
    ("cls    N201 x3 ", M, 201, D, 1, 0, X3, "f32", False, False, 1, None, "f32"),
]
if __name__ == "__main__":
    print(f"{'shape':18s}" + "".join(f"{n:>14s}" for n, _ in P.OPTS))
    for sh in P.SHAPES:
        row = []
        for n, o in P.OPTS:
            try:
                row.append(f"{P.run(sh, o):14.2f}")
            except Exception as e:  # noqa
                row.append(f"{'err':>14s}")
        print(f"{sh[0]:18s}" + "".join(row), flush=True)
    P.L.lib().pq3d_gemm_set_wk(1, 2048)
