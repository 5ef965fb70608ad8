"""In-graph latency of the small-M GEMM launches of config 2 under the whole-K kernel's tile plans vs the 64x64-tile pipeline
kernel: each shape is launched 40x back to back inside one captured HIP graph (dependent launches, as in the step)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from pq3d_amd import _lib as L

DEV = "cuda"
X3, BF = L.BF16X3, L.BF16
SHAPES = [  # name, M, N, K, groups, kconcat, ct, A dtype, transB, A2, splitk, act_grad, C dtype
    ("qproj  g3 x3 a2", 800, 256, 256, 3, 0, X3, "f32", False, True, 1, None, "bf16"),
    ("qkv    g3 x3 a2", 800, 256, 256, 3, 0, X3, "f32", False, True, 1, None, "f32"),
    ("fc     g1 x3   ", 800, 256, 256, 1, 0, X3, "f32", False, False, 1, None, "f32"),
    ("ffn1   N2048 x3", 800, 2048, 256, 1, 0, X3, "f32", False, False, 1, None, "f32"),
    ("ffn2   K512g4 x3", 800, 256, 512, 4, 0, X3, "f32", False, False, 1, None, "f32"),
    ("ffn2   K256g8 x3", 800, 256, 256, 8, 0, X3, "f32", False, False, 1, None, "f32"),
    ("outprj g3 bf16A", 800, 256, 256, 3, 0, BF, "bf16", False, False, 1, None, "f32"),
    ("dhp    N2048 T ", 800, 2048, 256, 1, 0, BF, "f32", True, False, 1, "relu", "bf16"),
    ("dx2    K2048 s4", 800, 256, 2048, 1, 0, BF, "bf16", True, False, 4, None, "f32"),
    ("dxn    g3 k3 T ", 800, 256, 256, 3, 3, BF, "bf16", True, False, 1, "add", "f32"),
    ("enc    M8992 x3", 8992, 256, 256, 1, 0, X3, "f32", False, False, 1, None, "f32"),
    ("objenc M8192g3x3", 8192, 256, 256, 3, 0, X3, "f32", False, False, 1, None, "f32"),
]
OPTS = [("old 64x64", 0), ("auto", 1), ("auto hw-order", 1 | (1 << 9)), ("32/256", 1 | (1 << 4) | (2 << 6) | (1 << 8)), ("64/256", 1 | (2 << 4) | (2 << 6) | (1 << 8)),
        ("32/128", 1 | (1 << 4) | (1 << 6) | (1 << 8)), ("64/128", 1 | (2 << 4) | (1 << 6) | (1 << 8))]
td = lambda n: torch.bfloat16 if n == "bf16" else torch.float32


def run(shape, opt, reps=40):
    name, M, N, K, G, kc, ct, adt, tb, a2, sk, ag, cdt = shape
    A = [torch.randn(M, K, device=DEV).to(td(adt)) for _ in range(G)]
    A2 = [torch.randn(M, K, device=DEV) for _ in range(G)] if a2 else None
    W = [torch.randn(K, N, device=DEV) if tb else torch.randn(N, K, device=DEV) for _ in range(G)]
    nout = G // kc if kc else G
    C_ = torch.zeros(nout, M, N, dtype=td(cdt), device=DEV)
    Cs = [C_[g // kc] if (kc and g % kc == 0) else (C_[g] if not kc else None) for g in range(G)]
    aux = [torch.randn(M, N, device=DEV).to(td(cdt if ag == "relu" else "f32")) if (not kc or g % kc == 0) else None for g in range(G)] if ag else None
    bias = None if (ag or sk > 1) else [torch.randn(N, device=DEV) if (not kc or g % kc == 0) else None for g in range(G)]
    L.lib().pq3d_gemm_set_wk(opt, 1 << 20)

    def call():
        L.gemm(M=M, N=N, K=K, A=A, A2=A2, B=W, bias=bias, Cs=Cs, aux=aux, ct=ct, lda=K, ldb=N if tb else K, ldc=N, transB=tb,
               kconcat=kc, splitk=sk, act_grad=ag, accumulate=sk > 1)
    call(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                call()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


if __name__ == "__main__":
    print(f"{'shape':18s}" + "".join(f"{n:>14s}" for n, _ in OPTS))
    for sh in SHAPES:
        row = []
        for n, o in OPTS:
            try:
                row.append(f"{run(sh, o):14.2f}")
            except Exception as e:  # noqa
                row.append(f"{'err':>14s}")
        print(f"{sh[0]:18s}" + "".join(row), flush=True)
    L.lib().pq3d_gemm_set_wk(1, 2048)
