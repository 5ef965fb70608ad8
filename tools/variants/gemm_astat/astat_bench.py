"""Probe: the hoisted K/V projection at config-2 / config-5 shapes -- A-stationary kernel (gemm_astat.hip) against the
128 x 128 per-tile kernel (gemm128.hip), replayed back to back in a HIP graph."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
dev = 'cuda'
def bench(M, fams, per, N=256, K=256, n=20):
    As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(fams)]
    G = fams * per
    W = [(torch.randn(N, K, device=dev) * 0.1).bfloat16() for _ in range(G)]
    b = [torch.randn(N, device=dev) for _ in range(G)]
    C_ = torch.empty(G, M, N, dtype=torch.bfloat16, device=dev)
    def call():
        L.gemm(M=M, N=N, K=K, A=[As[g % fams] for g in range(G)], B=W, bias=b, Cs=[C_[g] for g in range(G)], ct=L.BF16, lda=K, ldb=K, ldc=N)
    res = {}
    for name, opt in (("astat", 1), ("nt128", 1 | (1 << 9))):
        L.lib().pq3d_gemm_set_wk(opt, 0)
        call(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                call()
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t) / 5 / n * 1e6
    L.lib().pq3d_gemm_set_wk(1, 0)
    flops = 2.0 * M * N * K * G
    byts = fams * M * K * 2 + G * (N * K * 2 + M * N * 2)
    print(f"M={M} fams={fams} per={per}: astat {res['astat']:.1f} us ({flops / res['astat'] / 1e6:.0f} TFLOP/s, {byts / res['astat'] / 1e6:.2f} TB/s algorithmic)"
          f"  nt128 {res['nt128']:.1f} us ({flops / res['nt128'] / 1e6:.0f} TFLOP/s)", flush=True)
bench(8192, 6, 4)      # c2: 3 memories x (k, v), 4 layers
bench(16384, 6, 4)     # c4
bench(32768, 6, 3)     # c5: B16 x 2048, 6 layers = two launches of <= 32 groups in the product path (18 groups each)
bench(8192, 6, 1)
