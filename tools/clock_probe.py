import subprocess, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pq3d_amd import ops
from pq3d_amd._lib import BF16
def smi(tag):
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showperflevel", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        keep = [l for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Performance Level", "Power"))]
        print(tag, " | ".join(l.split(":", 1)[-1].strip() for l in keep)[:300])
    except Exception as e:
        print(tag, "rocm-smi failed", e)
dev="cuda"
x = torch.randn(8,100,256,device=dev); w = torch.randn(256,256,device=dev)*0.05; b=torch.zeros(256,device=dev)
h = torch.randn(8,100,2048,device=dev).bfloat16(); w2 = torch.randn(256,2048,device=dev)*0.05
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n//20): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
smi("idle:")
with torch.no_grad():
    f1 = lambda: ops.linear(x, w, b, ct=BF16, out_dtype=torch.bfloat16)
    f2 = lambda: ops.linear(h, w2, b, ct=BF16)
    print("cold  800x256x256 us", timeit(f1), " 800x2048x256 us", timeit(f2))
    A = torch.randn(8192,8192,device=dev,dtype=torch.bfloat16); Bm = torch.randn(8192,8192,device=dev,dtype=torch.bfloat16)
    t0=time.time()
    for _ in range(60): C = A@Bm
    torch.cuda.synchronize(); print("burn s", time.time()-t0, "TF", 60*2*8192**3/(time.time()-t0)/1e12)
    smi("after burn:")
    print("hot   800x256x256 us", timeit(f1), " 800x2048x256 us", timeit(f2))
    print("long  800x256x256 us", timeit(f1, 20000), " 800x2048x256 us", timeit(f2, 20000))
    smi("after long:")
    # empty-ish torch kernel for launch floor
    z = torch.zeros(16, device=dev)
    print("torch add_ tiny us", timeit(lambda: z.add_(1.0)))
    big = torch.zeros(8*1024*256, device=dev)
    print("torch add_ 8MB us", timeit(lambda: big.add_(1.0)))
