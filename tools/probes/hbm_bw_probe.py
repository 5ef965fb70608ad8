"""Probe: achievable HBM write / copy bandwidth for ~100 MB streams (the size of the hoisted K/V tensors at config 2)."""
import time, torch
dev = 'cuda'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for mb in (100, 400, 1600):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.empty(n, dtype=torch.bfloat16, device=dev)
    us = t(lambda: a.zero_()); print(f"{mb} MB fill : {us:7.1f} us  {mb * 1.048576 / us * 1e3 / 1e3:.2f} TB/s written")
    us = t(lambda: a.copy_(b)); print(f"{mb} MB copy : {us:7.1f} us  {2 * mb * 1.048576 / us * 1e3 / 1e3:.2f} TB/s moved")
    us = t(lambda: a.sum()); print(f"{mb} MB read : {us:7.1f} us  {mb * 1.048576 / us * 1e3 / 1e3:.2f} TB/s read")
# write-only streams of non-zero data (a GEMM epilogue's pattern), 537 MB = the hoisted K/V tensor of config 5
n = 537 * 1000 * 1000 // 2
a = torch.empty(n, dtype=torch.bfloat16, device=dev)
us = t(lambda: a.fill_(1.25)); print(f"537 MB fill(1.25): {us:7.1f} us  {537 / us:.2f} TB/s written")
