"""Timing of the PointNet++ operators at the object-encoder shapes of the reference ([B * N_obj, 1024, 3] clouds)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import pointnet2 as P
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e6
B, N = 400, 1024
xyz = torch.rand(B, N, 3, device='cuda') * 2 - 1
feats = torch.randn(B, 64, N, device='cuda')
print("fps 1024->512 x400 clouds   %.0f us" % t(lambda: P.furthest_point_sample(xyz, 512)))
idx = P.furthest_point_sample(xyz, 512)
new = P.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
print("ball query r0.2 ns32        %.0f us" % t(lambda: P.ball_query(0.2, 32, xyz, new)))
bq = P.ball_query(0.2, 32, xyz, new)
print("group [400,64,512,32]       %.0f us" % t(lambda: P.grouping_operation(feats, bq)))
print("three_nn 1024 <- 512        %.0f us" % t(lambda: P.three_nn(xyz, new)))
