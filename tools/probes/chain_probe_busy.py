import torch, sys
from pq3d_amd import _lib as L
mode = sys.argv[1]
x = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
if mode == "busy_side":
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(50): y = x @ x
    print("busy side:", L.lib().pq3d_chain_device_ok(1, L.stream()))
elif mode == "busy_small":
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2000): y = x[:64] + 1
    print("busy small:", L.lib().pq3d_chain_device_ok(1, L.stream()))
else:
    print("idle:", L.lib().pq3d_chain_device_ok(1, L.stream()))
torch.cuda.synchronize()
