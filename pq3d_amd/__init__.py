"""pq3d_amd: MI355X-native (gfx950) implementation of PQ3D's promptable query decoder hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); all
arithmetic runs in hand-written HIP kernels behind a C-ABI shared library (include/pq3d_hip.h).
"""
__version__ = "0.1.0"
