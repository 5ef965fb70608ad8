"""Probe: which physical (XCC, SE, CU) does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X?"""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboverlap_probe.so"))
P.op_stream_masked.restype = C.c_void_p
dev = "cuda"

def stream_of(bits):
    words = (C.c_uint32 * 8)(*[sum(((1 if (32 * w + b) in bits else 0) << b) for b in range(32)) for w in range(8)])
    p = P.op_stream_masked(words, 8)
    assert p
    return torch.cuda.ExternalStream(p)

def where(s, grid=4096):
    out = torch.zeros(grid, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        assert P.op_where(C.c_void_p(out.data_ptr()), grid, C.c_void_p(s.cuda_stream)) == 0
    torch.cuda.synchronize()
    v = out.cpu().tolist()
    c = collections.Counter()
    for x in v:
        xcc, hw = x & 15, (x >> 4) & 0xffffffff
        c[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)] += 1
    return c

full = where(torch.cuda.Stream())
print("full chip: distinct (xcc,se,sh,cu) =", len(full), " per xcc:", sorted(collections.Counter(k[0] for k in full).items()))
for i in (0, 1, 2, 7, 8, 9, 16, 31, 32, 33, 64, 100, 128, 200, 255):
    c = where(stream_of({i}), 512)
    print(f"bit {i:3d}: {sorted(c.items())}")
for name, bits in (("low32", set(range(32))), ("low64", set(range(64))), ("low128", set(range(128))), ("every2nd", set(range(0, 256, 2))),
                   ("every4th", set(range(0, 256, 4))), ("low96", set(range(96))), ("hi128", set(range(128, 256)))):
    c = where(stream_of(bits))
    print(f"{name:9s}: distinct CUs {len(c):3d}; per xcc: {sorted(collections.Counter(k[0] for k in c).items())}")
