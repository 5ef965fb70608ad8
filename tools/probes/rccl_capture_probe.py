"""Does capturing an RCCL collective inside a HIP graph work on this stack (torch 2.10 + ROCm 7 RCCL)?  One variant per
subprocess (a crash in hipStreamEndCapture is a segfault, not an exception).  Single rank: a one-GPU box cannot host two.
usage: python tools/probes/rccl_capture_probe.py            (runs every variant)
       python tools/probes/rccl_capture_probe.py <variant>  (one variant, in-process)"""
import os, subprocess, sys

VARIANTS = ["sync_main_sum", "sync_main_avg", "async_side_sum", "async_side_avg", "async_side_avg_global", "async_side_avg_relaxed",
            "async_side_avg_nowait", "sync_main_avg_kernels_around"]


def one(v):
    import socket
    import torch, torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(1 << 20, device=dev)
    dist.all_reduce(t); torch.cuda.synchronize()           # communicator created eagerly
    op = dist.ReduceOp.AVG if "avg" in v else dist.ReduceOp.SUM
    mode = "global" if "global" in v else ("relaxed" if "relaxed" in v else "thread_local")
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        if "kernels_around" in v:
            t.mul_(2.0)
        if v.startswith("sync_main"):
            dist.all_reduce(t, op=op)
        else:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                h = dist.all_reduce(t, op=op, async_op=True)
            t2 = t * 1.0 if False else None
            with torch.cuda.stream(side):
                if "nowait" not in v:
                    h.wait()
            cur.wait_stream(side)
        if "kernels_around" in v:
            t.add_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"VARIANT {v}: ok, t[0] = {float(t[0])}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        for v in VARIANTS:
            p = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), v], capture_output=True, text=True, timeout=300)
            ok = [l for l in p.stdout.splitlines() if l.startswith("VARIANT")]
            print(ok[0] if ok else f"VARIANT {v}: rc = {p.returncode}; " + " | ".join(l for l in p.stderr.splitlines() if "File" in l or "Error" in l or "error" in l)[-600:], flush=True)
