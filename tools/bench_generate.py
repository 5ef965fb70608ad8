"""Greedy caption generation (generation_head.py:28) at the t5-small architecture: the HIP KV-cache decoder replayed
from a HIP graph vs the stock HF generate on PyTorch-ROCm.  Random-init weights, synthetic query tokens; EOS disabled so
both decode the full budget.  Usage: python tools/bench_generate.py [--batch 16] [--nq 100] [--new-tokens 50]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pq3d_amd import synth, t5  # noqa: E402
from pq3d_amd.modules import T5  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--nq", type=int, default=100)
    ap.add_argument("--new-tokens", type=int, default=50)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda"
    out = {"workload": f"t5-small greedy decode B{a.batch} Nq{a.nq} {a.new_tokens} new tokens", "data": "synthetic"}
    q = torch.randn(a.batch, a.nq, 256, generator=torch.Generator().manual_seed(0)).to(dev)
    mask = torch.ones(a.batch, a.nq, dtype=torch.bool, device=dev)
    for compute in ("fp32", "bf16"):
        head = T5(None, variant="t5-small", input_size=256, use_projection=True, body="hip", max_new_tokens=a.new_tokens)
        head.compute = compute
        synth.fill_module(head, 0)
        head.to(dev).eval()
        head.model.generation_config.eos_token_id = None     # decode the whole budget on both sides
        for use_graph in (True, False):
            with torch.no_grad():
                enc = torch.nn.Sequential(*head.input_proj)(q)
                dec = t5.GreedyDecoder(head.model, a.batch, a.nq, head.ct, a.new_tokens, dev, use_graph=use_graph)
                toks = dec(enc, mask)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    toks = dec(enc, mask)
                torch.cuda.synchronize()
            out[f"hip_{compute}_{'graph' if use_graph else 'eager'}_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 2)
        if compute == "fp32":
            from transformers.modeling_outputs import BaseModelOutput
            with torch.no_grad():
                gen = lambda: head.model.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc), attention_mask=mask,  # noqa: E731
                                                  do_sample=False, max_new_tokens=a.new_tokens)[:, 1:]
                ref = gen()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    ref = gen()
                torch.cuda.synchronize()
            out["hf_generate_fp32_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 2)
            out["fp32_tokens_equal_hf"] = bool(torch.equal(ref, toks)) if ref.shape == toks.shape else False
            out["fp32_token_agreement"] = float((ref == toks).float().mean()) if ref.shape == toks.shape else None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
