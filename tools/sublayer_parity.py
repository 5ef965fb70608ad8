#!/usr/bin/env python3
"""Per-sublayer parity report at config-2 shapes (tests/sublayer.py): HIP fp32 / HIP bf16 against the float64 oracle,
plus the oracle with bf16 operand rounding (what ideal bf16-operand arithmetic gives) for comparison.  Run on the GPU
box; the output is committed under profiles/."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import sublayer as S  # noqa: E402


def fmt(m):
    return f"{m[0]:9.2e} {m[1]:9.2e} {1 - m[2]:9.2e}"


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    names = sys.argv[1:] or list(S.CASES)
    have_gpu = torch.cuda.is_available()
    print(f"{'sublayer':18s} {'path':22s} {'quantity':46s} {'max/scale':>9s} {'relL2':>9s} {'1-cos':>9s}")
    for n in names:
        case = S.CASES[n]()
        ref = case.run_oracle()
        emu = case.run_oracle(emulate=torch.bfloat16)
        rows = [("oracle bf16-operands", emu)]
        if have_gpu:
            rows = [("HIP fp32", case.run_hip("fp32")), ("HIP bf16", case.run_hip("bf16"))] + rows
        for path, (o, g) in rows:
            for i, (a, b) in enumerate(zip(o, ref[0])):
                print(f"{n:18s} {path:22s} {'out' + str(i):46s} {fmt(S.metrics(a, b))}")
            worst = None
            # gradients that are analytically zero (the key bias of a softmax attention: shifting every key by a constant
            # changes no probability) have no scale to be relative to: listed as such, kept out of the WORST row -- the rule of
            # tests/test_gpu_sublayer_parity.py
            gmax = max(float(v.double().norm()) for v in ref[1].values())
            for k, v in ref[1].items():
                if k in g and g[k] is not None:
                    if float(v.double().norm()) < 1e-6 * gmax:
                        print(f"{n:18s} {path:22s} {'d ' + k:46s} analytically zero (|ref| {float(v.double().norm()):.1e}, "
                              f"|got| {float(g[k].double().norm()):.1e}; largest gradient {gmax:.1e})")
                        continue
                    m = S.metrics(g[k], v)
                    print(f"{n:18s} {path:22s} {'d ' + k:46s} {fmt(m)}")
                    worst = m if worst is None or m[1] > worst[1] else worst
            if worst:
                print(f"{n:18s} {path:22s} {'WORST gradient (relL2)':46s} {fmt(worst)}")


if __name__ == "__main__":
    main()
