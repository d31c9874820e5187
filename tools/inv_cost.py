#!/usr/bin/env python3
"""How many multiplications is one field inversion on this GPU?  (the batched-affine accounting rests on it.)
Times og_field_op_d 'inv' (Fermat: x^(p-2), one lane per element) against a dependent multiplication chain of the same launch
shape (og_field_mulchain_d), Fq, 2^20 elements.  -> gpurun_out/inv_cost.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api  # noqa: E402


def main():
    ctx = api.Context(0)
    n = 1 << 20
    g = torch.Generator().manual_seed(1)
    a = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    a[:, 31] &= 0x1F
    a = a.cuda()
    b = a.flip(0).contiguous()
    out = {}
    for field, name in ((api.FQ, "Fq"), (api.FR, "Fr")):
        ctx.field_op(field, "inv", a)
        torch.cuda.synchronize()
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.field_op(field, "inv", a)
            torch.cuda.synchronize()
            t.append(time.perf_counter() - t0)
        inv_ms = min(t) * 1e3
        iters = 2000
        x = a.clone()
        ms = ctx.field_mulchain(field, x, b, iters)          # kernel time of n lanes x iters dependent multiplications
        mul_ns = ms * 1e6 / iters / n
        out[name] = {"n": n, "inv_ms": round(inv_ms, 3), "ns_per_inversion_per_lane_slot": round(inv_ms * 1e6 / n, 3),
                     "mulchain_ms": round(ms, 3), "ns_per_multiplication_per_lane_slot": round(mul_ns, 4),
                     "multiplications_per_inversion": round(inv_ms * 1e6 / n / mul_ns, 1)}
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "inv_cost.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
