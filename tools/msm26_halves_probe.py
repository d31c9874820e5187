#!/usr/bin/env python3
"""Feasibility probe for VERDICT r4 item 5's "retry the two-half overlap": the lone 2^26-point MSM as TWO window halves (even /
odd windows: the window-sharded entry points og_msm_windows_d / og_msm_combine_d, already tested) issued from two contexts of
ONE GPU -- i.e. on two sets of streams -- by two host threads, the second started `delay` ms after the first, so that its
digit sort runs beside the first half's bucket accumulation and the first half's tail beside the second's accumulation.
Against the single-call MSM on the same box.  No library change: if this does not win, neither would the in-library form.
Writes gpurun_out/msm26_halves.json."""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api, groth16  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    n = 1 << log_n
    a_ctx, b_ctx = api.Context(0), api.Context(0)
    g = torch.Generator(device="cuda").manual_seed(26)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    k[:, 31] &= 0x0F
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    s[:, 31] &= 0x0F
    s[::16] = 0
    s[1::16] = 0
    s[1::16, 0] = 1
    pts = a_ctx.scalar_mul(1, groth16.G1_GEN_BYTES, k)
    bases = api.Bases(a_ctx, 1, pts, 16, False)
    del pts, k
    torch.cuda.synchronize()

    def whole():
        t0 = time.perf_counter()
        out = bases.msm(s)
        return (time.perf_counter() - t0) * 1e3, out[0].tobytes()

    lib = a_ctx._lib
    import ctypes as C
    pb = bases.partial_bytes()
    parts = torch.empty(2 * pb, dtype=torch.uint8, device="cuda")

    def half(ctx, rank, delay_ms, t_start, done):
        while (time.perf_counter() - t_start) * 1e3 < delay_ms:
            pass
        ctx._check(lib.og_msm_windows_d(ctx._h, bases._h, C.c_void_p(s.data_ptr()), n, rank, 2, C.c_void_p(parts.data_ptr() + rank * pb)))
        done[rank] = (time.perf_counter() - t_start) * 1e3

    def halves(delay_ms):
        torch.cuda.synchronize()
        done = [0.0, 0.0]
        t0 = time.perf_counter()
        th = [threading.Thread(target=half, args=(a_ctx, 0, 0.0, t0, done)), threading.Thread(target=half, args=(b_ctx, 1, delay_ms, t0, done))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        out = bases.msm_combine(parts, 2)
        return (time.perf_counter() - t0) * 1e3, out.tobytes(), done

    res = {"log_n": log_n, "whole_ms": [], "halves": {}}
    ref = None
    for _ in range(3):
        ms, out = whole()
        res["whole_ms"].append(round(ms, 2))
        ref = out
    # the split's own cost: the two halves one after the other on one context
    t0 = time.perf_counter()
    for r in (0, 1):
        a_ctx._check(lib.og_msm_windows_d(a_ctx._h, bases._h, C.c_void_p(s.data_ptr()), n, r, 2, C.c_void_p(parts.data_ptr() + r * pb)))
    out = bases.msm_combine(parts, 2)
    res["halves_sequential_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    assert out.tobytes() == ref
    for delay in (0.0, 4.0, 7.0, 10.0, 14.0, 20.0):
        runs = []
        for _ in range(3):
            ms, out, done = halves(delay)
            assert out == ref, "two-half MSM differs from the single call"
            runs.append({"total_ms": round(ms, 2), "half_done_ms": [round(x, 2) for x in done]})
        res["halves"][str(delay)] = runs
        print(delay, runs, flush=True)
    print("whole", res["whole_ms"], "sequential halves", res["halves_sequential_ms"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "msm26_halves.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
