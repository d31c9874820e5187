#!/usr/bin/env python3
"""Real kernels beside the real persistent accumulation: thread A (its own og_ctx) runs G1 MSMs of the L query's shape
(n = 2^18, batch = BATCH proofs -> one ~60 ms k_accumulate_p launch each) back to back; thread B (another og_ctx, i.e. other
HIP streams) times short library calls -- batched MiMC7 hashes (VALU), NTTs (256-lane workgroups, 36 KB LDS), field
multiplications (streaming) -- alone and beside A.  -> gpurun_out/coresidency_real.json"""
import json
import os
import statistics
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api, groth16 as g16  # noqa: E402


class Ctx(api.Context):
    def _pre(self):   # no device-wide synchronize: the two threads must not wait for each other's streams
        pass


def rand_fr(*shape, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (*shape, 32), dtype=torch.uint8, generator=g)
    x[..., 31] &= 0x1F
    return x.cuda()


def main():
    a, b = Ctx(0), Ctx(0)
    n1, batch = 1 << 18, int(os.environ.get("BATCH", "240"))
    p1 = a.scalar_mul(1, g16.G1_GEN_BYTES, rand_fr(n1, seed=2))
    torch.cuda.synchronize()
    b1 = api.Bases(a, 1, p1, 16, True)
    s1 = rand_fr(batch, n1, seed=4)
    h_l, h_r = rand_fr(1 << 16, seed=5), rand_fr(1 << 16, seed=6)
    ntt_in = rand_fr(8, 1 << 17, seed=7)
    f_a, f_b = rand_fr(1 << 20, seed=8), rand_fr(1 << 20, seed=9)
    torch.cuda.synchronize()
    fillers = {
        "mimc7_hash2 x 2^16 (VALU)": lambda: b.mimc7_hash2(h_l, h_r),
        "ntt 2^17 x 8 (256-lane workgroups, 36 KB LDS)": lambda: b.ntt(ntt_in),
        "field mul x 2^20 (streaming)": lambda: b.field_op(api.FR, "mul", f_a, f_b),
    }
    for f in fillers.values():
        f()
    b1.msm(s1)
    torch.cuda.synchronize()
    stop = threading.Event()
    count = [0]

    def run_a():
        while not stop.is_set():
            b1.msm(s1)
            count[0] += 1

    def measure(reps):
        out = {}
        for name, f in fillers.items():
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                f()            # every library call synchronises its own stream before returning
                ts.append((time.perf_counter() - t) * 1e3)
            out[name] = {"median_ms": round(statistics.median(ts), 3), "min_ms": round(min(ts), 3), "max_ms": round(max(ts), 3)}
        return out

    res = {"config": {"batch": batch, "OG_ACC_WAVES_G1": os.environ.get("OG_ACC_WAVES_G1", "default")}, "alone": measure(20)}
    t0 = time.perf_counter(); b1.msm(s1); res["msm_alone_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    th = threading.Thread(target=run_a)
    th.start()
    time.sleep(0.3)
    t0, c0 = time.perf_counter(), count[0]
    res["beside"] = measure(20)
    dt, dc = time.perf_counter() - t0, count[0] - c0
    stop.set()
    th.join()
    res["msm_beside_ms"] = round(dt / max(dc, 1) * 1e3, 2)
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = os.environ.get("TAG", "")
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"coresidency_real{tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
