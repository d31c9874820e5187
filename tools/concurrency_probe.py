"""Do G1 and G2 bucket accumulation share the chip better than they run back to back?  Two og_ctx on one device, one host
thread each: thread A runs R G1 MSMs (n = 2^18, batch 64: the shape of the L query), thread B one G2 MSM (n = 2^17, batch 64).
Prints sequential and concurrent wall times.  -> gpurun_out/concurrency_probe.json"""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api, groth16 as g16  # noqa: E402


def rand_scalars(*shape, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (*shape, 32), dtype=torch.uint8, generator=g)
    x[..., 31] &= 0x1F
    return x.cuda()


def main():
    a, b = api.Context(0), api.Context(0)
    n1, n2, batch, reps = 1 << 18, 1 << 17, 64, int(os.environ.get("REPS", "6"))
    p1 = a.scalar_mul(1, g16.G1_GEN_BYTES, rand_scalars(n1, seed=2))
    p2 = b.scalar_mul(2, g16.G2_GEN_BYTES, rand_scalars(n2, seed=3))
    b1 = api.Bases(a, 1, p1, 16, True)
    b2 = api.Bases(b, 2, p2, 16, True)
    s1, s2 = rand_scalars(batch, n1, seed=4), rand_scalars(batch, n2, seed=5)

    def run_a():
        for _ in range(reps):
            b1.msm(s1)

    def run_b():
        b2.msm(s2)

    run_a(); run_b()  # warm
    torch.cuda.synchronize()
    res = {}
    for rnd in range(2):
        t = time.time(); run_a(); ta = time.time() - t
        t = time.time(); run_b(); tb = time.time() - t
        th = [threading.Thread(target=run_a), threading.Thread(target=run_b)]
        t = time.time()
        for x in th:
            x.start()
        for x in th:
            x.join()
        tc = time.time() - t
        res[f"round{rnd}"] = {"g1_x%d_ms" % reps: ta * 1e3, "g2_ms": tb * 1e3, "sequential_ms": (ta + tb) * 1e3, "concurrent_ms": tc * 1e3}
        print(res[f"round{rnd}"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "concurrency_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
