"""Does a memory-bound G2 accumulation (k_accumulate_affine, OG_G2_AFFINE=1) hide under the VALU-bound G1 accumulations?
Two contexts on one GPU, one host thread each: a G1 MSM over 2^18 points x 192 scalar vectors (the A, B1, L queries of 64
proofs) beside a G2 MSM over 2^18 points x 64 vectors (their B2 query), window tables, 17-bit windows -- alone, and together.
Writes gpurun_out/corun_probe.json."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OWSHEN_GPU_LIB", os.path.join(ROOT, "owshen_amd", "libowshen_gpu_hooks.so"))  # OG_G2_AFFINE is a hooks-build switch
from owshen_amd import api, groth16  # noqa: E402


def rand_fr(rng, *shape):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= 0x1F
    return a


def main():
    n = 1 << int(os.environ.get("PROBE_LOG_N", "18"))
    nb = int(os.environ.get("PROBE_BATCH", "64"))
    rng = np.random.default_rng(3)
    c1, c2 = api.Context(0), api.Context(0)
    ks = c1.to_device(rand_fr(rng, n))
    b1 = api.Bases(c1, 1, c1.scalar_mul(1, groth16.G1_GEN_BYTES, ks), 17, True)
    b2 = api.Bases(c2, 2, c2.scalar_mul(2, groth16.G2_GEN_BYTES, c2.to_device(rand_fr(rng, n))), 17, True)
    s1 = c1.to_device(rand_fr(rng, 3 * nb, n))
    s2 = c2.to_device(rand_fr(rng, nb, n))
    out = {"n": n, "g1_vectors": 3 * nb, "g2_vectors": nb}

    def g1():
        return b1.msm(s1)

    def g2():
        return b2.msm(s2)

    def timed(*fns):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            ths = [threading.Thread(target=f) for f in fns]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return round(best * 1e3, 1)

    ref2 = None
    for waves in ("12", "8"):
        os.environ["OG_ACC_WAVES_G1"] = waves
        for aff in ("0", "1"):
            os.environ["OG_G2_AFFINE"] = aff
            r = g2()
            if ref2 is None:
                ref2 = r.copy()
            assert r.tobytes() == ref2.tobytes()
            key = f"g1_waves_{waves}_g2_{'affine' if aff == '1' else 'xyzz'}"
            out[key] = {"g1_alone_ms": timed(g1), "g2_alone_ms": timed(g2), "together_ms": timed(g1, g2)}
            print(key, out[key], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "corun_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
