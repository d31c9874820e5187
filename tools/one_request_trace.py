"""One request in the host-chains mode under `rocprofv3 --kernel-trace`: proves the same record a few times; the timeline of the
LAST call's kernels is printed by tools/trace_timeline.py --min 0 --from <t>.  usage: one_request_trace.py [--natural] [--gpu-only]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api, circuit, groth16  # noqa: E402


def main():
    ctx = api.Context(0)
    depth = 32
    n_pad3, n_pad2 = (0, 0) if "--natural" in sys.argv else circuit.baseline_shape(depth)
    r1 = circuit.withdraw_r1cs_native(ctx, depth, n_pad3, n_pad2)
    blob, _ = groth16.setup(ctx, r1, 11, 12, 13, 14, 15)
    pk = groth16.ProvingKey(ctx, blob)
    rng = np.random.default_rng(1)
    inputs = rng.integers(0, 256, (1, 8 + depth, 32), dtype=np.uint8)
    inputs[:, :, 31] &= 0x1F
    inputs[:, 5, 8:] = 0
    inputs[:, 5, :8] = (inputs[:, 5, :8].copy().view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
    rs = rng.integers(0, 256, (1, 64), dtype=np.uint8)
    rs[:, 31] &= 0x1F
    rs[:, 63] &= 0x1F
    d = ctx.to_device(inputs)
    if "--gpu-only" not in sys.argv:
        ctx.set_host_chains(16)
    for k in range(6):
        torch.cuda.synchronize()
        time.sleep(0.05)   # a visible gap in the trace in front of every call
        t0 = time.perf_counter()
        circuit.prove_from_inputs(ctx, pk, depth, d, rs, n_pad3, n_pad2)
        print("call", k, round((time.perf_counter() - t0) * 1e3, 3), "ms")


if __name__ == "__main__":
    main()
