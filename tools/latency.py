"""Single-request latency of the withdraw path (the `withdraw_handler` case): input record -> 256-byte proof, batch 1 / 8 / 64,
benchmark circuit (2^18 wires) or, with --natural, the depth-32 statement without padding gates (26 385 wires).
Writes gpurun_out/latency.json (latency_natural.json)."""
import json
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
    host_chains = "--host-chains" in sys.argv  # og_set_host_chains(64): the requests' MiMC7 chains on the host CPU
    if host_chains:
        ctx.set_host_chains(64)
    natural = "--natural" in sys.argv      # the depth-32 statement alone (26 385 wires): what withdraw_handler would prove
    n_pad3, n_pad2 = (0, 0) if natural else circuit.baseline_shape(depth)
    r1 = circuit.withdraw_r1cs_native(ctx, depth, n_pad3, n_pad2)
    blob, _ = groth16.setup(ctx, r1, 11, 12, 13, 14, 15)
    pk = groth16.ProvingKey(ctx, blob)
    rng = np.random.default_rng(1)
    out = {}
    sizes = tuple(int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(",")) if "--sizes" in sys.argv else (1, 8, 64)
    for b in sizes:
        inputs = rng.integers(0, 256, (b, 8 + depth, 32), dtype=np.uint8)
        inputs[:, :, 31] &= 0x1F
        inputs[:, 5, 8:] = 0
        inputs[:, 5, :8] = (inputs[:, 5, :8].copy().view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
        rs = rng.integers(0, 256, (b, 64), dtype=np.uint8)
        rs[:, 31] &= 0x1F
        rs[:, 63] &= 0x1F
        d = ctx.to_device(inputs)
        ts = []
        for _ in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            circuit.prove_from_inputs(ctx, pk, depth, d, rs, n_pad3, n_pad2)
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        out[f"batch_{b}"] = {"median_ms": round(ts[len(ts) // 2] * 1e3, 2), "min_ms": round(ts[0] * 1e3, 2),
                             "proofs_per_s": round(b / ts[len(ts) // 2], 1)}
    # where one request's time goes: HIP-event regions of one more single-request call (the regions overlap across the two
    # streams of a one-proof call, so they do not add up to the wall time)
    for b in sizes:
        db = ctx.to_device(inputs[:b])
        ctx.profile(True)
        circuit.prove_from_inputs(ctx, pk, depth, db, rs[:b], n_pad3, n_pad2)
        out[f"batch_{b}_regions_ms"] = {k: round(v[0], 3) for k, v in ctx.profile_read().items() if v[1]}
        ctx.profile(False)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out["circuit"] = "natural depth-32 statement, 26385 wires" if natural else "benchmark shape, 2^18 wires"
    out["host_chains"] = host_chains
    name = ("latency_natural" if natural else "latency") + ("_host_chains" if host_chains else "") + ".json"
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
