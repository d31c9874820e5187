"""What the boundary costs when the caller's buffers live in HOST memory: the same 2^18-wire dense circuit,
  a) input records resident in HBM               og_withdraw_prove_batch_d        (bench.py's `value`)
  b) input records in host memory                og_multi_withdraw_prove_batch    (1.3 KB per proof over PCIe)
  c) whole witnesses in host memory              og_prove_batch                   (8.4 MB per proof over PCIe, pageable)
Writes gpurun_out/host_boundary.json.  One context at a time: a context's sub-batch scratch is ~165 GB."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api, circuit, groth16, multi  # noqa: E402

DEPTH = 32


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    B, Bw = int(os.environ.get("HB_BATCH", "1024")), int(os.environ.get("HB_WITNESS_BATCH", "256"))
    ctx = api.Context(0)
    n_pad3, n_pad2 = circuit.baseline_shape(DEPTH, dense=True)
    r1 = circuit.withdraw_r1cs_native(ctx, DEPTH, n_pad3, n_pad2, dense=True)
    blob, _ = groth16.setup(ctx, r1, 11, 12, 13, 14, 15)
    rng = np.random.default_rng(3)
    inputs = rng.integers(0, 256, (B, 8 + DEPTH, 32), dtype=np.uint8)
    inputs[:, :, 31] &= 0x1F
    inputs[:, 5, 8:] = 0
    inputs[:, 6, 20:] = 0
    inputs[:, 7, 8:] = 0
    inputs[:, 5, :8] = (inputs[:, 5, :8].copy().view(np.uint64) & np.uint64((1 << DEPTH) - 1)).view(np.uint8)
    rs = rng.integers(0, 256, (B, 64), dtype=np.uint8)
    rs[:, 31] &= 0x1F
    rs[:, 63] &= 0x1F
    out = {"circuit": "depth-32 withdraw, 2^18 wires, dense padding", "batch": B}

    pk = groth16.ProvingKey(ctx, blob)
    inputs_d = ctx.to_device(inputs)
    ref = []
    dt = timed(lambda: ref.append(circuit.prove_from_inputs(ctx, pk, DEPTH, inputs_d, rs, n_pad3, n_pad2)))
    out["a_records_in_hbm"] = {"entry": "og_withdraw_prove_batch_d", "proofs_per_s": round(B / dt, 1), "ms": round(dt * 1e3, 1)}
    # c) whole witnesses from host memory
    wit = ctx.to_host(circuit.witness(ctx, DEPTH, ctx.to_device(inputs[:Bw]), n_pad3, n_pad2))
    wit = np.ascontiguousarray(wit)
    rs_w = np.ascontiguousarray(rs[:Bw])
    got = []
    dt_c = timed(lambda: got.append(pk.prove_batch(wit, rs_w)))
    assert got[-1].tobytes() == ref[-1][:Bw].tobytes(), "host-witness proofs differ from the fused call's"
    dev = ctx.to_device(wit)
    dt_cd = timed(lambda: pk.prove_batch_device(dev, rs_w))
    out["c_witnesses_in_host_memory"] = {"entry": "og_prove_batch", "batch": Bw, "proofs_per_s": round(Bw / dt_c, 1), "ms": round(dt_c * 1e3, 1),
                                         "bytes_over_pcie_per_proof": int(wit.shape[1]) * 32,
                                         "same_witnesses_resident": {"entry": "og_prove_batch_d", "proofs_per_s": round(Bw / dt_cd, 1)}}
    del dev, wit, inputs_d
    pk.close()
    ctx.release_scratch()
    ctx.close()
    torch.cuda.empty_cache()
    # b) records from host memory through the one-process multi-device entry point (here: one device)
    m = multi.Multi(1)
    pks = m.load_key(blob)
    got = []
    dt_b = timed(lambda: got.append(m.withdraw_prove_batch(pks, DEPTH, inputs, rs, n_pad3, n_pad2)))
    assert got[-1].tobytes() == ref[-1].tobytes(), "host-record proofs differ from the device-record call's"
    out["b_records_in_host_memory"] = {"entry": "og_multi_withdraw_prove_batch (1 device)", "proofs_per_s": round(B / dt_b, 1),
                                       "ms": round(dt_b * 1e3, 1), "bytes_over_pcie_per_proof": (8 + DEPTH) * 32 + 64}
    m.free_key(pks)
    m.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "host_boundary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
