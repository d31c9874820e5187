"""One leaf appended to the depth-32 commitment tree (og_mimc7_append_d; what `mint_tx` does per deposit): the level kernels (32
launches of a lone lane pair) against the host-chains mode (og_set_host_chains: the 32 hashes on a host core).  Also 8 and 64
leaves per call.  Writes gpurun_out/append_latency.json."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api  # noqa: E402


def main():
    ctx = api.Context(0)
    depth = 32
    rng = np.random.default_rng(3)
    out = {"what": __doc__, "depth": depth, "rows": {}}
    for k in (1, 8, 64):
        leaves = rng.integers(0, 256, (k, 32), dtype=np.uint8)
        leaves[:, 31] &= 0x1F
        lv = ctx.to_device(leaves)
        row = {}
        roots = {}
        for mode, bound in (("kernels", 0), ("host_chains", 64)):
            ctx.set_host_chains(bound)
            f = ctx.to_device(np.zeros((depth, 32), dtype=np.uint8))
            n = 12345          # (some frontier state: 12345 leaves appended before, as zeros -- only the timing matters here)
            ts = []
            for _ in range(12):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                f2, root = ctx.mimc7_append(depth, f, n, lv)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = sorted(ts[2:])
            row[mode] = {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3)}
            roots[mode] = bytes(ctx.to_host(root))
        row["same_root"] = roots["kernels"] == roots["host_chains"]
        out["rows"][f"leaves_{k}"] = row
        print(k, row)
    ctx.set_host_chains(0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "append_latency.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
