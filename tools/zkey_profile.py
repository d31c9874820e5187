"""The headline's key out as a .zkey and back (og_zkey_export / og_zkey_import), for `rocprofv3 --kernel-trace --stats`: where a
load-time import spends its time (profiles/r06g_zkey_kernel_stats.csv).  usage: python tools/zkey_profile.py [--natural]"""
import json
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from owshen_amd import api, circuit, groth16 as g16, zkey as zk   # noqa: E402

ctx = api.Context(0)
depth = 32
dense = "--natural" not in sys.argv
n_pad3, n_pad2 = circuit.baseline_shape(depth, dense=True) if dense else (0, 0)
r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=dense)
blob, vk = g16.setup(ctx, r1, 11, 12, 13, 14, 15)
vkb = g16.vk_to_bytes(vk)
out = {"n_wires": r1.n_wires, "domain": r1.domain_size}
for rep in range(2):
    t0 = time.perf_counter()
    data = zk.export_zkey(ctx, blob, vkb)
    t1 = time.perf_counter()
    pk2, vk2 = zk.import_zkey(ctx, data)
    t2 = time.perf_counter()
    out[f"run{rep}"] = {"export_s": round(t1 - t0, 3), "import_s": round(t2 - t1, 3)}
out["zkey_bytes"] = len(data)
out["identical"] = vk2 == vkb and pk2[-64 * (r1.domain_size - 1):] == blob[-64 * (r1.domain_size - 1):]
print(json.dumps(out))
ctx.close()
