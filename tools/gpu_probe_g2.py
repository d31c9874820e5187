"""Quick GPU sanity probe (seconds): G1/G2 scalar multiplication and small G1/G2 MSMs against the python oracle,
with OG_DEBUG_SYNC step logging so a hang is attributable to one kernel."""
import os, sys, random, time
os.environ.setdefault("OG_DEBUG_SYNC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from owshen_amd import api, groth16
from oracle.py import fields
from oracle.py.curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes, g1_from_bytes, g2_from_bytes

def say(*a):
    print(f"[{time.time() - T0:6.1f}s]", *a, flush=True)

T0 = time.time()
ctx = api.Context(0)
say("ctx ok")
rnd = random.Random(1)
ks = [rnd.randrange(fields.R) for _ in range(5)] + [0, 1]
kd = ctx.to_device(api.ints_to_bytes(ks))
got = ctx.to_host(ctx.scalar_mul(1, groth16.G1_GEN_BYTES, kd))
assert all(g1_from_bytes(got[i].tobytes()) == (G1.mul(G1_GEN, k) if k else None) for i, k in enumerate(ks)); say("G1 scalar_mul ok")
got = ctx.to_host(ctx.scalar_mul(2, groth16.G2_GEN_BYTES, kd))
assert all(g2_from_bytes(got[i].tobytes()) == (G2.mul(G2_GEN, k) if k else None) for i, k in enumerate(ks)); say("G2 scalar_mul ok")
for group, G, GEN, tob, fromb in ((1, G1, G1_GEN, g1_to_bytes, g1_from_bytes), (2, G2, G2_GEN, g2_to_bytes, g2_from_bytes)):
    n = 19
    pk = [rnd.randrange(1, fields.R) for _ in range(n)]
    pts = [G.mul(GEN, k) for k in pk]
    sc = [rnd.randrange(fields.R) for _ in range(n)]
    pb = 64 if group == 1 else 128
    b = np.frombuffer(b"".join(tob(q) for q in pts), dtype=np.uint8).reshape(-1, pb).copy()
    want = G.msm_naive(sc, pts)
    for window, precomp in ((8, False), (8, True)):
        bases = api.Bases(ctx, group, ctx.to_device(b), window, precomp)
        say(f"G{group} bases c={window} precomp={precomp} ok")
        got = bases.msm(ctx.to_device(api.ints_to_bytes(sc)))
        assert fromb(got[0].tobytes()) == want
        say(f"G{group} msm c={window} precomp={precomp} ok")
say("probe ok")
