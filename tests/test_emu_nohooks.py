"""The SHIPPED configuration on the CPU interpreter: tests/hipemu's default build carries -DOG_AB_HOOKS, so the
configuration that ships -- every OG_HOOK_* a compile-time constant, injected_failure stubbed, the rejected kernels and fe_*_lat
compiled out -- used to run only on the GPU box.  Here the same sources are compiled WITHOUT the macro (make nohooks) and a smoke
subset runs on them in a fresh interpreter (tests/emu.py binds its library once per process: OG_EMU_LIB selects the other build):
withdraw proofs end to end (witness walk, sym / split schedules, the merged L + H pair on the serial path, og_verify),
a submitted call with og_job_poll / og_job_wait, and a window-sharded call through og_multi with one pretend device."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu")
LIB = os.path.join(EMU, "_build_nohooks", "libowshen_emu_nohooks.so")

SMOKE = r"""
import os, random
import numpy as np
from tests import emu, withdraw_cases as cases
assert emu._load is not None and os.environ["OG_EMU_LIB"].endswith("libowshen_emu_nohooks.so")
os.environ["OG_PIPE_MIN"] = "1"          # a hook the shipped configuration must NOT hear: the call below stays off the stage pipeline
ctx = emu.Ctx()
cases.case_withdraw_end_to_end(ctx, 1, 2, 3)                  # prove + verify, sym / split schedules, merged L + H
# (the dense rows have their cases on the hooks build; here every proof of a handful walks the wave-wide forms the shipped
# configuration defaults to -- the witness walk, the assembly's four products -- at ~25 s per case on the interpreter)
ctx.set_lanes(1)
cases.case_withdraw_end_to_end(ctx, 1, 2, 3)                  # serial path: MSM_FIRST / MSM_SECOND on one stream
ctx.set_lanes(2)
from owshen_amd import circuit, groth16 as g16
from oracle.py import fields
r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), 1, 2, 3)
blob, vk = g16.setup(ctx, r1, 21, 22, 23, 24, 25)
pk = g16.ProvingKey(ctx, blob)
assert pk.plan(5)[0] != "stage pipeline", pk.plan(5)        # OG_PIPE_MIN is not read by this build
rnd = random.Random(2)
recs = np.stack([circuit.pack_inputs(rnd.randrange(fields.R), rnd.randrange(fields.R), 5, 6, rnd.randrange(fields.R), rnd.randrange(2),
                                     [rnd.randrange(fields.R)], token=rnd.randrange(1 << 160), chain_id=1387) for _ in range(3)])
rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in range(3)]
want = circuit.prove_from_inputs(ctx, pk, 1, ctx.to_device(recs), rs, 2, 3)
job = circuit.submit_from_inputs(ctx, pk, 1, ctx.to_device(recs), rs, 2, 3)
assert job.done() is True
assert job.wait().tobytes() == want.tobytes()
pk.close()
# (og_multi_* is not part of this smoke: with more than one device the shipped build runs one host thread per device, and the
# fiber interpreter is single-threaded -- the hooks build's OG_MULTI_SEQUENTIAL exists for exactly that.  One device has no
# workers: the window-sharded call through og_multi with a single pretend device still covers the front / back split.)
from owshen_amd import multi
m = multi.Multi(1, lib=emu.lib)
pks = m.load_key(blob)
rsb = pk._rs_bytes(rs)
assert m.withdraw_prove_sharded(pks, 1, recs, rsb, 2, 3).tobytes() == want.tobytes()
os.environ["OG_MULTI_FAIL"] = "withdraw:0"   # failure injection does not exist in this build: the call must simply work
assert m.withdraw_prove_batch(pks, 1, recs, rsb, 2, 3).tobytes() == want.tobytes()
m.free_key(pks); m.close(); ctx.close()
print("NOHOOKS-SMOKE-OK")
"""


def test_the_shipped_configuration_runs_on_the_interpreter():
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU, "nohooks"])
    import re
    strs = subprocess.run(["strings", LIB], capture_output=True, text=True).stdout
    own = {"OG_SUB_BATCH", "OG_DEBUG_SYNC", "OG_EMU_DEVICES", "OG_EMU_PROF"}   # (the last two: the interpreter runtime's own)
    hooks = sorted(set(re.findall(r"^OG_[A-Z0-9_]+$", strs, flags=re.M)) - own)
    assert not hooks, f"the no-hooks interpreter build still names {hooks}"
    env = dict(os.environ, OG_EMU_LIB=LIB, PYTHONPATH=ROOT)
    for k in ("OG_PIPE_MIN", "OG_MULTI_FAIL", "OG_EMU_DEVICES"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", SMOKE], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0 and "NOHOOKS-SMOKE-OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
