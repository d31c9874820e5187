"""libowshen_verify.so: the verifier for hosts without ROCm user space.  The library must not need the HIP
runtime or RCCL to load, and must give the same verdicts as the full library's og_verify."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "owshen_amd", "libowshen_verify.so")


def test_needs_no_rocm_library():
    out = subprocess.run(["readelf", "-d", SO], capture_output=True, text=True, check=True).stdout
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", out)
    assert needed, out
    assert not [n for n in needed if re.search(r"hip|rccl|hsa|roc|amd", n, flags=re.I)], needed
    syms = subprocess.run(["nm", "-D", "--undefined-only", SO], capture_output=True, text=True, check=True).stdout
    assert not re.search(r"\bhip[A-Z]|\bnccl|\bhsa_", syms), syms


def test_loads_and_verifies_in_a_process_without_torch_or_hip():
    """a fresh interpreter that imports ONLY ctypes + the wrapper: accept the golden proof, reject a wrong input, a tampered
    proof, an input >= r whose IC base is irrelevant, and report a malformed key as an error"""
    code = r'''
import json, sys
sys.path.insert(0, %r)
import importlib.util, os
spec = importlib.util.spec_from_file_location("verify_only", os.path.join(%r, "owshen_amd", "verify_only.py"))
vo = importlib.util.module_from_spec(spec); spec.loader.exec_module(vo)
assert "torch" not in sys.modules and "owshen_amd" not in sys.modules
case = json.load(sys.stdin)
vk, pub, proof = bytes.fromhex(case["vk"]), [int(x) for x in case["pub"]], bytes.fromhex(case["proof"])
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
assert vo.verify(vk, pub, proof)
assert not vo.verify(vk, [pub[0] + 1] + pub[1:], proof)
bad = bytearray(proof); bad[70] ^= 1
assert not vo.verify(vk, pub, bytes(bad))
assert not vo.verify(vk, [pub[0] + R] + pub[1:], proof) if pub[0] + R < 2 ** 256 else True
try:
    vo.verify(vk[:-1], pub, proof)
    raise SystemExit("malformed key accepted")
except ValueError:
    pass
for n in open("/proc/self/maps").read().splitlines():
    assert not any(t in n for t in ("libamdhip64", "librccl", "libhsa-runtime")), n
print("ok")
''' % (ROOT, ROOT)
    import json
    sys.path.insert(0, ROOT)
    from oracle.py import groth16 as og16
    from oracle.py.curve import g1_to_bytes, g2_to_bytes
    from tests.golden_cases import groth16_instance, GOLD
    n_wires, cons, z, toxic, r, s = groth16_instance()
    n_pub = GOLD["groth16"]["n_pub"]
    ro = og16.R1CS(n_wires, n_pub, cons)
    _pk, vk = og16.setup(ro, *toxic)
    blob = (b"OWVK0001" + n_pub.to_bytes(8, "little") + g1_to_bytes(vk["alpha_g1"]) + g2_to_bytes(vk["beta_g2"]) +
            g2_to_bytes(vk["gamma_g2"]) + g2_to_bytes(vk["delta_g2"]) + b"".join(g1_to_bytes(p) for p in vk["ic"]))
    case = {"vk": blob.hex(), "pub": [str(v) for v in z[1:n_pub + 1]], "proof": GOLD["groth16"]["proof"]}
    out = subprocess.run([sys.executable, "-c", code], input=json.dumps(case), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
