"""The verifying key, a proof and its public inputs in the JSON files `snarkjs groth16 verify` reads.

Why: nothing in /root/reference pins a Groth16 proof ("parity unpinned"), and this image has no network
to fetch a third-party verifier.  A maintainer who has one can close that loop in one command -- the three files written
here are `verification_key.json`, `public.json` and `proof.json` of the snarkjs / circom tool chain (the one the Solidity side
of SURVEY.md 8f-2 follows), so

    snarkjs groth16 verify verification_key.json public.json proof.json

checks a proof of this library with code none of this repository's authors wrote.  In the repository the same three files go
through oracle/js/bn254_pairing_second.js --snarkjs (tests/test_snarkjs_json.py): format and acceptance are tested here, the
external run is the maintainer's.

Conventions (snarkjs `groth16_verify.js`, `zkey_export_verificationkey.js`; restated, the package is not in the image):
  * every coordinate is a DECIMAL string of the canonical value; points are projective with z = "1" (G2: ["1", "0"]);
  * an Fq2 element is [c0, c1] -- the REAL part first.  Only `soliditycalldata` swaps the halves (owshen_amd/evm.py);
  * `vk_alphabeta_12` (a cached pairing value) is not read by `groth16 verify` and is not written.
The library's own formats are 32-byte little-endian with G2 as x.c0 | x.c1 | y.c0 | y.c1 (include/owshen_gpu.h; the
reference's `Fp::to_repr()`, /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11).  Pure byte shuffling:
no arithmetic, no import of the GPU library.
"""
import json
import os
import struct

VK_MAGIC = b"OWVK0001"


def _le(b):
    return str(int.from_bytes(bytes(b), "little"))


def g1_json(p64):
    p64 = bytes(p64)
    assert len(p64) == 64
    if p64 == bytes(64):                      # the library writes the point at infinity as 64 zero bytes
        return ["0", "1", "0"]
    return [_le(p64[0:32]), _le(p64[32:64]), "1"]


def g2_json(p128):
    p128 = bytes(p128)
    assert len(p128) == 128
    if p128 == bytes(128):
        return [["0", "0"], ["1", "0"], ["0", "0"]]
    c = [_le(p128[32 * i:32 * i + 32]) for i in range(4)]
    return [[c[0], c[1]], [c[2], c[3]], ["1", "0"]]


def verification_key(vk_blob):
    """"OWVK0001" blob (og_setup / og_vk_export; include/owshen_gpu.h) -> the dict of verification_key.json"""
    vk_blob = bytes(vk_blob)
    if vk_blob[:8] != VK_MAGIC:
        raise ValueError("not an OWVK0001 verifying key")
    n_pub = struct.unpack("<Q", vk_blob[8:16])[0]
    o = 16
    if len(vk_blob) != o + 64 + 3 * 128 + (n_pub + 1) * 64:
        raise ValueError("verifying key blob has the wrong length for its n_pub")
    g2 = [vk_blob[o + 64 + 128 * k:o + 64 + 128 * (k + 1)] for k in range(3)]
    ic = o + 64 + 384
    return {"protocol": "groth16", "curve": "bn128", "nPublic": int(n_pub),
            "vk_alpha_1": g1_json(vk_blob[o:o + 64]),
            "vk_beta_2": g2_json(g2[0]), "vk_gamma_2": g2_json(g2[1]), "vk_delta_2": g2_json(g2[2]),
            "IC": [g1_json(vk_blob[ic + 64 * i:ic + 64 * (i + 1)]) for i in range(n_pub + 1)]}


def proof(proof256):
    """256-byte proof (A 64 | B 128 | C 64) -> the dict of proof.json"""
    p = bytes(proof256)
    if len(p) != 256:
        raise ValueError("a proof is 256 bytes")
    return {"pi_a": g1_json(p[0:64]), "pi_b": g2_json(p[64:192]), "pi_c": g1_json(p[192:256]), "protocol": "groth16", "curve": "bn128"}


def public(public_inputs):
    """n_pub field elements (ints, or 32-byte little-endian values as the calls return them) -> the list of public.json"""
    return [str(int(x)) if isinstance(x, int) else _le(x) for x in public_inputs]


def write(directory, vk_blob, proof256, public_inputs):
    """the three files side by side in `directory`; returns their paths"""
    os.makedirs(directory, exist_ok=True)
    out = {}
    for name, obj in (("verification_key.json", verification_key(vk_blob)), ("public.json", public(public_inputs)),
                      ("proof.json", proof(proof256))):
        out[name] = os.path.join(directory, name)
        with open(out[name], "w") as f:
            json.dump(obj, f, indent=1)
    return out


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="OWVK0001 key + 256-byte proof + 32-byte LE public inputs -> snarkjs's three JSON files")
    ap.add_argument("vk"), ap.add_argument("proof"), ap.add_argument("public"), ap.add_argument("outdir")
    a = ap.parse_args(argv)
    with open(a.vk, "rb") as f:
        vk = f.read()
    with open(a.proof, "rb") as f:
        pf = f.read()
    with open(a.public, "rb") as f:
        pub = f.read()
    if len(pub) % 32:
        raise SystemExit("public inputs: a multiple of 32 bytes")
    for path in write(a.outdir, vk, pf, [pub[i:i + 32] for i in range(0, len(pub), 32)]).values():
        print(path)


if __name__ == "__main__":
    main()
