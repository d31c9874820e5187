"""Batched BabyJubJub EdDSA verification with the MiMC7 sponge against oracle/py/babyjubjub.py; shared by the
CPU-interpreter run and the GPU run."""
import random

import numpy as np

from oracle.py import babyjubjub as bj
from oracle.py.fields import R


def _rec(pk, rr, s, msg):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in (pk[0], pk[1], rr[0], rr[1], s, msg)), dtype=np.uint8).reshape(6, 32)


def case_eddsa_batch(ctx, n_valid=6, seed=1):
    rnd = random.Random(seed)
    recs, want = [], []
    # the reference's fixed-input test (babyjubjub/tests.rs:40-51: sk, randomness and message are small constants), on the real hash
    fixed = [(123456, 2345, 123456)]
    for k in range(n_valid):
        sk, rand_, msg = fixed[k] if k < len(fixed) else (rnd.randrange(1, R), rnd.randrange(R), rnd.randrange(R))
        while True:
            try:
                rr, s = bj.sign_mimc7(sk, rand_, msg)
                break
            except ValueError:   # s >= r: the reference errors out ("Invalid repr"); draw again
                rand_ = rnd.randrange(R)
        pk = bj.multiply(bj.BASE, sk)
        assert bj.verify_mimc7(pk, msg, (rr, s))
        recs.append(_rec(pk, rr, s, msg)); want.append(1)
        if k < 3:
            recs.append(_rec(pk, rr, s, (msg + 1) % R)); want.append(0)            # other message
            recs.append(_rec(pk, rr, (s + 1) % R, msg)); want.append(0)            # other s
            recs.append(_rec(pk, bj.affine_double(rr), s, msg)); want.append(0)    # other R (on the curve)
            recs.append(_rec(bj.affine_double(pk), rr, s, msg)); want.append(0)    # other key
            recs.append(_rec((pk[0], (pk[1] + 1) % R), rr, s, msg)); want.append(0)  # pk off the curve
            recs.append(_rec(pk, rr, s + R if s + R < (1 << 256) else s, msg)); want.append(0 if s + R < (1 << 256) else 1)  # non-canonical s
    # s = 0 with R = identity and h pk = identity cannot be forged without the key: identity R, s = 0 must reject for a random pk
    pk = bj.multiply(bj.BASE, 77)
    recs.append(_rec(pk, (0, 1), 0, 5)); want.append(1 if bj.verify_mimc7(pk, 5, ((0, 1), 0)) else 0)
    got = ctx.eddsa_verify(ctx.to_device(np.stack(recs)))
    assert got.tolist() == want
    for rec, w in zip(recs, want):   # the oracle agrees record by record (canonical ones)
        v = [int.from_bytes(rec[i].tobytes(), "little") for i in range(6)]
        if all(x < R for x in v):
            assert bj.verify_mimc7((v[0], v[1]), v[5], ((v[2], v[3]), v[4])) == bool(w)
