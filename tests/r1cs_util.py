"""Shared test helper: random satisfiable R1CS instances + witnesses, in the oracle's dict-row form."""
import random

from oracle.py.fields import R


def random_r1cs(n_constraints, n_pub, seed, n_free=3, bool_every=0):
    """Each constraint k defines a new wire: w_new = (LC_a) * (LC_b), with the LCs over earlier wires.
    Wires: 0 = one, 1..n_pub public (the first n_pub private-free values are promoted to public),
    then free inputs, then one product wire per constraint.  Returns (n_wires, constraints, witness)."""
    rnd = random.Random(seed)
    z = [1] + [rnd.randrange(R) for _ in range(n_pub + n_free)]
    if bool_every:
        for i in range(1, len(z), bool_every):
            z[i] = rnd.randrange(2)
    constraints = []
    for _ in range(n_constraints):
        def lc():
            k = rnd.randrange(1, 4)
            return {rnd.randrange(len(z)): rnd.choice([1, 2, R - 1, rnd.randrange(R)]) for _ in range(k)}
        a, b = lc(), lc()
        av = sum(v * z[i] for i, v in a.items()) % R
        bv = sum(v * z[i] for i, v in b.items()) % R
        z.append(av * bv % R)
        constraints.append((a, b, {len(z) - 1: 1}))
    return len(z), constraints, z


def csr_from_rows(rows):
    """list of {col: coeff} -> (ptr u32, col u32, val u8 [nnz, 32]) -- independent of the product's builder."""
    import numpy as np
    ptr, col, val = [0], [], []
    for r in rows:
        for c in sorted(r):
            if r[c] % R:
                col.append(c)
                val.append((r[c] % R).to_bytes(32, "little"))
        ptr.append(len(col))
    v = np.frombuffer(b"".join(val), dtype=np.uint8).reshape(-1, 32).copy() if val else np.zeros((0, 32), np.uint8)
    return np.array(ptr, dtype=np.uint32), np.array(col, dtype=np.uint32), v


def oracle_c_key(r1cs_o, pk_o):
    """oracle.py R1CS + key -> oracle.c PreparedKey (the C restatement / CPU baseline)."""
    import numpy as np
    from oracle.c import binding as oc
    from oracle.py.curve import g1_to_bytes, g2_to_bytes
    rows = r1cs_o.rows()
    csr = {"a": csr_from_rows([r[0] for r in rows]), "b": csr_from_rows([r[1] for r in rows]),
           "c": csr_from_rows([r[2] for r in rows])}

    def pts(lst, f, w):
        return np.frombuffer(b"".join(f(p) for p in lst), dtype=np.uint8).reshape(-1, w).copy() if lst else np.zeros((0, w), np.uint8)
    points = {
        "alpha_g1": pts([pk_o["alpha_g1"]], g1_to_bytes, 64), "beta_g1": pts([pk_o["beta_g1"]], g1_to_bytes, 64),
        "beta_g2": pts([pk_o["beta_g2"]], g2_to_bytes, 128), "delta_g1": pts([pk_o["delta_g1"]], g1_to_bytes, 64),
        "delta_g2": pts([pk_o["delta_g2"]], g2_to_bytes, 128),
        "a_query": pts(pk_o["a_query"], g1_to_bytes, 64), "b_g1_query": pts(pk_o["b_g1_query"], g1_to_bytes, 64),
        "b_g2_query": pts(pk_o["b_g2_query"], g2_to_bytes, 128), "l_query": pts(pk_o["l_query"], g1_to_bytes, 64),
        "h_query": pts(pk_o["h_query"], g1_to_bytes, 64),
    }
    d = r1cs_o.domain_size
    return oc.PreparedKey(r1cs_o.n_wires, r1cs_o.n_pub, d.bit_length() - 1, len(rows), csr, points)


def oracle_c_key_from_blob(blob):
    from oracle.c import binding as oc
    return oc.prepared_key_from_blob(blob)
