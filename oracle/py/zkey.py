"""snarkjs' key and witness files (`.zkey`, `.wtns`: iden3 "binfile" containers) and snarkjs' Groth16 prover, restated.
TEST INFRASTRUCTURE ONLY -- the product's importer is owshen_amd/csrc/zkey.hip (og_zkey_import / og_wtns_read); this file is its
oracle and the writer the tests use.

No reference counterpart: the snapshot under /root/reference holds no prover and no key.  What is
restated here is the PUBLISHED format and algorithm of the lineage BASELINE.json's north_star names (circom / snarkjs 0.7.x,
ffjavascript 0.3.x), from the builder's knowledge of those sources -- there is no network, no snarkjs and no snarkjs-made file
in this environment, so nothing below has met a file snarkjs wrote ("parity unpinned").  The functions
named in the comments are where each fact lives upstream.

Container (@iden3/binfileutils `readBinFile` / `createBinFile`): 4 magic bytes | u32 version | u32 nSections, then per section
u32 id | u64 byte length | payload.  Everything little-endian.

`.zkey`, magic "zkey", version 1, Groth16 (snarkjs src/zkey_new.js `newZKey`, src/zkey_utils.js `readHeader` / `readZKey`):
  1  u32 protocol id = 1 (Groth16)
  2  u32 n8q | q | u32 n8r | r | u32 nVars | u32 nPublic | u32 domainSize | alpha1 | beta1 | beta2 | gamma2 | delta1 | delta2
  3  IC: nPublic + 1 G1 points
  4  u32 nCoeffs, then nCoeffs x (u32 matrix (0 = A, 1 = B) | u32 constraint | u32 signal | value): the C matrix is NOT stored --
     snarkjs' prover takes C z = (A z) o (B z) on the domain.  The list includes the nPublic + 1 rows `A[nConstraints + s][s] = 1`.
     value = coefficient x R^2 mod r as a plain integer (R = 2^256; zkey_utils.js `readFr2` multiplies by `Rri2` = R^-2 on the way
     in; the prover multiplies the raw bytes, taken as a Montgomery number, with the raw normal-form witness in one Montgomery
     product and gets the Montgomery form of the product)
  5  A: nVars G1      6  B1: nVars G1      7  B2: nVars G2      8  C (the L query): nVars - nPublic - 1 G1
  9  H: domainSize G1 points, H[i] = L_{2i+1}(tau) / delta . G1 with L the Lagrange basis of the size-2 domainSize domain:
     the ODD points (zkey_new.js copies `tauG1` Lagrange points 2i + 1 of power cirPower + 1 from the .ptau), so that
     sum_i (A B - C)(psi^(2i+1)) H[i] = (h Z)(tau) / delta with NO division by Z and no inverse transform (groth16_prove.js)
  10 64-byte circuit hash | u32 nContributions | contributions
 Points: affine, coordinates as n8q-byte little-endian MONTGOMERY numbers (x R mod q; ffjavascript `toRprLEM`), Fq2 as c0 | c1,
 the point at infinity as zeros.
 Roots of unity (ffjavascript F1Field / FFT): nqr = the smallest quadratic non-residue (5 for BN254's r), w[28] = nqr^((r-1)/2^28),
 w[k] = w[k+1]^2; constraint c is evaluation point w[power]^c.  This repository's domain uses 7^((r-1)/n) (the reference's
 `Fp` generator, /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:9): the same SET of points in another order,
 so an import is a permutation of the rows.

`.r1cs`, magic "r1cs", version 1 (iden3 r1csfile; circom writes it, `snarkjs groth16 setup` reads it): section 1 = u32 n8 | prime |
 u32 nWires | u32 nPubOut | u32 nPubIn | u32 nPrvIn | u64 nLabels | u32 mConstraints; section 2 = per constraint the linear
 combinations A, B, C, each u32 n | n x (u32 wire | n8-byte little-endian NORMAL-form coefficient); section 3 = wire -> label (u64
 each).  Wire 0 is the constant 1, then the public outputs and inputs.

`.wtns`, magic "wtns", version 2 (snarkjs src/wtns_utils.js): section 1 = u32 n8 | prime | u32 nWitness; section 2 = nWitness x n8
 little-endian NORMAL-form values.
"""
import struct

from .fields import R, P, inv, fr_root_of_unity
from .curve import G1, G2, G1_GEN, G2_GEN

MONT_R = 1 << 256
R2_R = MONT_R * MONT_R % R           # the coefficient section's factor
RI_Q = inv(MONT_R % P, P)            # point coordinates: x R mod q on file


# ---- ffjavascript's roots of unity ----------------------------------------------------------------------------------------
def ff_nqr(p=R):
    """ffjavascript F1Field: the smallest quadratic non-residue, from 2 up"""
    a = 2
    while pow(a, (p - 1) // 2, p) != p - 1:
        a += 1
    return a


FF_NQR = ff_nqr()
FF_W28 = pow(FF_NQR, (R - 1) >> 28, R)
assert pow(FF_W28, 1 << 27, R) == R - 1


def ff_root(log_n):
    """ffjavascript's primitive 2^log_n-th root of unity: Fr.w[log_n]"""
    assert 0 <= log_n <= 28
    return pow(FF_W28, 1 << (28 - log_n), R)


def dlog_pow2(base, target, log_n):
    """k (mod 2^log_n) with base^k = target, both of exact order 2^log_n: bit by bit"""
    k = 0
    binv = inv(base, R)
    for b in range(log_n):
        t = target * pow(binv, k, R) % R          # order divides 2^(log_n - b)
        if pow(t, 1 << (log_n - 1 - b), R) != 1:
            k |= 1 << b
    assert pow(base, k, R) == target
    return k


def constraint_of_row(log_d):
    """this repository's row i sits at 7^((r-1)/d)^i; snarkjs' constraint c at w^c.  Returns k (odd) with row i <-> constraint
    i k mod d."""
    if log_d == 0:
        return 1
    return dlog_pow2(ff_root(log_d), fr_root_of_unity(log_d), log_d)


# ---- containers -----------------------------------------------------------------------------------------------------------
def read_binfile(data, magic, max_version):
    data = bytes(data)
    if len(data) < 12 or data[:4] != magic:
        raise ValueError(f"not a {magic.decode()} file")
    version, n_sections = struct.unpack_from("<II", data, 4)
    if version > max_version:
        raise ValueError(f"{magic.decode()} version {version} not supported")
    off, sections = 12, {}
    for _ in range(n_sections):
        if off + 12 > len(data):
            raise ValueError("truncated section header")
        sid, size = struct.unpack_from("<IQ", data, off)
        off += 12
        if off + size > len(data):
            raise ValueError(f"section {sid} runs past the end of the file")
        sections.setdefault(sid, data[off:off + size])    # (binfileutils keeps every occurrence; the readers use the first)
        off += size
    return version, sections


def write_binfile(magic, version, sections):
    out = [magic, struct.pack("<II", version, len(sections))]
    for sid, payload in sections:
        out.append(struct.pack("<IQ", sid, len(payload)))
        out.append(payload)
    return b"".join(out)


def _fq_m(x):
    return (x * MONT_R % P).to_bytes(32, "little")


def _fq_from_m(b):
    v = int.from_bytes(b, "little")
    if v >= P:
        raise ValueError("coordinate >= q")
    return v * RI_Q % P


def g1_to_lem(pt):
    return bytes(64) if pt is None else _fq_m(pt[0]) + _fq_m(pt[1])


def g2_to_lem(pt):
    return bytes(128) if pt is None else b"".join(_fq_m(v) for v in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]))


def g1_from_lem(b):
    if not any(b):
        return None
    pt = (_fq_from_m(b[:32]), _fq_from_m(b[32:64]))
    if not G1.is_on_curve(pt):
        raise ValueError("G1 point not on the curve")
    return pt


def g2_from_lem(b):
    if not any(b):
        return None
    v = [_fq_from_m(b[i * 32:i * 32 + 32]) for i in range(4)]
    pt = ((v[0], v[1]), (v[2], v[3]))
    if not G2.is_on_curve(pt):
        raise ValueError("G2 point not on the curve")
    return pt


def read_zkey(data):
    """-> dict: n_vars, n_public, domain_size, power, alpha1 .. delta2, ic, coeffs [(matrix, constraint, signal, value)], a, b1, b2,
    c, h (affine points as int tuples / None), cs_hash, n_contributions"""
    _v, sec = read_binfile(data, b"zkey", 1)
    for sid in range(1, 10):
        if sid not in sec:
            raise ValueError(f"zkey: section {sid} missing")
    if struct.unpack("<I", sec[1][:4])[0] != 1:
        raise ValueError("zkey: not a Groth16 key (protocol id != 1)")
    s, o = sec[2], 0

    def u32():
        nonlocal o
        v = struct.unpack_from("<I", s, o)[0]
        o += 4
        return v

    def take(n):
        nonlocal o
        b = s[o:o + n]
        if len(b) != n:
            raise ValueError("zkey: header truncated")
        o += n
        return b
    if u32() != 32 or int.from_bytes(take(32), "little") != P:
        raise ValueError("zkey: base field is not BN254's")
    if u32() != 32 or int.from_bytes(take(32), "little") != R:
        raise ValueError("zkey: scalar field is not BN254's")
    z = {"n_vars": u32(), "n_public": u32(), "domain_size": u32()}
    d = z["domain_size"]
    if d & (d - 1) or not d:
        raise ValueError("zkey: domain size is not a power of two")
    z["power"] = d.bit_length() - 1
    z["alpha1"], z["beta1"] = g1_from_lem(take(64)), g1_from_lem(take(64))
    z["beta2"], z["gamma2"] = g2_from_lem(take(128)), g2_from_lem(take(128))
    z["delta1"], z["delta2"] = g1_from_lem(take(64)), g2_from_lem(take(128))
    m, l = z["n_vars"], z["n_public"]

    def points(sid, n, width, conv):
        b = sec[sid]
        if len(b) != n * width:
            raise ValueError(f"zkey: section {sid} holds {len(b)} bytes, expected {n * width}")
        return [conv(b[i * width:(i + 1) * width]) for i in range(n)]
    z["ic"] = points(3, l + 1, 64, g1_from_lem)
    z["a"] = points(5, m, 64, g1_from_lem)
    z["b1"] = points(6, m, 64, g1_from_lem)
    z["b2"] = points(7, m, 128, g2_from_lem)
    z["c"] = points(8, m - l - 1, 64, g1_from_lem)
    z["h"] = points(9, d, 64, g1_from_lem)
    cs = sec[4]
    n_coef = struct.unpack_from("<I", cs, 0)[0]
    if len(cs) != 4 + n_coef * 44:
        raise ValueError("zkey: coefficient section length does not match its count")
    ri2 = inv(R2_R, R)
    coeffs = []
    for i in range(n_coef):
        mt, c, sg = struct.unpack_from("<III", cs, 4 + i * 44)
        v = int.from_bytes(cs[4 + i * 44 + 12:4 + i * 44 + 44], "little")
        if mt > 1 or c >= d or sg >= m or v >= R:
            raise ValueError(f"zkey: coefficient {i} out of range")
        coeffs.append((mt, c, sg, v * ri2 % R))
    z["coeffs"] = coeffs
    if 10 in sec and len(sec[10]) >= 68:
        z["cs_hash"] = sec[10][:64]
        z["n_contributions"] = struct.unpack_from("<I", sec[10], 64)[0]
    return z


def write_zkey(z):
    """the inverse of read_zkey (same dict)"""
    m, l, d = z["n_vars"], z["n_public"], z["domain_size"]
    assert len(z["ic"]) == l + 1 and len(z["a"]) == m and len(z["b1"]) == m and len(z["b2"]) == m
    assert len(z["c"]) == m - l - 1 and len(z["h"]) == d
    hdr = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<I", 32) + R.to_bytes(32, "little")
    hdr += struct.pack("<III", m, l, d)
    hdr += g1_to_lem(z["alpha1"]) + g1_to_lem(z["beta1"]) + g2_to_lem(z["beta2"]) + g2_to_lem(z["gamma2"])
    hdr += g1_to_lem(z["delta1"]) + g2_to_lem(z["delta2"])
    cs = [struct.pack("<I", len(z["coeffs"]))]
    for mt, c, sg, v in z["coeffs"]:
        cs.append(struct.pack("<III", mt, c, sg) + (v % R * R2_R % R).to_bytes(32, "little"))
    sections = [
        (1, struct.pack("<I", 1)), (2, hdr), (3, b"".join(g1_to_lem(p) for p in z["ic"])), (4, b"".join(cs)),
        (5, b"".join(g1_to_lem(p) for p in z["a"])), (6, b"".join(g1_to_lem(p) for p in z["b1"])),
        (7, b"".join(g2_to_lem(p) for p in z["b2"])), (8, b"".join(g1_to_lem(p) for p in z["c"])),
        (9, b"".join(g1_to_lem(p) for p in z["h"])),
        (10, z.get("cs_hash", bytes(64)) + struct.pack("<I", 0)),
    ]
    return write_binfile(b"zkey", 1, sections)


def read_r1cs(data):
    """-> (n_wires, n_pub, constraints as a list of (a, b, c) row dicts)"""
    _v, sec = read_binfile(data, b"r1cs", 1)
    if 1 not in sec or 2 not in sec:
        raise ValueError("r1cs: section missing")
    h = sec[1]
    if len(h) != 64 or struct.unpack_from("<I", h, 0)[0] != 32 or int.from_bytes(h[4:36], "little") != R:
        raise ValueError("r1cs: not BN254's scalar field")
    n_wires, n_out, n_in, _n_prv = struct.unpack_from("<IIII", h, 36)
    n_cons = struct.unpack_from("<I", h, 60)[0]
    b, o, cons = sec[2], 0, []
    for _ in range(n_cons):
        rows = []
        for _k in range(3):
            n = struct.unpack_from("<I", b, o)[0]
            o += 4
            row = {}
            for _i in range(n):
                w = struct.unpack_from("<I", b, o)[0]
                v = int.from_bytes(b[o + 4:o + 36], "little")
                if w >= n_wires or v >= R:
                    raise ValueError("r1cs: entry out of range")
                row[w] = (row.get(w, 0) + v) % R
                o += 36
            rows.append(row)
        cons.append(tuple(rows))
    if o != len(b):
        raise ValueError("r1cs: constraint section length")
    return n_wires, n_out + n_in, cons


def write_r1cs(n_wires, n_pub, constraints, n_outputs=0):
    """circom's layout: `n_outputs` of the n_pub public wires are outputs (they come first)"""
    body = []
    for rows in constraints:
        for row in rows:
            body.append(struct.pack("<I", len(row)))
            for w in sorted(row):
                body.append(struct.pack("<I", w) + (row[w] % R).to_bytes(32, "little"))
    hdr = struct.pack("<I", 32) + R.to_bytes(32, "little") + struct.pack("<IIIIQI", n_wires, n_outputs, n_pub - n_outputs, n_wires - n_pub - 1,
                                                                          n_wires, len(constraints))
    return write_binfile(b"r1cs", 1, [(1, hdr), (2, b"".join(body)), (3, b"".join(struct.pack("<Q", w) for w in range(n_wires)))])


def read_wtns(data):
    _v, sec = read_binfile(data, b"wtns", 2)
    if 1 not in sec or 2 not in sec:
        raise ValueError("wtns: section missing")
    n8 = struct.unpack_from("<I", sec[1], 0)[0]
    if n8 != 32 or int.from_bytes(sec[1][4:36], "little") != R:
        raise ValueError("wtns: not BN254's scalar field")
    n = struct.unpack_from("<I", sec[1], 36)[0]
    if len(sec[2]) != n * 32:
        raise ValueError("wtns: witness section length does not match its count")
    w = [int.from_bytes(sec[2][i * 32:i * 32 + 32], "little") for i in range(n)]
    if any(v >= R for v in w):
        raise ValueError("wtns: value >= r")
    return w


def write_wtns(w):
    s1 = struct.pack("<I", 32) + R.to_bytes(32, "little") + struct.pack("<I", len(w))
    return write_binfile(b"wtns", 2, [(1, s1), (2, b"".join(int(v % R).to_bytes(32, "little") for v in w))])


# ---- a key the way snarkjs makes one (zkey_new.js over a .ptau), from explicit toxic waste ---------------------------------
def _lagrange_at(tau, log_n, root):
    n = 1 << log_n
    zt = (pow(tau, n, R) - 1) * inv(n, R) % R
    out, wk = [], 1
    for _ in range(n):
        out.append(zt * wk % R * inv((tau - wk) % R, R) % R)
        wk = wk * root % R
    return out


def _mul_many(scalars, g2=False):
    """scalars -> k . generator, through the C restatement's fixed-base routine when it is built (fast), else Python ints"""
    try:
        import numpy as np
        from ..c import binding as oc
        from .curve import g1_to_bytes, g2_to_bytes, g1_from_bytes, g2_from_bytes
        sc = np.frombuffer(b"".join(int(k % R).to_bytes(32, "little") for k in scalars), dtype=np.uint8).reshape(-1, 32).copy()
        if g2:
            out = oc.fixed_base_g2(np.frombuffer(g2_to_bytes(G2_GEN), dtype=np.uint8), sc)
            return [g2_from_bytes(out[i].tobytes()) for i in range(len(scalars))]
        out = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), sc)
        return [g1_from_bytes(out[i].tobytes()) for i in range(len(scalars))]
    except (ImportError, OSError):
        grp, gen = (G2, G2_GEN) if g2 else (G1, G1_GEN)
        tab = grp.fixed_base_table(gen)
        return [grp.fixed_base_mul(tab, k % R) for k in scalars]


def snarkjs_setup(n_wires, n_pub, constraints, tau, alpha, beta, gamma, delta, row_map=None):
    """the zkey snarkjs would hold for the R1CS `constraints` (list of (a, b, c) row dicts; wire 0 = 1, wires 1..n_pub public)
    after a ceremony whose secrets are (tau, alpha, beta, gamma, delta): constraints in order at w^0, w^1, .. (ffjavascript's
    root), then the n_pub + 1 public-input rows, the C matrix only inside the points.  `row_map` (a permutation of the domain)
    places constraint i at w^row_map[i] instead -- the tests use it to lay THIS repository's key out as a zkey."""
    n_rows = len(constraints) + n_pub + 1
    power = max(1, (n_rows - 1).bit_length())      # (this repository's minimum domain is 2; snarkjs: log2(n_rows - 1) + 1)
    d = 1 << power
    rows = list(constraints) + [({i: 1}, {}, {}) for i in range(n_pub + 1)]
    at = (lambda i: i) if row_map is None else (lambda i: row_map[i])
    lag = _lagrange_at(tau, power, ff_root(power))
    a, b, c = [0] * n_wires, [0] * n_wires, [0] * n_wires
    coeffs = []
    for i, (ra, rb, rc) in enumerate(rows):
        lk = lag[at(i)]
        for s, v in ra.items():
            a[s] = (a[s] + v * lk) % R
            if v % R:
                coeffs.append((0, at(i), s, v % R))
        for s, v in rb.items():
            b[s] = (b[s] + v * lk) % R
            if v % R:
                coeffs.append((1, at(i), s, v % R))
        for s, v in rc.items():
            c[s] = (c[s] + v * lk) % R
    gi, di = inv(gamma, R), inv(delta, R)
    kk = [(beta * a[i] + alpha * b[i] + c[i]) % R for i in range(n_wires)]
    lag2 = _lagrange_at(tau, power + 1, ff_root(power + 1))
    hs = [lag2[2 * i + 1] * di % R for i in range(d)]
    g1s = _mul_many([alpha, beta, delta] + [kk[i] * gi % R for i in range(n_pub + 1)] + a + b
                    + [kk[i] * di % R for i in range(n_pub + 1, n_wires)] + hs)
    g2s = _mul_many([beta, gamma, delta] + b, g2=True)
    o = 3
    ic, o = g1s[o:o + n_pub + 1], o + n_pub + 1
    aq, o = g1s[o:o + n_wires], o + n_wires
    b1q, o = g1s[o:o + n_wires], o + n_wires
    cq, o = g1s[o:o + n_wires - n_pub - 1], o + n_wires - n_pub - 1
    hq = g1s[o:o + d]
    return {"n_vars": n_wires, "n_public": n_pub, "domain_size": d, "power": power, "alpha1": g1s[0], "beta1": g1s[1], "delta1": g1s[2],
            "beta2": g2s[0], "gamma2": g2s[1], "delta2": g2s[2], "ic": ic, "coeffs": coeffs, "a": aq, "b1": b1q, "b2": g2s[3:], "c": cq, "h": hq}


# ---- snarkjs' prover, restated (src/groth16_prove.js) -----------------------------------------------------------------------
def _fft(a, root):
    """natural order in and out: out[i] = sum_j a[j] root^(i j)"""
    n = len(a)
    if n == 1:
        return list(a)
    ev, od = _fft(a[0::2], root * root % R), _fft(a[1::2], root * root % R)
    out, t = [0] * n, 1
    for i in range(n // 2):
        x = od[i] * t % R
        out[i], out[i + n // 2] = (ev[i] + x) % R, (ev[i] - x) % R
        t = t * root % R
    return out


def snarkjs_prove(z, w, r, s):
    """`buildABC1` (A z and B z per constraint, C := A z o B z), `ifft`, `batchApplyKey(.., 1, w[power + 1])`, `fft`, `joinABC`
    (a b - c on the odd coset), then the five multiexps and the blinding -- the proof (A, B, C) as affine points"""
    d, power, m, l = z["domain_size"], z["power"], z["n_vars"], z["n_public"]
    assert len(w) == m and w[0] == 1
    ae, be = [0] * d, [0] * d
    for mt, c, sg, v in z["coeffs"]:
        if mt == 0:
            ae[c] = (ae[c] + v * w[sg]) % R
        else:
            be[c] = (be[c] + v * w[sg]) % R
    ce = [x * y % R for x, y in zip(ae, be)]
    root, inc, ninv = ff_root(power), ff_root(power + 1), inv(d, R)
    odd = []
    for ev in (ae, be, ce):
        co = [x * ninv % R for x in _fft(ev, inv(root, R))]
        sh, t = [], 1
        for x in co:
            sh.append(x * t % R)
            t = t * inc % R
        odd.append(_fft(sh, root))
    p_odd = [(x * y - c) % R for x, y, c in zip(*odd)]
    A = G1.add(G1.add(z["alpha1"], G1.msm(w, z["a"])), G1.mul(z["delta1"], r))
    B2 = G2.add(G2.add(z["beta2"], G2.msm(w, z["b2"])), G2.mul(z["delta2"], s))
    B1 = G1.add(G1.add(z["beta1"], G1.msm(w, z["b1"])), G1.mul(z["delta1"], s))
    C = G1.add(G1.msm(w[l + 1:], z["c"]), G1.msm(p_odd, z["h"]))
    C = G1.add(C, G1.mul(A, s))
    C = G1.add(C, G1.mul(B1, r))
    C = G1.add(C, G1.neg(G1.mul(z["delta1"], r * s % R)))
    return A, B2, C


def vk_of(z):
    """the verifying key oracle/py/groth16.verify takes"""
    return {"alpha_g1": z["alpha1"], "beta_g2": z["beta2"], "gamma_g2": z["gamma2"], "delta_g2": z["delta2"], "ic": z["ic"]}


# ---- the import: a zkey as this repository's key ("OWPK0001" / "OWVK0001", include/owshen_gpu.h) ----------------------------
def h_query_from_odd_lagrange(hp, power, msm=None):
    """H[j] = tau^j Z(tau) / delta . G1 (j < d - 1), the coefficient-basis H query, from the zkey's odd-coset Lagrange points:
    x^j Z(x) vanishes on the domain and equals -2 psi^j w^(i j) at psi^(2i+1), so  H[j] = -2 psi^j sum_i w^(i j) H'[i]  -- a DFT
    over group elements.  Here as d direct sums (the product runs it as an FFT: zkey.hip)."""
    d = 1 << power
    msm = msm or (lambda sc, pts: G1.msm(sc, pts))
    w, psi = ff_root(power), ff_root(power + 1)
    out = []
    for j in range(d - 1):
        wj = pow(w, j, R)
        sc, t = [], (-2 * pow(psi, j, R)) % R
        for _ in range(d):
            sc.append(t)
            t = t * wj % R
        out.append(msm(sc, hp))
    return out


def zkey_to_owshen(z, msm=None, constraints=None):
    """-> (OWPK0001 bytes with header flag 1 = "C z is (A z) o (B z)", OWVK0001 bytes).  Constraint c moves to row c k^-1 mod d
    (constraint_of_row); rows keep their columns ascending, duplicates summed, zeros dropped (oracle/py/keygen._csr).
    constraints (the key's .r1cs, read_r1cs): its C matrix rides in the key and the flag is 0."""
    from .curve import g1_to_bytes, g2_to_bytes
    m, l, d, power = z["n_vars"], z["n_public"], z["domain_size"], z["power"]
    if power < 1:
        raise ValueError("domain of one point")
    kinv = inv(constraint_of_row(power), d) if d > 1 else 1
    rows = [({}, {}) for _ in range(d)]
    for mt, c, sg, v in z["coeffs"]:
        row = rows[c * kinv % d][mt]
        row[sg] = (row.get(sg, 0) + v) % R

    def pad32(b):
        return b + b"\0" * (-len(b) % 32)
    mats = []
    for which in (0, 1):
        ptr, col, val = [0], [], []
        for row in rows:
            for sg in sorted(row[which]):
                if row[which][sg]:
                    col.append(sg)
                    val.append(row[which][sg].to_bytes(32, "little"))
            ptr.append(len(col))
        mats.append((struct.pack(f"<{d + 1}I", *ptr), struct.pack(f"<{len(col)}I", *col), b"".join(val), len(col)))
    crow = [{} for _ in range(d)]
    for c, (_a, _b, rc) in enumerate(constraints or []):
        for sg, v in rc.items():
            crow[c * kinv % d][sg] = (crow[c * kinv % d].get(sg, 0) + v) % R
    ptr, col, val = [0], [], []
    for row in crow:
        for sg in sorted(row):
            if row[sg]:
                col.append(sg)
                val.append(row[sg].to_bytes(32, "little"))
        ptr.append(len(col))
    mats.append((struct.pack(f"<{d + 1}I", *ptr), struct.pack(f"<{len(col)}I", *col), b"".join(val), len(col)))
    hq = h_query_from_odd_lagrange(z["h"], power, msm)
    pk = b"OWPK0001" + struct.pack("<9Q", m, l, power, d, mats[0][3], mats[1][3], mats[2][3], 0 if constraints is not None else 1, 0)
    pk += g1_to_bytes(z["alpha1"]) + g1_to_bytes(z["beta1"]) + g1_to_bytes(z["delta1"]) + bytes(64)
    pk += g2_to_bytes(z["beta2"]) + g2_to_bytes(z["delta2"])
    for ptr, col, val, _n in mats:
        pk += pad32(ptr) + pad32(col) + pad32(val)
    pk += pad32(b"".join(g1_to_bytes(p) for p in z["a"])) + pad32(b"".join(g1_to_bytes(p) for p in z["b1"]))
    pk += pad32(b"".join(g2_to_bytes(p) for p in z["b2"])) + pad32(b"".join(g1_to_bytes(p) for p in z["c"]))
    pk += pad32(b"".join(g1_to_bytes(p) for p in hq))
    vk = b"OWVK0001" + struct.pack("<Q", l) + g1_to_bytes(z["alpha1"]) + g2_to_bytes(z["beta2"]) + g2_to_bytes(z["gamma2"])
    vk += g2_to_bytes(z["delta2"]) + b"".join(g1_to_bytes(p) for p in z["ic"])
    return pk, vk
