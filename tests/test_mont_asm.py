"""The generated gfx950 Montgomery sequences (owshen_amd/csrc/mont_gfx950.inc), checked without a GPU:
* the committed file is what tools/gen_mont_asm.py writes;
* every routine, executed instruction by instruction on Python integers (the six opcodes it uses), returns exactly
  (sum of its products + addend * 2^261) / 2^261 as the column algorithm defines it -- value congruent mod N, limbs
  normalized, below the documented bound -- for random and adversarial limb patterns in both fields.
The GPU tests (test_gpu_field_mimc7.py and everything above it) then run the same text on the hardware."""
import importlib.util
import os
import random
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_mont_asm", os.path.join(ROOT, "tools", "gen_mont_asm.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MASK = (1 << 29) - 1
M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def limbs(v):
    return [(v >> (29 * i)) & MASK for i in range(9)]


def value(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def run(ins, regs):
    """regs: dict name -> int; 'v30'/'v31' are the accumulator halves.  Returns regs."""
    def rd(tok):
        tok = tok.strip()
        if tok.startswith("%["):
            return regs[tok[2:-1]]
        if tok == "v[30:31]":
            return regs["v30"] | (regs["v31"] << 32)
        if tok in ("v30", "v31"):
            return regs[tok]
        return int(tok, 0)

    def wr(tok, v):
        tok = tok.strip()
        if tok == "v[30:31]":
            assert 0 <= v <= M64, "64-bit column accumulator overflow"
            regs["v30"], regs["v31"] = v & M32, v >> 32
        elif tok.startswith("%["):
            regs[tok[2:-1]] = v & M32
        else:
            regs[tok] = v & M32

    for line in ins:
        op, rest = line.split(" ", 1)
        a = [t.strip() for t in rest.split(",")]
        if op == "v_mad_u64_u32":      # dst, vcc, s0, s1, s2
            x, y = rd(a[2]), rd(a[3])
            assert x <= M32 and y <= M32
            wr(a[0], x * y + rd(a[4]))  # wr asserts no 64-bit overflow (the carry-out is unused)
        elif op == "v_mul_lo_u32":
            wr(a[0], (rd(a[1]) * rd(a[2])) & M32)
        elif op == "v_and_b32":
            wr(a[0], rd(a[1]) & rd(a[2]))
        elif op == "v_lshrrev_b64":
            wr(a[0], rd(a[2]) >> rd(a[1]))
        elif op == "v_alignbit_b32":    # ({s0, s1} >> s2[4:0]) low 32 bits
            wr(a[0], (((rd(a[1]) << 32) | rd(a[2])) >> (rd(a[3]) & 31)) & M32)
        elif op == "v_add_u32":
            wr(a[0], (rd(a[1]) + rd(a[2])) & M32)
        else:
            raise AssertionError(f"unexpected opcode {op}")
    return regs


def test_generated_file_is_current():
    text = open(gen.OUT).read()
    for name, (terms, plus) in gen.ROUTINES.items():
        body = re.search(r"#define OG_MONT_ASM_%s \\\n((?:  \".*\n)+)" % name, text)
        assert body, name
        got = [re.match(r'"(.*?)(?:\\n\\t)?"', l.strip()).group(1) for l in body.group(1).splitlines()]
        assert got == gen.routine(terms, plus), f"{name}: mont_gfx950.inc is stale -- run python tools/gen_mont_asm.py"


def operand_sets(rng, n_mod, kind):
    """values for one Fe operand: normalized < 2N, or lazy (limbs < 2^30, e.g. 8N - x limb-wise)"""
    if kind == "norm":
        return limbs(rng.randrange(0, 2 * n_mod))
    if kind == "max":
        return limbs(2 * n_mod - 1)
    if kind == "allones":
        return [MASK] * 8 + [limbs(2 * n_mod - 1)[8]]
    if kind == "zero":
        return [0] * 9
    if kind == "lazy":  # 8N - x limb-wise with borrowed 2^29s (fe_neg_lazy): limbs in (0, 2^30), value <= 8N
        x, n8 = limbs(rng.randrange(0, 2 * n_mod)), limbs(8 * n_mod)
        return [n8[j] + ((1 << 29) if j < 8 else 0) - (1 if j > 0 else 0) - x[j] for j in range(9)]
    raise AssertionError(kind)


@pytest.mark.parametrize("mod_name", ["Fq", "Fr"])
@pytest.mark.parametrize("name", list(gen.ROUTINES))
def test_routine_semantics(name, mod_name):
    n_mod = Q if mod_name == "Fq" else R
    terms, plus = gen.ROUTINES[name]
    ins = gen.routine(terms, plus)
    rng = random.Random(hash((name, mod_name)) & 0xFFFF)
    inv = (-pow(n_mod, -1, 1 << 29)) % (1 << 29)
    big_r = 1 << 261
    names = sorted({t for term in terms for t in term[1:]})
    for trial in range(40):
        kinds = ["norm", "max", "allones", "zero"]
        vals = {}
        for nm in names:
            k = "norm" if trial >= 12 else kinds[(trial + ord(nm)) % 4]
            vals[nm] = operand_sets(rng, n_mod, k)
        # the documented operand contract: one operand of a product may be lazy (at most two lazy products in the 4-term forms)
        if trial % 3 == 0:
            for term in [t for t in terms if t[0] == "mul"][:2]:
                vals[term[1]] = operand_sets(rng, n_mod, "lazy")
                vals[term[2]] = operand_sets(rng, n_mod, "max" if trial % 2 else "norm")
        # keep the sum of products below 169 N^2 (the routines' contract): scale operands down when many terms
        regs = {"v30": 0xDEADBEEF, "v31": 0xDEADBEEF}
        for j in range(9):
            regs[f"n{j}"] = limbs(n_mod)[j]
            regs[f"r{j}"] = 0xDEADBEEF
        regs["inv"] = inv
        for nm, l in vals.items():
            for j in range(9):
                regs[f"{nm}{j}"] = l[j]
            for j in range(8):
                regs[f"{nm}d{j}"] = (l[j] << 1) & M32
        addend = [0] * 9
        if plus:
            x = limbs(rng.randrange(0, 2 * n_mod))
            neg4 = limbs(4 * n_mod)
            # the lazy 4N - x of field.hip.h: limb-wise with borrowed 2^29s, limbs < 2^30
            c = [neg4[j] + ((1 << 29) if j < 8 else 0) - (1 if 0 < j else 0) - x[j] for j in range(9)]
            assert value(c) == 4 * n_mod - value(x) and all(0 <= v < (1 << 30) for v in c)
            addend = c
            for j in range(9):
                regs[f"p{j}"] = c[j]
        total = 0
        for term in terms:
            if term[0] == "mul":
                total += value(vals[term[1]]) * value(vals[term[2]])
            else:
                total += value(vals[term[1]]) ** 2
        assert total < 169 * n_mod * n_mod
        run(ins, regs)
        out = [regs[f"r{j}"] for j in range(9)]
        assert all(0 <= v <= MASK for v in out[:8]) and out[8] < (1 << 29), (name, out)
        got = value(out)
        assert (got * big_r - total - value(addend) * big_r) % n_mod == 0
        assert got < 2 * n_mod + value(addend) + 1
        # exact: (total + m N) / R + addend with m the Montgomery multiplier
        m = (-total * pow(n_mod, -1, big_r)) % big_r
        assert got == (total + m * n_mod) // big_r + value(addend)


def _mads(name):
    return sum(1 for i in gen.routine(*gen.ROUTINES[name]) if i.startswith("v_mad_u64_u32"))


def _mul_los(name):
    return sum(1 for i in gen.routine(*gen.ROUTINES[name]) if i.startswith("v_mul_lo_u32"))


def test_instruction_counts_quoted_in_the_roofline_tooling():
    """tools/pmc_traffic.py turns SQ counters into `mad_issue_frac` with the multiply-add count of one mixed addition
    (ec.hip.h xyzz_madd_signed): G1 = P, R, D (product + addend), PP (square), PPP, X3 (square + product), Y3 (two products),
    ZZ3, ZZZ3; G2 = the Fq2 forms of the same, two reductions each."""
    g1 = 3 * _mads("MUL_PLUS") + _mads("SQR") + _mads("MUL") + _mads("SQR_ADD") + _mads("MUL_ADD") + 2 * _mads("MUL")
    g2 = (3 * 2 * _mads("MUL_ADD_PLUS")                      # P, R, D
          + _mads("SQR_ADD") + _mads("MUL")                  # PP = P^2 in Fq2
          + 2 * _mads("MUL_ADD")                             # PPP
          + _mads("SQR_ADD3") + _mads("MUL_ADD3")            # X3
          + 2 * _mads("MUL_ADD4")                            # Y3
          + 2 * 2 * _mads("MUL_ADD"))                        # ZZ3, ZZZ3
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
    pt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pt)
    assert pt.MADS["accumulate_g1"] == (g1, 9 * _mul_los("MUL")) == (1572, 81)
    assert pt.MADS["accumulate_g2"] == (g2, 18 * _mul_los("MUL")) == (4836, 162)
    assert _mads("MUL") == 162 and len(gen.routine(*gen.ROUTINES["MUL"])) == 205


def worst_lazy31(n_mod):
    """the laziest operand the radix-4 NTT hands fe_mul (field.hip.h's contract; ntt.hip k_ntt_block4): eight limbs at 2^31 - 1 and a
    top limb that takes the value to just under 42 N"""
    low = [(1 << 31) - 1] * 8
    top = (42 * n_mod - 1 - value(low + [0])) >> 232
    l = low + [top]
    assert 41 * n_mod < value(l) < 42 * n_mod and top < (1 << 31)
    return l


@pytest.mark.parametrize("mod_name", ["Fq", "Fr"])
def test_mul_takes_one_operand_with_limbs_up_to_2_31(mod_name):
    """the lazy butterflies of k_ntt_block4 multiply a sum whose limbs reach 2^31 (value < 42 N) by a normalized
    twiddle.  The generated MUL sequence must hold that in its single 64-bit column accumulator -- the interpreter asserts every
    accumulator write stays below 2^64 -- and return the exact Montgomery product: all-limbs-at-the-bound against the largest
    normalized partners, and random lazy operands"""
    n_mod = Q if mod_name == "Fq" else R
    terms, plus = gen.ROUTINES["MUL"]
    assert not plus and len(terms) == 1
    ins = gen.routine(terms, plus)
    _, an, bn = terms[0]
    rng = random.Random(31)
    inv = (-pow(n_mod, -1, 1 << 29)) % (1 << 29)
    big_r = 1 << 261
    partners = [limbs(2 * n_mod - 1), [MASK] * 8 + [limbs(2 * n_mod - 1)[8]], limbs(0), limbs(1)] + [limbs(rng.randrange(2 * n_mod)) for _ in range(20)]
    lazies = [worst_lazy31(n_mod)] + [[rng.randrange(1 << 31) for _ in range(8)] + [rng.randrange(worst_lazy31(n_mod)[8] + 1)] for _ in range(20)]
    for la in lazies:
        for pb in partners:
            for x, y in ((la, pb), (pb, la)):            # either operand may be the lazy one
                regs = {"v30": 0xDEADBEEF, "v31": 0xDEADBEEF, "inv": inv}
                for j in range(9):
                    regs[f"n{j}"], regs[f"r{j}"] = limbs(n_mod)[j], 0xDEADBEEF
                    regs[f"{an}{j}"], regs[f"{bn}{j}"] = x[j], y[j]
                run(ins, regs)
                out = [regs[f"r{j}"] for j in range(9)]
                total = value(x) * value(y)
                assert total < 169 * n_mod * n_mod
                m = (-total * pow(n_mod, -1, big_r)) % big_r
                assert all(0 <= v <= MASK for v in out[:8]) and value(out) == (total + m * n_mod) // big_r and value(out) < 2 * n_mod
