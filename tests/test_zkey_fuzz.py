"""og_zkey_import / og_wtns_read fed what a file from outside may hold (the interpreter build: no GPU; under tools/sanitize_emu.sh
the same cases run on the ASan / UBSan builds): a valid .zkey with random bytes flipped, runs overwritten, sections cut,
lengths and counts rewritten -- every input is either refused with OG_ERR_INVALID (and a reason) or imported into a key that
og_pk_load takes; never a crash, never another error code.  The reference's convention for bytes from the network: refuse,
do not panic (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11, `Fp::from_repr` -> None)."""
import random
import struct

import pytest

from oracle.py import zkey as zo
from tests.r1cs_util import random_r1cs


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


def _mutants(good, rnd, n):
    secs, off = [], 12
    for _ in range(struct.unpack_from("<I", good, 8)[0]):
        sid, size = struct.unpack_from("<IQ", good, off)
        secs.append((sid, off, size))
        off += 12 + size
    for k in range(n):
        b = bytearray(good)
        kind = k % 7
        if kind == 0:                                   # a few random bit flips anywhere
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 1:                                 # a run of random bytes
            o = rnd.randrange(len(b))
            for i in range(o, min(len(b), o + rnd.randrange(1, 80))):
                b[i] = rnd.randrange(256)
        elif kind == 2:                                 # truncation
            b = b[:rnd.randrange(len(b))]
        elif kind == 3:                                 # a section's length field rewritten
            _sid, o, size = rnd.choice(secs)
            struct.pack_into("<Q", b, o + 4, rnd.choice([0, 1, size - 1, size + 1, size * 2, 1 << 40, (1 << 64) - 1]))
        elif kind == 4:                                 # a header count rewritten (nVars, nPublic, domainSize, nCoeffs)
            hdr = next(o for sid, o, _ in secs if sid == 2) + 12
            cof = next(o for sid, o, _ in secs if sid == 4) + 12
            o = rnd.choice([hdr + 72, hdr + 76, hdr + 80, cof])
            struct.pack_into("<I", b, o, rnd.choice([0, 1, 2, 3, 7, 8, 64, 1 << 20, (1 << 31) - 1, (1 << 32) - 1]))
        elif kind == 5:                                 # a section id rewritten (a missing / duplicated section), or the count
            _sid, o, _size = rnd.choice(secs)
            if rnd.random() < 0.5:
                struct.pack_into("<I", b, o, rnd.randrange(0, 14))
            else:
                struct.pack_into("<I", b, 8, rnd.choice([0, 1, 9, 11, 1000, (1 << 32) - 1]))
        else:                                           # a coefficient record rewritten
            cof = next(o for sid, o, _ in secs if sid == 4) + 12
            n_coef = struct.unpack_from("<I", good, cof)[0]
            o = cof + 4 + 44 * rnd.randrange(n_coef) + rnd.choice([0, 4, 8])
            struct.pack_into("<I", b, o, rnd.choice([0, 1, 2, 5, 1 << 16, (1 << 32) - 1]))
        yield kind, bytes(b)


def test_mutated_zkeys_are_refused_or_imported_never_worse(ectx):
    from owshen_amd import groth16 as g16, zkey as zk
    from owshen_amd.api import OwshenGpuError
    n_wires, cons, _z = random_r1cs(5, 1, seed=31337)
    rnd = random.Random(20260930)
    good = zo.write_zkey(zo.snarkjs_setup(n_wires, 1, cons, *(rnd.randrange(1, zo.R) for _ in range(5))))
    refused = imported = 0
    for kind, data in _mutants(good, rnd, 350):
        try:
            pk, vk = zk.import_zkey(ectx, data)
        except OwshenGpuError as e:
            assert e.code == -1 and "og_zkey_import" in str(e), (kind, str(e))
            refused += 1
            continue
        imported += 1                                   # e.g. a flipped bit inside a coefficient value, or in section 10
        g16.ProvingKey(ectx, pk).close()
        assert vk[:8] == b"OWVK0001"
    assert refused > 200 and refused + imported == 350, (refused, imported)


def test_mutated_wtns_are_refused_or_read(ectx):
    from owshen_amd import zkey as zk
    from owshen_amd.api import OwshenGpuError
    rnd = random.Random(7)
    good = zo.write_wtns([1] + [rnd.randrange(zo.R) for _ in range(20)])
    for k in range(300):
        b = bytearray(good)
        if k % 3 == 0:
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif k % 3 == 1:
            b = b[:rnd.randrange(len(b))]
        else:
            struct.pack_into("<Q" if k % 2 else "<I", b, rnd.choice([8, 12, 16, 24, 60, 64, 68]), rnd.choice([0, 1, 21, 22, 1 << 31, (1 << 32) - 1]))
        try:
            w = zk.read_wtns(bytes(b), lib=ectx._lib)
            assert w.shape[1] == 32
        except OwshenGpuError as e:
            assert e.code == -1 and "og_wtns_read" in str(e)
