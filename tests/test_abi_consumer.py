"""A compiled C consumer of the C ABI (tests/abi_consumer.c): the header is valid C99 and the program links against
libowshen_gpu.so on a CPU-only host; on the GPU box it runs and its proofs are compared with the same request made
through the ctypes mirror."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "abi_consumer")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "owshen_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_consumer.c"), "-o", EXE, "-L", lib_dir, "-lowshen_gpu",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])


def test_consumer_compiles_and_links_as_c99():
    _build()
    assert os.path.exists(EXE)


def _xorshift_stream():
    st = 0x9E3779B97F4A7C15
    mask = (1 << 64) - 1
    while True:
        st ^= st >> 12
        st ^= (st << 25) & mask
        st ^= st >> 27
        yield ((st * 2685821657736338717) & mask) >> 56


@pytest.mark.gpu
def test_consumer_proofs_equal_the_ctypes_path(ctx):
    from owshen_amd import circuit, groth16 as g16
    _build()
    depth, n_pad3, n_pad2, n = 3, 40, 500, 3
    out = subprocess.run([EXE, str(depth), str(n_pad3), str(n_pad2), str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[-2:] == ["verify: ok", "boundary: ok"] and len(lines) == n + 2
    got = [bytes.fromhex(ln.split()[2]) for ln in lines[:n]]
    # the same request through the Python mirror
    gen = _xorshift_stream()
    rec = (8 + depth) * 32
    inputs = np.array([next(gen) for _ in range(n * rec)], dtype=np.uint8).reshape(n, 8 + depth, 32)
    inputs[:, :, 31] &= 0x1F
    inputs[:, 5, 8:] = 0
    inputs[:, 5, :8] = (inputs[:, 5, :8].copy().view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
    inputs[:, 6, 20:] = 0
    inputs[:, 7, 8:] = 0
    rs = np.array([next(gen) for _ in range(n * 64)], dtype=np.uint8).reshape(n, 64)
    rs[:, 31] &= 0x1F
    rs[:, 63] &= 0x1F
    r1 = circuit.withdraw_r1cs_native(ctx, depth, n_pad3, n_pad2)
    blob, _vk = g16.setup(ctx, r1, 101, 102, 103, 104, 105)
    pk = g16.ProvingKey(ctx, blob)
    want = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(inputs), rs, n_pad3, n_pad2)
    for k in range(n):
        assert got[k] == want[k].tobytes()
    pk.close()
