"""snarkjs' .zkey / .wtns files on the CPU interpreter (tests/hipemu); cases in tests/zkey_cases.py."""
import pytest

from tests import zkey_cases as cases


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


@pytest.mark.parametrize("n_constraints,n_pub", [(11, 2), (1, 0), (29, 1)])
def test_emu_zkey_import_matches_oracle_and_snarkjs_prover(ectx, n_constraints, n_pub):
    cases.case_import_matches_oracle_and_snarkjs_prover(ectx, n_constraints, n_pub)


def test_emu_own_key_through_a_zkey(ectx):
    cases.case_own_key_through_a_zkey(ectx, 12, 2)


def test_emu_zkey_export_is_what_snarkjs_would_prove_with(ectx):
    cases.case_export_is_what_snarkjs_would_prove_with(ectx, 13, 1)


def test_emu_wtns(ectx):
    cases.case_wtns(ectx._lib)


def test_emu_zkey_refusals(ectx):
    cases.case_refusals(ectx)


def test_emu_prove_files_and_the_second_engine(ectx, tmp_path):
    import shutil
    import subprocess
    paths = cases.case_prove_files(ectx, tmp_path)
    node = shutil.which("node")
    if node is None:
        pytest.skip("no node: the second pairing engine is not available")
    import os
    js = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "js", "bn254_pairing_second.js")
    out = subprocess.run([node, js, "--snarkjs", paths["verification_key.json"], paths["public.json"], paths["proof.json"]],
                         capture_output=True, text=True, timeout=600)
    assert out.stdout.strip().splitlines()[-1] == "OK", out.stdout + out.stderr


def test_emu_r1cs(ectx):
    cases.case_r1cs(ectx._lib)


def test_emu_zkey_import_with_r1cs(ectx):
    cases.case_import_with_r1cs(ectx, *(14, 2))
