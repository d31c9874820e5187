"""Groth16 key generation + batched proving on the CPU interpreter (tests/hipemu): the same cases the
GPU parity run executes (tests/groth16_cases.py)."""
import pytest

from tests import groth16_cases as cases


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


@pytest.mark.parametrize("n_constraints,n_pub", [(6, 1), (57, 3)])
def test_emu_setup_matches_oracle_setup(ectx, n_constraints, n_pub):
    cases.case_setup_matches_oracle_setup(ectx, n_constraints, n_pub)


def test_emu_prove_batch_matches_oracle_and_verifies(ectx):
    cases.case_prove_batch_matches_oracle_and_verifies(ectx)


def test_emu_unsatisfied_witness_is_rejected(ectx):
    cases.case_unsatisfied_witness_is_rejected(ectx)


def test_emu_noncanonical_witness_is_rejected(ectx):
    cases.case_noncanonical_witness_is_rejected(ectx)


def test_emu_pk_load_rejects_malformed_blobs(ectx):
    cases.case_pk_load_rejects_malformed_blobs(ectx)


def test_emu_medium_circuit_sub_batched(ectx):
    cases.case_medium_circuit_vs_c_oracle(ectx, 600, 3, sub_batch=2)


def test_emu_degenerate_circuits(ectx):
    cases.case_degenerate_circuits(ectx)


def test_emu_random_shapes(ectx):
    cases.case_random_shapes(ectx, range(1000, 1008))


def test_emu_stage_pipeline_and_tail_stream(ectx, monkeypatch):
    """the prep / math / tail-stream pipeline of prove_batch (taken from 64 proofs per sub-batch on) forced at toy size:
    9 proofs with at most 2 per sub-batch -> the ramped plan 1, 2, 2, 2, 2 (a scratch slot is reused from the fourth
    sub-batch on), persistent accumulation launches, the H query's sort on its own stream -> same bytes as the oracle"""
    monkeypatch.setenv("OG_SUB_BATCH", "2")
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    cases.case_medium_circuit_vs_c_oracle(ectx, 60, 9, None)


@pytest.mark.parametrize("n_proofs,split_max", [(2, None), (5, None), (16, None), (17, None), (2, 1), (5, 1), (16, 1)])
def test_emu_a_handful_of_requests_in_both_schedules(ectx, monkeypatch, n_proofs, split_max):
    """a call that fits one sub-batch fans its queries out over the streams like a single request (round 6: measured 10-45 % faster
    than two half-batches side by side / a one-sub-batch pipeline); the hooks build's OG_SPLIT_MAX=1 (round 5's rule) takes the
    symmetric lanes: the C restatement's bytes either way, and the plan says which schedule ran"""
    from owshen_amd import groth16 as g16
    if split_max is not None:
        monkeypatch.setenv("OG_SPLIT_MAX", str(split_max))
    want = "query fan-out" if split_max is None else "symmetric lanes"
    seen = {}
    orig = g16.ProvingKey.prove_batch

    def spy(self, wit, rs):
        seen["plan"] = self.plan(len(rs))[0]
        return orig(self, wit, rs)
    monkeypatch.setattr(g16.ProvingKey, "prove_batch", spy)
    cases.case_medium_circuit_vs_c_oracle(ectx, 60, n_proofs, None)
    assert seen.get("plan") == want, (seen, want)


@pytest.mark.parametrize("n_proofs,glv", [(1, "1"), (5, "1"), (3, "0")])
def test_emu_fan_out_assembles_in_one_part_or_two(ectx, monkeypatch, n_proofs, glv):
    """a fanned-out call forms the four products, A and the sum of C's products on a side stream beside the quotient and the H
    query, and C = L + H + that sum behind them (k_assemble_g1_early / _late, the default); OG_ASM_EARLY=0 is the one-part
    assembly behind the H query it replaced: the C restatement's bytes either way, with and without the GLV halves.  Likewise B's
    half: a wave per proof adding the 64 table entries of s delta2 as a tree (k_assemble_g2_tree), or OG_ASM_G2_TREE=0, a lane per
    proof adding them one after the other"""
    monkeypatch.setenv("OG_GLV", glv)
    monkeypatch.setenv("OG_ASM_EARLY", "0")
    monkeypatch.setenv("OG_ASM_G2_TREE", "0")
    cases.case_medium_circuit_vs_c_oracle(ectx, 60, n_proofs, None)
    monkeypatch.setenv("OG_ASM_EARLY", "1")
    monkeypatch.setenv("OG_ASM_G2_TREE", "1")
    cases.case_medium_circuit_vs_c_oracle(ectx, 60, n_proofs, None)


@pytest.mark.parametrize("n_proofs,early", [(1, "1"), (3, "0")])
def test_emu_assembly_products_by_a_wave_per_chain(ectx, monkeypatch, n_proofs, early):
    """k_assemble_g1_muls_w9: r delta, (r s) delta, s (alpha + A), r (beta + B1) as GLV halves, ONE WAVE per half-length chain with
    the group law in the wave-wide form (coordinates spread over nine lanes, strict products, differences through borrow-free
    multiples of the modulus) -- the default for calls of up to 8 requests; the interpreter's other cases take the lane per chain
    (tests/emu.py).  The C restatement's bytes, in the two-part and the one-part assembly"""
    monkeypatch.setenv("OG_ASM_W9_MAX", "8")
    monkeypatch.setenv("OG_ASM_EARLY", early)
    cases.case_medium_circuit_vs_c_oracle(ectx, 60, n_proofs, None)


def test_emu_explicit_sub_batch_plan(ectx, monkeypatch):
    """OG_SUB_PLAN: sizes above the sub-batch bound are clamped, the last size repeats, a short tail is allowed"""
    monkeypatch.setenv("OG_SUB_BATCH", "3")
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    monkeypatch.setenv("OG_SUB_PLAN", "1,7,2")
    cases.case_medium_circuit_vs_c_oracle(ectx, 60, 8, None)     # 1, 3, 2, 2


def test_emu_fixed_base_products_survive_a_scratch_release(ectx):
    """og_scalar_mul_d keeps the generator's fixed-base table for the life of the context -- outside the scratch arena, which
    og_release_scratch hands back: the products are the C restatement's before the release, after it, and after a proof call
    has regrown the arena; G1 and G2, scalars 0, 1, r - 1 among them"""
    import numpy as np
    from owshen_amd import groth16
    from oracle.c import binding as oc
    from oracle.py import fields
    rng = np.random.default_rng(77)
    for step in range(3):
        if step == 1:
            ectx.release_scratch()
        if step == 2:
            cases.case_medium_circuit_vs_c_oracle(ectx, 30, 2, None)
        ks = rng.integers(0, 256, (21, 32), dtype=np.uint8)
        ks[:, 31] &= 0x1F
        ks[0] = 0
        ks[1] = 0
        ks[1, 0] = 1
        ks[2] = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
        for group, gen, ref in ((1, groth16.G1_GEN_BYTES, oc.fixed_base_g1), (2, groth16.G2_GEN_BYTES, oc.fixed_base_g2)):
            got = ectx.to_host(ectx.scalar_mul(group, gen, ectx.to_device(ks)))
            assert got.tobytes() == ref(np.frombuffer(gen, dtype=np.uint8), ks).tobytes(), (step, group)


def test_emu_assembly_without_glv(ectx, monkeypatch):
    """the proof assembly's plain 254-bit scalar multiplications (what batches above 64 proofs use; by default the few proofs
    of an interpreter case take the GLV halves: glv.h, k_assemble_g1_muls_glv) give the same proofs"""
    monkeypatch.setenv("OG_GLV", "0")
    cases.case_prove_batch_matches_oracle_and_verifies(ectx)


@pytest.mark.parametrize("lanes", [2, 1])
def test_emu_l_and_h_queries_share_one_bucket_set(ectx, monkeypatch, lanes):
    """C = sum z L + sum h H as ONE multi-scalar multiplication (msm_run_phase: the L half accumulates and stops, the H half adds
    to the same buckets and is reduced once): same proofs as the C restatement -- which runs the two queries apart -- with the
    queries merged (3 G1 bucket reductions per sub-batch) and kept apart (OG_MERGE_LH=0: 4), in the stage pipeline and on one
    stream, and with the heavy-bucket path forced in BOTH halves (threshold 2 entries, a list of 3: it overflows, so some
    heavy buckets are accumulated inline and some are added in by k_heavy_combine)"""
    monkeypatch.setenv("OG_SUB_BATCH", "2")
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    ectx.set_lanes(lanes)
    try:
        regions = {}
        for merged in ("1", "0"):
            monkeypatch.setenv("OG_MERGE_LH", merged)
            for heavy in (None, ("2", "3"), ("1", "4096")):
                if heavy:
                    monkeypatch.setenv("OG_HEAVY", heavy[0])
                    monkeypatch.setenv("OG_HEAVY_CAP", heavy[1])
                else:
                    monkeypatch.delenv("OG_HEAVY", raising=False)
                    monkeypatch.delenv("OG_HEAVY_CAP", raising=False)
                ectx.profile(True)
                cases.case_medium_circuit_vs_c_oracle(ectx, 60, 5, None)     # (asserts byte equality with the C restatement)
                regions[merged] = ectx.profile_read()["reduce_g1"][1]
                ectx.profile(False)
        assert regions["1"] * 4 == regions["0"] * 3, regions
    finally:
        ectx.set_lanes(2)
