"""Withdraw circuit + witness generator on the CPU interpreter (tests/hipemu); cases in tests/withdraw_cases.py."""
import pytest

from tests import withdraw_cases as cases


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


@pytest.mark.parametrize("depth,n_pad3,n_pad2", [(1, 0, 0), (3, 7, 130), (2, 0, 64), (32, 0, 0)])
def test_emu_r1cs_and_witness_match_spec(ectx, depth, n_pad3, n_pad2):
    cases.case_r1cs_and_witness_match_spec(ectx, depth, n_pad3, n_pad2, n_proofs=2 if depth == 32 else 3)


@pytest.mark.parametrize("depth,n_pad3,n_pad2", [(1, 0, 0), (3, 7, 130), (32, 0, 0)])
def test_emu_witness_lane_local_forms_match_spec(ectx, monkeypatch, depth, n_pad3, n_pad2):
    """OG_WITNESS_W9=0: the wave-per-proof kernel (k_withdraw_core_lat) a handful of requests took until round 6 -- still what a hooks
    build can select, and its lane-pair sibling is what calls of more than 512 requests run"""
    monkeypatch.setenv("OG_WITNESS_W9", "0")
    cases.case_r1cs_and_witness_match_spec(ectx, depth, n_pad3, n_pad2, n_proofs=2 if depth == 32 else 3)


@pytest.mark.parametrize("form,depth,n_pad3,n_pad2", [("0", 1, 0, 0), ("0", 3, 7, 130), ("1", 3, 7, 130)])
def test_emu_witness_w9_earlier_forms_match_spec(ectx, monkeypatch, form, depth, n_pad3, n_pad2):
    """OG_W9_ROWS=0: the wave-wide walk with rounds four products deep in ONE row of the wave; 1: three deep over two rows -- t^4
    beside t^3, t^6 beside t^7 -- with the 29-bit Montgomery digit; the launches use 2, the same with the 32-bit digit.  Hooks builds only"""
    monkeypatch.setenv("OG_W9_ROWS", form)
    cases.case_r1cs_and_witness_match_spec(ectx, depth, n_pad3, n_pad2, n_proofs=3)


def test_emu_witness_w9_bound_is_per_call(ectx, monkeypatch):
    """OG_WITNESS_W9_MAX: 3 proofs with the bound at 2 take the lane-local kernels, at 3 the wave-wide ones -- same wires"""
    monkeypatch.setenv("OG_WITNESS_W9_MAX", "2")
    cases.case_r1cs_and_witness_match_spec(ectx, 2, 0, 64)
    monkeypatch.setenv("OG_WITNESS_W9_MAX", "3")
    cases.case_r1cs_and_witness_match_spec(ectx, 2, 0, 64)


def test_emu_witness_two_lanes_per_proof_form(ectx, monkeypatch):
    """the throughput form of the witness walk (k_withdraw_core<true>: two lanes per proof, what a sub-batch of hundreds uses)
    forced at toy size -- by default a handful of proofs take the wave-per-proof form (k_withdraw_core_lat) -- and the bound
    between the two (OG_WITNESS_LAT_MAX): 3 proofs with the bound at 2 take the old kernel, at 3 the new one"""
    monkeypatch.setenv("OG_WITNESS_W9", "0")   # (off the wave-wide form, the default for calls this size since round 6)
    monkeypatch.setenv("OG_WITNESS_LAT", "0")
    cases.case_r1cs_and_witness_match_spec(ectx, 3, 7, 130)
    monkeypatch.delenv("OG_WITNESS_LAT")
    monkeypatch.setenv("OG_WITNESS_LAT_MAX", "2")
    cases.case_r1cs_and_witness_match_spec(ectx, 2, 0, 64)


@pytest.mark.parametrize("depth,n_pad3,n_pad2,prove", [(2, 2, 3, True), (3, 7, 130, False), (32, 0, 0, False)])
def test_emu_host_chains_gives_the_kernels_bytes(ectx, depth, n_pad3, n_pad2, prove):
    cases.case_host_chains_gives_the_kernels_bytes(ectx, depth, n_pad3, n_pad2, n_proofs=2 if depth == 32 else 3, prove=prove)


def test_emu_withdraw_end_to_end(ectx):
    cases.case_withdraw_end_to_end(ectx, 1, 2, 3)     # (the shape of test_emu_submitted_batches: one key, cases.py _key)


def test_emu_dense_rows(ectx):
    cases.case_dense_rows(ectx, 1, 2, 5)


def test_emu_withdraw_end_to_end_dense(ectx):
    cases.case_withdraw_end_to_end(ectx, 1, 2, 5, dense=True)


def test_emu_dense_stage_pipeline_shares_one_digit_sort(ectx, monkeypatch):
    """dense padding: the A, B and L queries hold the same wire list (L: the 7 public wires as points at infinity), so the
    stage pipeline sorts the witness digits once per sub-batch for all three (groth16.hip, sort_src) -- forced at toy size"""
    monkeypatch.setenv("OG_SUB_BATCH", "1")
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    cases.case_withdraw_end_to_end(ectx, 1, 2, 5, dense=True)


def test_emu_serial_path_reuses_shared_sorts(ectx):
    """one stream, sub-batches one after the other (og_set_lanes(1): the isolated-stage mode of bench.py): the A query's sorted
    entries serve every query on A's wire list -- all three with the dense padding; L but not B without it (A, L, then B)"""
    ectx.set_lanes(1)
    try:
        cases.case_withdraw_end_to_end(ectx, 1, 2, 5, dense=True)
        cases.case_withdraw_end_to_end(ectx, 1, 2, 3)
    finally:
        ectx.set_lanes(2)


@pytest.mark.parametrize("depth,n_pad3,n_pad2,dense", [(1, 0, 0, False), (2, 5, 130, True), (3, 0, 64, False)])
def test_emu_native_builder(ectx, depth, n_pad3, n_pad2, dense):
    cases.case_native_builder_equals_python_builder(ectx, depth, n_pad3, n_pad2, dense)


def test_emu_submitted_batches(ectx, monkeypatch):
    """the enqueue / finish split of prove_batch on the interpreter, stage pipeline forced at toy size (the witness-inside-the-
    pipeline path needs sub-batch x wires >= 2^26 on hardware: here the small-circuit path, which completes inside submit)"""
    monkeypatch.setenv("OG_SUB_BATCH", "2")
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    cases.case_submitted_batches_equal_blocking_calls(ectx, 1, 2, 3, [1])
    # ... and the witness-inside-the-pipeline path (OG_GEN_MIN lowers its threshold): calls really stay enqueued, two call
    # slots alternate, the scratch-slot counter runs on across calls (2 + 1 + 2 proofs = sub-batches 1,1 | 1 | 1,1)
    monkeypatch.setenv("OG_GEN_MIN", "1")
    cases.case_submitted_batches_equal_blocking_calls(ectx, 1, 2, 3, [2, 1, 2], third_is_refused=True)


def test_emu_malformed_records_are_rejected(ectx, monkeypatch):
    """boundary check of the input records: whole-slab path (small circuits), then the witness-inside-the-pipeline path"""
    cases.case_malformed_records_are_rejected(ectx, 1, 2, 3, quick=True)
    monkeypatch.setenv("OG_SUB_BATCH", "2")
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    monkeypatch.setenv("OG_GEN_MIN", "1")
    cases.case_malformed_records_are_rejected(ectx, 1, 2, 3, pipeline=True, quick=True)


def test_emu_jobs_are_consumed_once(ectx, monkeypatch):
    """og_job_wait / og_job_abandon handle validation and call-slot selection on jobs that really stay enqueued (OG_GEN_MIN=1;
    the completes-inside-submit jobs of small circuits run in the GPU suite)"""
    monkeypatch.setenv("OG_SUB_BATCH", "2")
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    monkeypatch.setenv("OG_GEN_MIN", "1")
    cases.case_jobs_are_consumed_once(ectx, 1, 2, 3, stays_enqueued=True, n_proofs=1)


def test_emu_too_many_witnesses_in_one_call_is_named_not_a_launch_error(ectx):
    """og_withdraw_witness_d with more than 65535 records used to reach the record-check kernel first (the record
    index is grid.y) and come back as an opaque launch failure after growing its scratch; now the documented OG_ERR_INVALID
    arrives before anything is launched or allocated"""
    import ctypes as C
    import numpy as np
    lib = ectx._lib
    before = ectx.mem_info()["scratch_bytes"]
    dummy = np.zeros(64, dtype=np.uint8)
    rc = lib.og_withdraw_witness_d(ectx._h, 2, 0, 0, dummy.ctypes.data_as(C.c_void_p), 65536, dummy.ctypes.data_as(C.c_void_p))
    assert rc == -1 and b"at most 65535" in lib.og_last_error()
    assert ectx.mem_info()["scratch_bytes"] == before
