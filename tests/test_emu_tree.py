"""Incremental-tree batched append on the CPU interpreter (tests/hipemu); cases in tests/tree_cases.py."""
import pytest

from tests import tree_cases as cases


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


@pytest.mark.parametrize("depth,batches", [(3, [1, 1, 2, 3, 1]), (5, [7, 1, 8, 16]), (6, [64]), (32, [3, 5])])
def test_emu_append(ectx, depth, batches):
    cases.case_append_matches_incremental_tree(ectx, depth, batches, seed=depth)


def test_emu_append_rejects_overflow(ectx):
    import numpy as np
    from owshen_amd.api import OwshenGpuError
    f = ectx.to_device(np.zeros((2, 32), dtype=np.uint8))
    with pytest.raises(OwshenGpuError):
        ectx.mimc7_append(2, f, 3, ectx.to_device(np.zeros((2, 32), dtype=np.uint8)))


@pytest.mark.parametrize("depth,batches", [(3, [1, 1, 2, 1, 3]), (6, [1, 5, 1]), (32, [1, 2])])
def test_emu_host_append_equals_the_kernels(ectx, depth, batches):
    cases.case_host_append_equals_the_kernels(ectx, depth, batches, seed=40 + depth)


def test_emu_one_and_two_lanes_per_hash_agree(ectx, monkeypatch):
    cases.case_one_and_two_lanes_per_hash_agree(ectx, monkeypatch, n_hash=7, n_paths=3, depth=4, n_leaves=16, witness_depth=2)
