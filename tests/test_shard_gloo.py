"""The N > 1 path on CPU: world-size-2 gloo, one process per rank (the GPU-side compute is played by the CPU
interpreter build of the kernels, tests/hipemu): proof sharding with a replicated key, the window-sharded MSM
(broadcast of the scalars, all-gather of the per-window points, Horner combine), the point-sharded MSM with its
all-gather of partial points, and window-sharded PROVING (partial points of the five queries all-gathered, every rank assembles)."""
import os
import random
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import emu
        from tests.r1cs_util import random_r1cs
        from owshen_amd import shard, groth16 as g16
        from oracle.py import fields, groth16 as og16
        from oracle.py.curve import G1, G1_GEN, g1_to_bytes
        from oracle.c import binding as oc
        ctx = emu.Ctx()
        # --- point-sharded MSM: 300 points split 150/150, partial points all-gathered and summed
        n = 301
        rng = np.random.default_rng(3)
        ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        ks[:, 31] &= 0x1F
        sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        sc[:, 31] &= 0x1F
        bases = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
        lo, hi = shard.partition(n, world, rank)
        got = shard.msm_point_sharded(ctx, 1, ctx.to_device(bases[lo:hi]), ctx.to_device(sc[lo:hi]))
        assert got.tobytes() == oc.msm_g1(bases, sc).tobytes()
        # --- window-sharded MSM (BASELINE.json configs[3], the sharding north_star names): bases replicated, scalars broadcast
        # from rank 0, rank g accumulates the windows k = g (mod world), the per-window points all-gathered, Horner combine
        # on every rank -- plain bases (one bucket set per window) and precomputed window tables, G1 and G2
        from owshen_amd import api
        from oracle.py.curve import G2_GEN, g2_to_bytes
        for group_id, gen_bytes, fixed, msm_ref, window, precomp in (
                (1, g1_to_bytes(G1_GEN), oc.fixed_base_g1, oc.msm_g1, 8, False), (1, g1_to_bytes(G1_GEN), oc.fixed_base_g1, oc.msm_g1, 12, True),
                (2, g2_to_bytes(G2_GEN), oc.fixed_base_g2, oc.msm_g2, 8, False)):
            nw = 97
            pts = fixed(np.frombuffer(gen_bytes, dtype=np.uint8), ks[:nw])
            b = api.Bases(ctx, group_id, ctx.to_device(pts), window, precomp)
            sc_w = sc[:nw].copy()
            sc_w[:2] = 0
            sc_w[1, 0] = 1
            mine = ctx.to_device(sc_w if rank == 0 else np.zeros_like(sc_w))   # only rank 0 holds the scalars ...
            mine = shard.broadcast_bytes(ctx, mine, src=0)                       # ... until the broadcast
            got = shard.msm_window_sharded(b, mine)
            assert got.tobytes() == msm_ref(pts, sc_w).tobytes(), (group_id, window, precomp)
            b.close()
        # --- sharded Merkle tree: each rank builds the subtree of its half of 64 leaves, roots all-gathered
        from oracle.py import mimc7
        leaves = [rng.integers(0, 256, 32, dtype=np.uint8) for _ in range(64)]
        for lf in leaves:
            lf[31] &= 0x1F
        lv = np.stack(leaves)
        lo, hi = shard.partition(64, world, rank)
        root, _nodes = shard.tree_build_sharded(ctx, ctx.to_device(lv[lo:hi]))
        want = mimc7.tree_build([int.from_bytes(x.tobytes(), "little") for x in leaves])[-1][0]
        assert int.from_bytes(root, "little") == want
        # --- proof sharding: 5 proofs over 2 ranks (3 + 2), same key on both ranks, results gathered
        n_wires, cons, z0 = random_r1cs(12, 1, seed=9)
        blob, _vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, 1, cons), 3, 5, 7, 11, 13)
        pk = g16.ProvingKey(ctx, blob)
        rnd = random.Random(4)
        total = 5
        zs, rs = [], []
        for t in range(total):
            z = list(z0)
            r2 = random.Random(50 + t)
            for i in range(1, n_wires - len(cons)):
                z[i] = r2.randrange(fields.R)
            for k, (a, b, c) in enumerate(cons):
                av = sum(v * z[i] for i, v in a.items()) % fields.R
                bv = sum(v * z[i] for i, v in b.items()) % fields.R
                z[n_wires - len(cons) + k] = av * bv % fields.R
            zs.append(np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(-1, 32))
            rs.append((rnd.randrange(fields.R), rnd.randrange(fields.R)))
        counts = [shard.partition(total, world, r)[1] - shard.partition(total, world, r)[0] for r in range(world)]
        lo, hi = shard.partition(total, world, rank)
        local = pk.prove_batch(np.stack(zs[lo:hi]), rs[lo:hi])
        allp = shard.gather_proofs(local, counts)
        ck = oc.prepared_key_from_blob(blob)
        for t in range(total):
            assert allp[t].tobytes() == ck.prove(zs[t], *rs[t])
        # --- window-sharded PROVING (round 6; BASELINE.json north_star / configs[3]): both ranks hold all 5 witnesses and the
        # blinding, rank g accumulates the windows k = g (mod 2) of the five queries, the partial points (768 B per proof and
        # rank) are all-gathered, every rank assembles: the C restatement's bytes on BOTH ranks; then the withdraw form from
        # input records (witnesses generated on every rank), its public inputs, and a refused record on both ranks
        from owshen_amd import circuit
        got = shard.prove_window_sharded(ctx, pk, rs, witnesses_d=ctx.to_device(np.stack(zs)))
        for t in range(total):
            assert got[t].tobytes() == ck.prove(zs[t], *rs[t]), ("window-sharded proof", t)
        depth, n_pad3, n_pad2 = 1, 2, 3
        r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
        wblob, _wvk = g16.setup(ctx, r1, 21, 22, 23, 24, 25)
        wpk = g16.ProvingKey(ctx, wblob)
        rnd2 = random.Random(2)
        recs = np.stack([circuit.pack_inputs(rnd2.randrange(fields.R), rnd2.randrange(fields.R), 5, 6, rnd2.randrange(fields.R), rnd2.randrange(2),
                                             [rnd2.randrange(fields.R)], token=rnd2.randrange(1 << 160), chain_id=1387) for _ in range(3)])
        wrs = [(rnd2.randrange(fields.R), rnd2.randrange(fields.R)) for _ in range(3)]
        wit = circuit.witness(ctx, depth, ctx.to_device(recs), n_pad3, n_pad2)
        wck = oc.prepared_key_from_blob(wblob)
        wgot, wpub = shard.prove_window_sharded(ctx, wpk, wrs, inputs_d=ctx.to_device(recs), depth=depth, n_pad3=n_pad3, n_pad2=n_pad2,
                                                return_public=True)
        for t in range(3):
            assert wgot[t].tobytes() == wck.prove(wit[t], *wrs[t]), ("window-sharded withdraw proof", t)
        assert wpub.tobytes() == np.ascontiguousarray(wit[:, 1:7]).tobytes()
        bad = recs.copy()
        bad[1, 0] = np.frombuffer(fields.R.to_bytes(32, "little"), dtype=np.uint8)
        try:
            shard.prove_window_sharded(ctx, wpk, wrs, inputs_d=ctx.to_device(bad), depth=depth, n_pad3=n_pad3, n_pad2=n_pad2)
            raise AssertionError("a malformed record was proved")
        except api.OwshenGpuError as e:
            assert "input record 1: field 0" in str(e)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_partition_is_balanced_and_covers():
    from owshen_amd import shard
    for n in (0, 1, 5, 1024, 4097):
        for world in (1, 2, 3, 8):
            parts = [shard.partition(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo_sharding():
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
