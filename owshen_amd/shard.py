"""Multi-GPU sharding of the prover path: one process per GPU, `torch.distributed` (RCCL on GPUs).

Two shardings exist on this path:

* **Proofs** are independent units: rank g proves the slice `partition(n, world, g)` of a batch
  with a replicated key.  There is no data-path collective; `gather_proofs` is a convenience
  all-gather of the 256-byte results.
* **One large MSM** (the 2^26 micro-benchmark shape) shards by points: every rank owns a slice of
  the bases and scalars, computes its partial sum, and the partial POINTS are all-gathered
  (`world x 64|128` bytes -- RCCL cannot add curve points, so this is an all-gather followed by a
  local group-law sum, never an all-reduce) and summed on every rank.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import api


def partition(n, world, rank):
    """contiguous balanced slice [lo, hi) of n items for `rank` of `world`"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _all_gather_bytes(local, group=None):
    """local: np.uint8 [k] -> np.uint8 [world, k] on every rank (device tensors under RCCL, host under gloo)"""
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint8).reshape(-1).copy())
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    out = torch.empty(world * t.shape[0], dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.cpu().numpy().reshape((world,) + tuple(np.shape(local)))


def gather_proofs(local_proofs, counts, group=None):
    """local_proofs: np.uint8 [n_local, 256]; counts: proofs per rank -> np.uint8 [sum(counts), 256] on every rank"""
    world = dist.get_world_size(group)
    width = max(counts)
    pad = np.zeros((width, 256), dtype=np.uint8)
    pad[: local_proofs.shape[0]] = local_proofs
    allp = _all_gather_bytes(pad.reshape(-1), group).reshape(world, width, 256)
    return np.concatenate([allp[r, : counts[r]] for r in range(world)])


def msm_point_sharded(ctx, group_id, points_local, scalars_local, group=None, window_bits=0):
    """points_local / scalars_local: this rank's slice (device buffers, canonical).  Returns the full MSM
    (np.uint8 [64 | 128], canonical affine) on every rank."""
    pb = 64 if group_id == 1 else 128
    if points_local.shape[0]:
        part = api.Bases(ctx, group_id, points_local, window_bits, False).msm(scalars_local)[0]
    else:
        part = np.zeros(pb, dtype=np.uint8)
    if group is None and not dist.is_initialized():
        return part
    parts = _all_gather_bytes(part, group)  # [world, pb]: partial points, one per rank
    ones = np.zeros((parts.shape[0], 32), dtype=np.uint8)
    ones[:, 0] = 1
    return api.Bases(ctx, group_id, ctx.to_device(parts), 8, False).msm(ctx.to_device(ones))[0]


def msm_window_sharded(bases, scalars, group=None):
    """Window-sharded MSM (BASELINE.json configs[3]): `bases` (api.Bases) replicated on every rank and
    `scalars` (device uint8 [n,32]) identical on every rank (broadcast by the caller); rank g accumulates the windows
    k = g (mod world); the per-window points (<= 16 x 256 B per rank) are all-gathered and every rank runs the Horner
    combine.  Returns the MSM (np.uint8 [64 | 128]) on every rank."""
    if group is None and not dist.is_initialized():
        return bases.msm_combine(bases.msm_windows(scalars, 0, 1), 1)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    part = bases.msm_windows(scalars, rank, world)
    ctx = bases.ctx
    if dist.get_backend(group) == "nccl":      # device-to-device over RCCL / xGMI
        gathered = ctx.empty(world * part.shape[0])
        dist.all_gather_into_tensor(gathered, part, group=group)
    else:                                      # gloo: host round trip (CPU tests)
        gathered = ctx.to_device(_all_gather_bytes(ctx.to_host(part), group).reshape(-1))
    return bases.msm_combine(gathered, world)


def prove_window_sharded(ctx, pk, rs, *, inputs_d=None, depth=None, n_pad3=0, n_pad2=0, witnesses_d=None, group=None, return_public=False):
    """Window-sharded PROVING, one process per GPU (BASELINE.json north_star / configs[3]; include/owshen_gpu.h
    og_withdraw_prove_partials_d + og_prove_from_partials_d).  Every rank holds the SAME inputs (withdraw records `inputs_d`
    with `depth`, or witnesses `witnesses_d`; device buffers) and the same blinding `rs`; rank g walks the witnesses and the
    quotient itself and accumulates the windows k = g (mod world) of the five queries over its copy of the key; the partial
    points (768 B per proof and rank) are all-gathered -- RCCL device-to-device over xGMI, never an all-reduce: curve points do
    not add limb-wise -- and every rank adds the shares and assembles (all ranks return the same n x 256 bytes, byte-identical
    to pk.prove_batch_device / circuit.prove_from_inputs on one GPU)."""
    from . import circuit
    solo = group is None and not dist.is_initialized()
    world, rank = (1, 0) if solo else (dist.get_world_size(group), dist.get_rank(group))
    pub = None
    if inputs_d is not None:
        res = circuit.partials_from_inputs(ctx, pk, depth, inputs_d, rank, world, n_pad3, n_pad2, return_public=return_public)
        part, pub = res if return_public else (res, None)
    else:
        part = pk.prove_partials_device(witnesses_d, rank, world)
    if solo or world == 1:
        gathered = part
    elif dist.get_backend(group) == "nccl":      # device-to-device over RCCL / xGMI
        gathered = ctx.empty(world * part.shape[0])
        dist.all_gather_into_tensor(gathered, part, group=group)
    else:                                        # gloo: host round trip (CPU tests)
        gathered = ctx.to_device(_all_gather_bytes(ctx.to_host(part), group).reshape(-1))
    proofs = pk.prove_from_partials(gathered, world, rs)
    return (proofs, pub) if return_public else proofs


def broadcast_bytes(ctx, buf, src=0, group=None):
    """device uint8 buffer broadcast from rank `src` (RCCL broadcast on GPUs; host round trip under gloo)"""
    if group is None and not dist.is_initialized():
        return buf
    if dist.get_backend(group) == "nccl":
        dist.broadcast(buf, src, group=group)
        return buf
    t = torch.from_numpy(np.ascontiguousarray(ctx.to_host(buf)).copy())
    dist.broadcast(t, src, group=group)
    return ctx.to_device(t.numpy())


def tree_build_sharded(ctx, leaves_local, group=None):
    """MiMC7 Merkle root over `world x n_local` leaves (BASELINE.json configs[4]: 2^20 leaves on 8 GPUs).
    Rank g owns the contiguous slice g of the leaves (n_local a power of two, world a power of two): it builds
    its subtree on its GPU, the `world` subtree roots (32 B each) are all-gathered, and every rank hashes the
    top log2(world) levels redundantly.  Returns (root: 32 bytes, local subtree nodes: device buffer)."""
    nodes = ctx.mimc7_tree_build(leaves_local)
    sub_root = ctx.to_host(nodes[-1:]).reshape(32)
    if group is None and not dist.is_initialized():
        return sub_root.tobytes(), nodes
    roots = _all_gather_bytes(sub_root, group)  # [world, 32]
    world = roots.shape[0]
    assert world & (world - 1) == 0, "world size must be a power of two"
    if world == 1:
        return roots[0].tobytes(), nodes
    top = ctx.mimc7_tree_build(ctx.to_device(roots))
    return ctx.to_host(top[-1:]).reshape(32).tobytes(), nodes
