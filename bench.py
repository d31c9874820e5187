#!/usr/bin/env python3
"""bench.py -- withdraw proofs/sec on MI355X (BASELINE.json metric), one process per GPU.

A "step" proves one batch of withdraw statements end to end on the GPU: batched MiMC7 witness
generation -> R1CS products -> H-polynomial (7 NTTs) -> 5 Pippenger MSMs -> proof assembly.
Workload = BASELINE.json configs[1]: batch of 1024 proofs of the depth-32 MiMC7 Merkle withdraw
circuit sized to n_wires = 2^18 / NTT domain 2^17 (MSM ~2^20 G1 points per proof) with synthetic
padding gates; inputs (per-proof secrets, paths, blinding) are synthetic and resident in HBM when
the timed region starts; the proving key is generated in-process from fixed toxic waste.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--natural] [--no-cpu]

N > 1: launched by torch.distributed.run, one rank per GPU; proofs are independent units, so each
rank proves its own batch with a replicated key and there is no data-path collective (weak
scaling); time = max over ranks between barriers.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
TOXIC = (0x1F3A5C7E9B2D4F60718293A4B5C6D7E8F9, 0x2A4C6E8091B3D5F7, 0x3B5D7F91A3C5E7, 0x4C6E80A2C4E6, 0x5D7F91B3D5F7A9)
G1_POINT_BYTES, G2_POINT_BYTES = 96, 160   # algorithmic bytes per MSM point: affine base + 32 B scalar


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="proofs per step per GPU")
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--natural", action="store_true", help="the natural circuit (no padding gates): ~2^15 constraints")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    import numpy as np
    import torch
    import torch.distributed as dist

    device = local_rank % max(1, torch.cuda.device_count())  # one rank per GPU (modulo only matters for 1-GPU dry runs)
    torch.cuda.set_device(device)
    control = None
    if world > 1:
        # The path has NO data-path collective (proofs are independent); torch.distributed only carries the barriers and
        # the max-over-ranks of the elapsed time.  RCCL first; gloo if RCCL cannot initialise (e.g. a dry run with two
        # ranks on one GPU), so a control-plane hiccup never costs the measurement.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("OG_BENCH_BACKEND", "nccl")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device))
                t = torch.zeros(1, device="cuda")
                dist.all_reduce(t)  # fail here, not inside the timed region
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] rank {rank}: {backend} control plane failed ({type(e).__name__}: {e}); falling back to gloo")
            if dist.is_initialized():
                dist.destroy_process_group()
            backend = "gloo"
            dist.init_process_group("gloo")
        control = backend

    from owshen_amd import api, circuit, groth16

    t_setup = time.time()
    ctx = api.Context(device)
    depth = args.depth
    n_pad3, n_pad2 = (0, 0) if args.natural else circuit.baseline_shape(depth)
    r1cs = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
    blob, _vk = groth16.setup(ctx, r1cs, *TOXIC)
    pk = groth16.ProvingKey(ctx, blob)
    m, d = pk.n_wires, 1 << pk.log_d
    if rank == 0:
        log(f"[bench] circuit: n_wires={m} constraints={r1cs.n_constraints} domain=2^{pk.log_d} "
            f"nnz=({r1cs.a.nnz},{r1cs.b.nnz},{r1cs.c.nnz}) key={len(blob) / 1e6:.0f} MB setup={time.time() - t_setup:.1f}s")

    # synthetic inputs, resident in HBM: per-proof records (nullifier, secret, amount, recipient, pad_seed, index, siblings)
    B = args.batch
    rng = np.random.Generator(np.random.PCG64(20241008 + rank))
    inputs = rng.integers(0, 256, (B, 6 + depth, 32), dtype=np.uint8)
    inputs[:, :, 31] &= 0x1F                       # < 2^253 < r
    inputs[:, 5, 8:] = 0                           # index: u64
    if depth < 64:
        inputs[:, 5, :8] = (inputs[:, 5, :8].view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
    rs = rng.integers(0, 256, (B, 64), dtype=np.uint8)
    rs[:, 31] &= 0x1F
    rs[:, 63] &= 0x1F
    inputs_d = ctx.to_device(inputs)

    def step():  # input records -> witnesses (batched MiMC7 walk) -> proofs, all on the GPU
        return circuit.prove_from_inputs(ctx, pk, depth, inputs_d, rs, n_pad3, n_pad2)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile(True)
    fence()
    t0 = time.perf_counter()
    proofs = None
    for _ in range(args.steps):
        proofs = step()
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert proofs is not None and proofs.any(), "prover returned empty proofs"

    total_proofs = B * args.steps * world
    value = total_proofs / dt

    def roofline_of(prof, note):
        """dominant kernel = the G1 bucket-accumulation kernel (most VALU work on the path; one launch per timed
        region of kind 0); achieved = algorithmic bytes per launch / average launch duration (HIP events)"""
        kms, kn, kunits = prof["accumulate_g1"]
        achieved = (kunits * G1_POINT_BYTES / kn) / (kms / kn * 1e-3) / 1e9 if kn and kms > 0 else 0.0
        # HBM-side traffic from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE per MSM point,
        # profiles/pmc_traffic.json; same bench command at batch 28), scaled to this run's launch size
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = int(json.load(f)["k_accumulate_g1"]["bytes_per_point"] * kunits / kn) if kn else None
        except (OSError, KeyError, ValueError):
            pass
        return {"bound": "hbm", "kernel": "k_accumulate<Fq> (G1 bucket accumulation)", "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "launches": kn, "avg_launch_ms": round(kms / kn, 4) if kn else None,
                "algorithmic_bytes_per_launch": int(kunits * G1_POINT_BYTES / kn) if kn else 0, "note": note}

    roofline = roofline_of(prof, "timed region (two lanes: launches share the GPU with the other lane's kernels). Modular "
                           "big-integer path: bound by 32-bit integer-multiply VALU issue, not HBM (DESIGN.md 4.1, 5)")
    breakdown = {k: round(v[0] / args.steps, 3) for k, v in prof.items()}
    # one extra, untimed, strictly serial step: kernel durations free of co-scheduling (what rocprof --stats of a
    # single-lane run shows), for the isolated roofline figure and a stage breakdown that adds up
    ctx.set_lanes(1)
    ctx.profile(True)
    step()
    torch.cuda.synchronize()
    prof1 = ctx.profile_read()
    ctx.profile(False)
    ctx.set_lanes(2)
    roofline_isolated = roofline_of(prof1, "extra untimed single-lane step")
    breakdown_isolated = {k: round(v[0], 3) for k, v in prof1.items()}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        n_cpu = min(8, B)
        wit_d = circuit.witness(ctx, depth, inputs_d[:n_cpu], n_pad3, n_pad2)  # the sample's witnesses, for the CPU prover
        cpu = cpu_baseline(ctx, blob, wit_d, rs, proofs, args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "withdraw proofs/sec (batch=1024)", "value": round(value, 3), "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (9 x 29-bit-limb Montgomery, 254-bit modular integers)", "data": "synthetic",
            "config": {"workload": ("natural depth-%d withdraw circuit" % depth) if args.natural else
                       "BASELINE.json configs[1]: batch of 1024 withdraw proofs, depth-32 MiMC7 Merkle circuit sized to "
                       "n_wires=2^18 / NTT 2^17 (G1 MSM ~2^20 points + G2 MSM 2^18 per proof) with synthetic padding gates",
                       "batch_per_gpu": B, "n_wires": m, "domain": d, "merkle_depth": depth,
                       "parallelism": f"proofs sharded across {world} GPU(s), key replicated, no data-path collective"
                       + (f" (barriers / max-time over {control})" if control else "")},
            "roofline": roofline,
            "roofline_isolated": roofline_isolated,
            "cpu_baseline": cpu,
            "stage_ms_per_step": breakdown,
            "stage_ms_per_step_isolated": breakdown_isolated,
            "algorithmic_MB_per_proof": round((3 * m * G1_POINT_BYTES + d * G1_POINT_BYTES + m * G2_POINT_BYTES + 7 * d * 64) / 1e6, 1),
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(ctx, blob, wit_d, rs, gpu_proofs, budget_s):
    """The C restatement of the prover (oracle/c, TEST INFRASTRUCTURE) timed on this host's cores over a bounded
    sample of the same batch; doubles as an end-of-run parity check at full size."""
    import numpy as np
    from oracle.c import binding as oc
    ck = oc.prepared_key_from_blob(blob)
    done, t_total = 0, 0.0
    while done < min(8, wit_d.shape[0]) and t_total < budget_s:
        w = ctx.to_host(wit_d[done])
        r = int.from_bytes(rs[done, :32].tobytes(), "little")
        s = int.from_bytes(rs[done, 32:].tobytes(), "little")
        t0 = time.perf_counter()
        p = ck.prove(w, r, s)
        t_total += time.perf_counter() - t0
        assert p == gpu_proofs[done].tobytes(), f"GPU proof {done} differs from the CPU restatement"
        done += 1
    return {"value": round(done / t_total, 4), "unit": "proofs/s", "cores": oc.prove_threads(),
            "kind": "port", "sample": f"{done} proof(s) of the same batch ({t_total:.1f} s), byte-identical to the GPU proofs; "
            "own C restatement -- the reference has no prover (SURVEY.md 0.1)", "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
