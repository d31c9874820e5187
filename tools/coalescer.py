#!/usr/bin/env python3
"""Latency against offered load for the request coalescer (tools/coalescer.c, the drain rule of ffi/burn_proof.patch's
`prover::coalescer`): Poisson arrivals replayed through the C ABI on one GPU, one request per call against coalesced batches
(blocking, and with one call kept ahead), on the natural depth-32 statement (what `withdraw_handler` would prove) and on the
2^18-wire benchmark shape.  Writes gpurun_out/coalescer.json (copied to profiles/r05_coalescer.json).

    python tools/coalescer.py [--quick] [--shapes natural,baseline]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "tools", "_build", "coalescer")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "owshen_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "coalescer.c"), "-o", EXE, "-L", lib_dir, "-lowshen_gpu", "-lm",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def run(shape, mode, max_batch, window_us, duration, rates, seed=1):
    depth, p3, p2, dense = shape
    cmd = [EXE, str(depth), str(p3), str(p2), str(dense), str(mode), str(max_batch), str(window_us), str(duration), str(seed)] + [str(r) for r in rates]
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    if out.returncode != 0:
        raise RuntimeError(f"coalescer failed ({out.returncode}): {out.stderr[-800:]}")
    rows = [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]
    print(f"  mode {mode} N={max_batch} T={window_us}us: {len(rows)} rates in {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    for r in rows:
        print(f"    {r['offered_per_s']:7.0f}/s offered -> {r['achieved_per_s']:8.1f}/s, batch {r['mean_batch']:7.1f} (max {r['largest_batch']:4d}), "
              f"p50 {r['latency_ms']['p50']:8.2f} p99 {r['latency_ms']['p99']:8.2f} ms", file=sys.stderr, flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shapes", default="natural,baseline")
    ap.add_argument("--host-chains", type=int, default=0, help="og_set_host_chains(ctx, k) in the replayed node (0 = off)")
    args = ap.parse_args()
    if args.host_chains:
        os.environ["COALESCER_HOST_CHAINS"] = str(args.host_chains)
    from owshen_amd import circuit
    build()
    depth = 32
    out = {"what": "open-loop Poisson arrivals replayed through include/owshen_gpu.h on one MI355X (tools/coalescer.c): latency of a request = "
                   "arrival -> the proofs of its batch are on the host.  mode 0 = one request per blocking call (the handler without a "
                   "coalescer), 1 = coalesced (drain up to max_batch, or whoever arrived window_us after the first), one blocking call per "
                   "batch, 2 = the same with one call kept ahead (submit / poll / wait).  A small circuit's submit proves synchronously "
                   "(include/owshen_gpu.h), so mode 2 is run on the 2^18-wire shape only.",
           "shapes": {}}
    dur = 2.0 if args.quick else 4.0
    if "natural" in args.shapes:
        shape = (depth, 0, 0, 0)
        print("natural depth-32 statement (26 385 wires)", file=sys.stderr, flush=True)
        rows = []
        rows += run(shape, 0, 1, 0, dur, [20, 50, 80] if args.quick else [10, 20, 40, 60, 80, 100, 120])
        rows += run(shape, 1, 1024, 0, dur, [100, 1000, 3000] if args.quick else [20, 50, 100, 200, 500, 1000, 2000, 3000, 3500, 4000])
        if not args.quick:
            rows += run(shape, 1, 1024, 2000, dur, [50, 200, 1000, 3000])
            rows += run(shape, 1, 64, 0, dur, [200, 1000, 2000])
        out["shapes"]["natural"] = {"n_wires": circuit.shape(depth, 0, 0)[0], "rows": rows}
    if "baseline" in args.shapes:
        p3, p2 = circuit.baseline_shape(depth, dense=True)
        shape = (depth, p3, p2, 1)
        print("2^18-wire benchmark shape (dense padding)", file=sys.stderr, flush=True)
        dur2 = 3.0 if args.quick else 6.0
        rows = []
        rows += run(shape, 0, 1, 0, dur2, [20, 60] if args.quick else [10, 30, 50, 70])
        rows += run(shape, 1, 1024, 0, dur2, [100, 500] if args.quick else [20, 50, 100, 200, 300, 400, 500, 540])
        rows += run(shape, 2, 1024, 0, dur2, [100, 500] if args.quick else [100, 300, 500, 540, 570])
        out["shapes"]["baseline_2^18"] = {"n_wires": 1 << 18, "rows": rows}
    out["host_chains"] = args.host_chains
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "coalescer_host_chains.json" if args.host_chains else "coalescer.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote gpurun_out/coalescer" + ("_host_chains" if args.host_chains else "") + ".json", file=sys.stderr)


if __name__ == "__main__":
    main()
