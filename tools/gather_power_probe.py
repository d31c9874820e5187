#!/usr/bin/env python3
"""Gather-power probe (round 6): what do the ~1.4 TB/s of random 64-byte window-table gathers of the bucket
accumulation cost through the package power cap?

The headline step (dense padding, batch 1024) runs on the hooks build of the library with OG_GATHER_MASK folding every table
index onto a slice that stays cache resident -- the same instruction stream and VALU work, no HBM gathers, wrong results (so
nothing is verified) -- against the unmasked arm (mask 2^32 - 1: the same extra s_load / v_and), interleaved on ONE box, while
bench.py's GpuTelemetry samples the engine clock and socket power of the timed steps.

  arms:  full   OG_GATHER_MASK unset       the real tables (1.3 GB of G1 tables per query + G2: HBM gathers)
         mall   OG_GATHER_MASK=0x1ffff     2^17 entries: 8 MB (G1) / 16 MB (G2) -- Infinity-Cache resident, not L2 resident
         l2     OG_GATHER_MASK=0x3fff      2^14 entries: 1 MB / 2 MB -- L2 resident

usage (on the GPU box): python tools/gather_power_probe.py [--rounds 2] [--steps 3]   -> gpurun_out/r06_gather_power_probe.json"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMS = [("full", None), ("mall", 0x1FFFF), ("l2", 0x3FFF)]


def run_arm(name, mask, steps, extra):
    env = dict(os.environ)
    env["OWSHEN_GPU_LIB"] = os.path.join(ROOT, "owshen_amd", "libowshen_gpu_hooks.so")
    env.pop("OG_GATHER_MASK", None)
    if mask is not None:
        env["OG_GATHER_MASK"] = str(mask)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", "--no-cpu", "--no-legs", "--dense",
           "--no-verify"] + extra
    t0 = time.time()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    row = {"arm": name, "mask": mask, "rc": p.returncode, "wall_s": round(time.time() - t0, 1)}
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        row["error"] = f"{type(e).__name__}: {e}"
        row["stderr_tail"] = p.stderr[-1500:]
        return row
    box = d.get("box", {})
    row.update({
        "proofs_per_s": d["value"], "ms_per_step": d["ms_per_step"], "step_ms": d.get("step_ms"),
        "sclk_MHz": box.get("sclk_MHz"), "socket_power_W": box.get("socket_power_W"), "temp_C": box.get("temp_C"),
        "stage_ms_per_step": d.get("stage_ms_per_step"), "stage_ms_per_step_isolated": d.get("stage_ms_per_step_isolated"),
        "g1_launch_ms": d.get("roofline", {}).get("avg_launch_ms"),
    })
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    extra = ["--batch", str(args.batch)]
    rows = []
    for r in range(args.rounds):
        for name, mask in ARMS:
            row = run_arm(name, mask, args.steps, extra)
            row["round"] = r
            rows.append(row)
            print(json.dumps({k: row.get(k) for k in ("arm", "round", "rc", "proofs_per_s", "ms_per_step", "sclk_MHz", "socket_power_W", "error")}), flush=True)
    out = {"what": __doc__.split("\n\n")[0], "command": " ".join(sys.argv), "rows": rows}
    # per-arm means
    summ = {}
    for name, _ in ARMS:
        ok = [x for x in rows if x["arm"] == name and "proofs_per_s" in x]
        if not ok:
            continue
        def mean(f):
            v = [f(x) for x in ok if f(x) is not None]
            return round(sum(v) / len(v), 2) if v else None
        summ[name] = {
            "runs": len(ok), "proofs_per_s": mean(lambda x: x["proofs_per_s"]), "ms_per_step": mean(lambda x: x["ms_per_step"]),
            "sclk_MHz": mean(lambda x: (x.get("sclk_MHz") or {}).get("mean")), "socket_power_W": mean(lambda x: (x.get("socket_power_W") or {}).get("mean")),
            "accumulate_g1_isolated_ms": mean(lambda x: (x.get("stage_ms_per_step_isolated") or {}).get("accumulate_g1")),
            "accumulate_g2_isolated_ms": mean(lambda x: (x.get("stage_ms_per_step_isolated") or {}).get("accumulate_g2")),
            "accumulate_g1_ms": mean(lambda x: (x.get("stage_ms_per_step") or {}).get("accumulate_g1")),
            "accumulate_g2_ms": mean(lambda x: (x.get("stage_ms_per_step") or {}).get("accumulate_g2")),
        }
    if "full" in summ:
        for name in summ:
            if name != "full" and summ[name]["proofs_per_s"] and summ["full"]["proofs_per_s"]:
                summ[name]["speedup_vs_full"] = round(summ[name]["proofs_per_s"] / summ["full"]["proofs_per_s"], 4)
    out["summary"] = summ
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_gather_power_probe.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
