#!/usr/bin/env python3
"""What does a short kernel on a second stream get done beside a persistent kernel?  (og_ubench_coresidency)
Sweeps the resident kernel's workgroups per CU and VALU activity against the filler's workgroup size, LDS and wave priority;
writes gpurun_out/coresidency_probe.json."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from owshen_amd import api
    ctx = api.Context(0)
    out = (C.c_float * 3)()
    rows = []

    def run(w, kind, iters, fb, ft, lds, prio, work, delay=2000):
        ctx._check(ctx._lib.og_ubench_coresidency(ctx._h, w, kind, iters, fb, ft, lds, prio, work, delay, out))
        r = {"resident_wgs_per_cu": w, "resident": "mad chain" if kind == 0 else "sleep", "filler_blocks": fb, "filler_threads": ft,
             "filler_lds": lds, "filler_prio3": bool(prio), "filler_work": work, "resident_ms": round(out[0], 3),
             "filler_beside_ms": round(out[1], 3), "filler_alone_ms": round(out[2], 3)}
        rows.append(r)
        print(json.dumps(r), flush=True)

    # calibrate: iterations for a ~40 ms resident kernel (16 mads per iteration, ~9.5 cycles each when alone on its SIMD)
    busy_iters, sleep_iters = 400000, 120000
    for kind, iters in ((0, busy_iters), (1, sleep_iters)):
        for w in (4, 8, 12, 16):
            for ft, fb in ((256, 24000), (64, 96000), (1024, 240)):
                for prio in (0, 1):
                    run(w, kind, iters, fb, ft, 0, prio, 256)
    # LDS-heavy filler (the sort kernels: 71 KB per 256-thread workgroup)
    for w in (8, 12):
        for prio in (0, 1):
            run(w, 0, busy_iters, 4000, 256, 71 * 1024, prio, 2048)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "coresidency_probe.json"), "w") as f:
        json.dump(rows, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
