"""A/B probe for the field layer: instruction-level chains (og_ubench_cycles) and fe_mul throughput at several occupancies
(og_field_mulchain_d).  usage: python tools/ab_probe.py <tag> [--ubench]   -> gpurun_out/ab_probe_<tag>.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api  # noqa: E402


def main():
    tag = sys.argv[1]
    ctx = api.Context(0)
    res = {"tag": tag}
    if "--ubench" in sys.argv:
        names = {0: "mad_independent", 21: "chains1", 15: "chains2", 23: "chains3", 16: "chains4", 25: "chains5", 26: "chains6", 28: "chains8", 32: "chains12",
                 41: "pingpong1", 42: "pingpong2", 44: "pingpong4"}
        if "--banks" in sys.argv:
            names = {100 + 5 * a + b: f"bank_a{a}_b{'S' if b == 4 else b}" for a in range(4) for b in range(5)}
        iters = 4096
        n_cu = torch.cuda.get_device_properties(0).multi_processor_count
        ub = {}
        for blocks in (n_cu, 2 * n_cu, 4 * n_cu):
            wps = blocks / n_cu  # 256-thread blocks: 4 waves, one per SIMD
            for k, name in names.items():
                ms, cyc = ctx.ubench_cycles(k, iters, blocks)
                ub[f"{name}@{wps:g}w"] = {"ms": ms, "cycles_per_wave_instr": cyc / (iters * 16),
                                          "cycles_per_instr_per_simd": cyc / (iters * 16 * wps)}
        res["ubench"] = ub
    mm = {}
    for n in () if "--no-mulmod" in sys.argv else (256 * 256, 256 * 256 * 2, 256 * 256 * 3, 256 * 256 * 4, 256 * 256 * 8, 1 << 22):
        g = torch.Generator().manual_seed(n)
        x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
        x[:, 31] &= 0x1F
        xd, yd = x.cuda(), x.flip(0).contiguous().cuda()
        for field in (0, 1):
            ctx.field_mulchain(field, xd, yd, 16)
            it = 512
            ms = ctx.field_mulchain(field, xd, yd, it)
            mm[f"n={n},field={field}"] = {"ms": ms, "mulmod_per_s": n * it / (ms * 1e-3)}
    res["mulmod"] = mm
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"ab_probe_{tag}.json"), "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.get("ubench", {}).items():
        print(k, round(v["cycles_per_wave_instr"], 2), round(v["cycles_per_instr_per_simd"], 2))
    for k, v in mm.items():
        print(tag, k, f"{v['mulmod_per_s']:.3e}")


if __name__ == "__main__":
    main()
