"""First-contact GPU probe: VALU instruction rates, Montgomery
mulmod/s, MiMC7 tree time.  Writes gpurun_out/probe.json."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owshen_amd import api  # noqa: E402

KINDS = ["v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co_u32", "v_addc_co_u32", "v_lshl_add_u64",
         "v_add_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_mov_b32", "v_fma_f64", "v_add_f64"]


def main():
    ctx = api.Context(0)
    res = {"device": torch.cuda.get_device_name(0), "cpu_count": os.cpu_count()}
    iters, blocks = 4096, 256 * 8
    rates = {}
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    waves_per_simd = blocks * 4 / (n_cu * 4)          # all resident at once (8 per SIMD at 2048 blocks)
    for k, name in enumerate(KINDS):
        ms, cyc = ctx.ubench_cycles(k, iters, blocks)
        lane_ops = iters * 16 * blocks * 256
        # measured in shader cycles (s_memtime inside the kernel): no clock assumption
        rates[name] = {"ms": ms, "lane_ops_per_s": lane_ops / (ms * 1e-3), "wave_cycles": cyc,
                       "cycles_per_wave_instr_per_simd": cyc / (iters * 16 * waves_per_simd),
                       "effective_clock_GHz": cyc / (ms * 1e-3) / 1e9}
    res["valu_rates"] = rates
    # occupancy sweep of the raw mulmod chain
    mm = {}
    for n in (256 * 256, 256 * 256 * 2, 256 * 256 * 4, 256 * 256 * 8, 1 << 22):
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
    # MiMC7 tree (C5 shape on one GPU)
    for k in (16, 20):
        n = 1 << k
        g = torch.Generator().manual_seed(2)
        x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
        x[:, 31] &= 0x1F
        xd = x.cuda()
        ctx.mimc7_tree_build(xd)
        torch.cuda.synchronize()
        t = time.time()
        ctx.mimc7_tree_build(xd)
        torch.cuda.synchronize()
        dt = time.time() - t
        res[f"mimc7_tree_2^{k}"] = {"s": dt, "hash_per_s": (n - 1) / dt, "mulmod_per_s": (n - 1) * 728 / dt}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
