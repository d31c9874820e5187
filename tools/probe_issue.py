"""VALU issue cost of the Montgomery product's instruction mix by occupancy (og_ubench_cycles, s_memtime shader cycles):
the single instructions, and one COLUMN of the generated product -- 9 v_mad_u64_u32 on one accumulator, the mask of the output
limb, the 64-bit shift of the carry -- as one asm statement, at 1 .. 8 resident waves per SIMD.  Writes gpurun_out/probe_issue.json
(profiles/r05_probe_issue.json)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from owshen_amd import api  # noqa: E402

KINDS = {0: "v_mad_u64_u32 x16 independent (sdst s[2:3])", 66: "v_mad_u64_u32 x16 independent (sdst vcc)", 17: "v_mad_u64_u32 one chain (vcc)",
         68: "v_mad_u64_u32 one chain, SGPR factor", 6: "v_add_u32", 61: "v_and_b32 literal", 62: "v_lshrrev_b32", 13: "v_lshrrev_b64",
         60: "v_alignbit_b32", 63: "v_bfe_u32", 64: "column: 9 mad + and + lshrrev_b64 (one asm statement)",
         65: "column: 9 mad + and (one asm statement)"}


def main():
    ctx = api.Context(0)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    iters = 4096
    out = {"what": __doc__, "device": torch.cuda.get_device_name(0), "unit": "shader cycles per X per SIMD (X = one instruction, or one column)", "by_waves_per_simd": {}}
    for wps in (1, 2, 3, 4, 5, 6, 7, 8):
        blocks = n_cu * wps                      # 256-lane blocks: one wave on each SIMD of a CU
        row = {}
        for k, name in KINDS.items():
            ms, cyc = ctx.ubench_cycles(k, iters, blocks)
            row[name] = round(cyc / (iters * 16 * wps), 3)
        out["by_waves_per_simd"][str(wps)] = row
        print(wps, "waves/SIMD:", {k.split(" (")[0][:28]: v for k, v in row.items()})
    # the generated Montgomery product itself (fe_mul: one asm statement of 205 instructions) as a chain per lane, by occupancy
    out["fe_mul_chain"] = {"unit": "shader cycles per wave-level product per SIMD (effective clock from the v_mad run of the same occupancy)", "by_waves_per_simd": {}}
    for wps in (1, 2, 3, 4, 5, 6, 7, 8):
        n = n_cu * 4 * 64 * wps
        g = torch.Generator().manual_seed(n)
        x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
        x[:, 31] &= 0x1F
        xd, yd = x.cuda(), x.flip(0).contiguous().cuda()
        row = {}
        for field, name in ((0, "Fr"), (1, "Fq")):
            ctx.field_mulchain(field, xd, yd, 64)
            it = 2048
            ms = ctx.field_mulchain(field, xd, yd, it)
            ms0, cyc0 = ctx.ubench_cycles(66, 4096, n_cu * wps)
            clk = cyc0 / (ms0 * 1e-3)                      # shader cycles per second during a multiply-add run at this occupancy
            row[name] = {"mulmod_per_s": round(n * it / (ms * 1e-3) / 1e9, 2), "cycles_per_product_per_simd": round(ms * 1e-3 * clk / (it * wps), 1)}
        out["fe_mul_chain"]["by_waves_per_simd"][str(wps)] = row
        print(wps, "waves/SIMD fe_mul chain:", row)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/probe_issue.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
