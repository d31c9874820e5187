"""Latency of ONE Montgomery product on a lone wave: the lane-local fe_mul (205 instructions, one lane per element) against the
wave-wide "w9" form (owshen_amd/csrc/field_w9.hip.h: the nine limbs in nine lanes, ~80 VALU instructions per product), as
chains x <- x y timed with s_memtime inside the kernel (og_field_mulchain_lat_d).  Bytes must agree.
Writes gpurun_out/w9_probe.json (profiles/r06h_w9_probe.json)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from owshen_amd import api  # noqa: E402


def main():
    ctx = api.Context(0)
    out = {"what": __doc__, "device": torch.cuda.get_device_name(0), "unit": "shader cycles per product (one wave, s_memtime)", "rows": []}
    iters = 4096
    for field, name in ((0, "Fr"), (1, "Fq")):
        for n in (1, 64, 256):
            g = torch.Generator().manual_seed(7 + n)
            x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
            x[:, 31] &= 0x1F
            y = x.flip(0).contiguous()
            res = {}
            ref = None
            for form, fname in ((0, "lane_local"), (1, "w9")):
                xd, yd = x.cuda(), y.cuda()
                ctx.field_mulchain_lat(field, form, xd, yd, 16)  # warm the code object
                xd = x.cuda()
                ms, cyc = ctx.field_mulchain_lat(field, form, xd, yd, iters)
                b = bytes(xd.cpu().numpy())
                if ref is None:
                    xr = x.cuda()
                    ctx.field_mulchain(field, xr, yd, iters)
                    ref = bytes(xr.cpu().numpy())
                res[fname] = {"cycles_per_product": round(cyc / iters, 1), "kernel_ms": round(ms, 4), "bytes_equal_throughput_form": b == ref}
            res["speedup"] = round(res["lane_local"]["cycles_per_product"] / res["w9"]["cycles_per_product"], 3)
            row = {"field": name, "elements": n, "waves_lane_local": (n + 63) // 64, "waves_w9": n, **res}
            out["rows"].append(row)
            print(json.dumps(row))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/w9_probe.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
