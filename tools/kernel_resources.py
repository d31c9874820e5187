#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy of every gfx950 kernel of the library, from the compiler's own remarks
(`hipcc -Rpass-analysis=kernel-resource-usage`, device-only compile of each translation unit).  What fits beside what
 is decided by these numbers.   usage: python tools/kernel_resources.py [> profiles/rNN_kernel_resources.txt]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "owshen_amd", "csrc")
UNITS = ["msm_g1.hip", "msm_g2.hip", "msm.hip", "ntt.hip", "groth16.hip", "witness.hip", "mimc7.hip", "eddsa.hip", "keygen.hip", "field_ops.hip"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for u in UNITS:
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--offload-device-only",
                   "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(SRC, u), "-o", os.path.join(tmp, u + ".o")]
            err = subprocess.run(cmd, capture_output=True, text=True, cwd=SRC).stderr
            cur = None
            for line in err.splitlines():
                m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k == "Function Name":
                    cur = {"unit": u, "name": v}
                    rows.append(cur)
                elif cur is not None:
                    cur[k.split(" ")[0]] = v
    names = demangle([r["name"] for r in rows])
    print(f"{'kernel':78s} {'VGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'waves/SIMD':>10s}  unit")
    for r, n in sorted(zip(rows, names), key=lambda x: (x[0]["unit"], x[1])):
        n = re.sub(r"\(.*", "", n).replace("og::", "").replace("void ", "")
        print(f"{n[:78]:78s} {r.get('VGPRs', '?'):>5s} {r.get('ScratchSize', '?'):>8s} {r.get('LDS', '?'):>7s} {r.get('Occupancy', '?'):>10s}  {r['unit']}")


if __name__ == "__main__":
    sys.exit(main())
