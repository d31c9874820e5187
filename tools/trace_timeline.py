#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: kernels longer than --min ms per stream, and the idle gaps of one stream.
usage: trace_timeline.py <kernel_trace.csv> [--min 2.0] [--from ms] [--to ms] [--gaps STREAM]"""
import csv
import re
import sys


def main():
    a = sys.argv[1:]
    path = a[0]
    mn = float(a[a.index("--min") + 1]) if "--min" in a else 2.0
    t_from = float(a[a.index("--from") + 1]) if "--from" in a else 0.0
    t_to = float(a[a.index("--to") + 1]) if "--to" in a else 1e18
    gaps = a[a.index("--gaps") + 1] if "--gaps" in a else None
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("og::", "").replace("void ", "")
    rows.sort(key=lambda r: r["s"])
    t0 = rows[0]["s"]
    prev_end = None
    for r in rows:
        s, e = (r["s"] - t0) / 1e6, (r["e"] - t0) / 1e6
        if s < t_from or s > t_to:
            continue
        if gaps is not None:
            if r["Stream_Id"] != gaps:
                continue
            if prev_end is not None and s - prev_end > 0.5:
                print(f"      gap {s - prev_end:7.2f} ms before {r['n'][:40]}")
            prev_end = e
        if e - s >= mn:
            print(f"{r['Stream_Id']:>2} {s:9.1f} {e:9.1f} {e - s:7.1f} {r['n'][:44]:44s} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg{r['Workgroup_Size_X']} v{r['VGPR_Count']}+{r['Accum_VGPR_Count']} lds{r['LDS_Block_Size']}")


if __name__ == "__main__":
    main()
