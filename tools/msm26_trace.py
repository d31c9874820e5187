#!/usr/bin/env python3
"""The last 2^26-point MSM of a `rocprofv3 --kernel-trace` run of `bench.py --workload msm26`, as a timeline of its two window
halves (og_msm_d runs them on the context's two lanes, capi.hip msm_lone_halves): per stream, every kernel with start / end
relative to the MSM's first kernel, then what the chip was doing in each millisecond slice -- which of {sort, accumulation,
heavy buckets, tail} had a kernel resident -- and the three numbers VERDICT r5 item 9 asks about: the span, the time in which an
accumulation kernel was resident (the VALU-bound part) and the time in which NONE was (what a better overlap could still hide).
usage: msm26_trace.py <kernel_trace.csv>"""
import csv
import re
import sys


def kind(name):
    if "k_accumulate_pieces" in name or "k_accumulate_p<" in name or "k_pieces_combine" in name:
        return "acc"
    if "k_accumulate_heavy" in name or "k_heavy" in name:
        return "heavy"
    if any(t in name for t in ("k_lone_", "k_sub_", "k_scan_slice", "k_sort_", "k_digit", "k_bucket_order")):
        return "sort"
    return "tail"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("og::", "").replace("void ", "")
    rows.sort(key=lambda r: r["s"])
    # the last MSM: everything from the last k_lone_digits launch pair on
    starts = [i for i, r in enumerate(rows) if "k_lone_digits" in r["n"]]
    if not starts:
        print("no k_lone_digits kernel in the trace")
        return
    first = starts[-2] if len(starts) >= 2 and rows[starts[-1]]["s"] - rows[starts[-2]]["s"] < 5e6 else starts[-1]
    ks = [r for r in rows[first:] if any(t in r["n"] for t in ("k_", ))]
    t0 = ks[0]["s"]
    end = max(r["e"] for r in ks)
    print(f"# last MSM of the run: {len(ks)} kernels, span {(end - t0) / 1e6:.2f} ms")
    print("# stream  start_ms  end_ms  dur_ms  kind   kernel")
    for r in ks:
        if (r["e"] - r["s"]) / 1e6 >= 0.3:
            print(f"{r['Stream_Id']:>6} {(r['s'] - t0) / 1e6:9.2f} {(r['e'] - t0) / 1e6:7.2f} {(r['e'] - r['s']) / 1e6:7.2f}  {kind(r['n']):5s}  {r['n'][:60]}")
    # occupancy of the span by kind (0.1 ms slices)
    step = 100000
    n = (end - t0 + step - 1) // step
    res = {k: [False] * n for k in ("sort", "acc", "heavy", "tail")}
    for r in ks:
        k = kind(r["n"])
        for i in range((r["s"] - t0) // step, min(n, (r["e"] - t0) // step + 1)):
            res[k][i] = True
    tot = lambda f: sum(1 for i in range(n) if f(i)) * step / 1e6   # noqa: E731
    print(f"# span {n * step / 1e6:.1f} ms; an accumulation kernel resident {tot(lambda i: res['acc'][i]):.1f} ms; "
          f"sort without accumulation {tot(lambda i: res['sort'][i] and not res['acc'][i]):.1f} ms; "
          f"heavy / tail without accumulation {tot(lambda i: (res['heavy'][i] or res['tail'][i]) and not res['acc'][i] and not res['sort'][i]):.1f} ms; "
          f"sort UNDER accumulation {tot(lambda i: res['sort'][i] and res['acc'][i]):.1f} ms; nothing resident {tot(lambda i: not any(res[k][i] for k in res)):.1f} ms")


if __name__ == "__main__":
    main()
