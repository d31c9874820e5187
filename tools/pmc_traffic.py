#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json (read by bench.py for roofline.traffic / valu_util / mad_issue_frac) from the per-dispatch
rocprofv3 PMC CSVs that tools/gpu_round2.sh (stage pmc / pmc_dense) leaves under gpurun_out/.

usage: pmc_traffic.py <tag> <variant: sparse|dense> <fetch_dir> <write_dir> [<sq_dir> <tcc_dir>] --points a,b,l,h [--proofs N]

--proofs: proofs per launch.  The persistent accumulation kernels (round 3) have the same grid whatever the sub-batch size, so
the PMC runs pin the sub-batch plan (OG_SUB_PLAN=N with --batch 2N: every launch covers N proofs) and say so here.

Only the FULL-SIZE launches (the largest grid of each kernel = one whole sub-batch of proofs) are used; the per-launch figures
are averages over the query launches (A, B1, L, H for G1; B2 for G2), the same average bench.py's HIP-event timing takes.
--points gives the per-proof point counts of the queries in launch order, from the bench line's config.n_dense.
Instruction-issue figures use the cycle-calibrated og_ubench_cycles costs (profiles/<tag>_probe.json) and the multiply-add
counts of one mixed addition from the generated field routines (owshen_amd/csrc/mont_gfx950.inc): G1 1572 v_mad_u64_u32
(819 product + 729 reduction terms + 24 addend limbs) + 81 v_mul_lo_u32, G2 4836 + 162.
"""
import csv
import glob
import json
import os
import sys

MADS = {"accumulate_g1": (1572, 81), "accumulate_g2": (4836, 162)}
KERNELS = {"accumulate_g1": ("k_accumulate<og::Fe<og::FqParams>", "k_accumulate_p<og::Fe<og::FqParams>"),
           "accumulate_g2": ("k_accumulate<og::Fq2", "k_accumulate_p<og::Fq2", "k_accumulate_g2_lds")}
N_SIMD, N_XCD, NWIN = 1024, 8, 16


def dispatches(d, kernel_sub):
    """[(grid, {counter: value}, duration_ns)] in dispatch order for kernels whose name contains kernel_sub"""
    out, index = [], {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if not any(k in row["Kernel_Name"] for k in kernel_sub):
                    continue
                key = row["Dispatch_Id"]
                if key not in index:
                    index[key] = len(out)
                    out.append([int(row["Grid_Size"]), {}, int(row["End_Timestamp"]) - int(row["Start_Timestamp"])])
                out[index[key]][1][row["Counter_Name"]] = float(row["Counter_Value"])
    return out


def main():
    args = sys.argv[1:]
    proofs_arg, total_proofs = None, None
    if "--proofs" in args:
        proofs_arg = int(args[args.index("--proofs") + 1])
        del args[args.index("--proofs"):args.index("--proofs") + 2]
    if "--total-proofs" in args:  # the profiled step proved this many proofs under the prover's OWN sub-batch plan (round 4): every
        total_proofs = int(args[args.index("--total-proofs") + 1])   # launch of the step is averaged, proofs per launch = total / sub-batches
        del args[args.index("--total-proofs"):args.index("--total-proofs") + 2]
    wins = None
    if "--windows" in args:  # windows per point of the A, B, L, H queries (og_pk_windows: 15 for 17-bit windows, 16 for 16-bit)
        wins = [int(x) for x in args[args.index("--windows") + 1].split(",")]
        del args[args.index("--windows"):args.index("--windows") + 2]
    pts = [int(x) for x in args[args.index("--points") + 1].split(",")]
    args = args[:args.index("--points")]
    tag, variant, dirs = args[0], args[1], args[2:]
    probe = {}
    try:
        probe = json.load(open(os.path.join("profiles", f"{tag}_probe.json")))["valu_rates"]
    except (OSError, KeyError, ValueError):
        pass
    c_mad = probe.get("v_mad_u64_u32", {}).get("cycles_per_wave_instr_per_simd", 4.4)
    c_mul = probe.get("v_mul_lo_u32", {}).get("cycles_per_wave_instr_per_simd", 4.2)
    res = {}
    for key, sub in KERNELS.items():
        per_dir = [dispatches(d, sub) for d in dirs]
        if not per_dir[0]:
            continue
        full = max(g for g, _, _ in per_dir[0])
        proofs = proofs_arg if proofs_arg else full // 32768
        sel = [[x for x in dd if x[0] == full] for dd in per_dir]
        nq = 4 if key == "accumulate_g1" else 1
        qpts = pts[:4] if key == "accumulate_g1" else [pts[1]]
        n = len(sel[0])
        if total_proofs:
            proofs = total_proofs * nq / n   # (the persistent launches all have the same grid: `sel` is every launch of the step)
        c = {}
        for dd in sel:
            for i, (_, cs, _) in enumerate(dd):
                for name, v in cs.items():
                    c.setdefault(name, [0.0] * len(dd))[i] = v
        dur_ms = sum(x[2] for x in sel[0]) / n * 1e-6
        points = sum(qpts[i % nq] for i in range(n)) / n * proofs
        qwin = (wins[:4] if key == "accumulate_g1" else [wins[1]]) if wins else [NWIN] * nq
        point_windows = sum(qpts[i % nq] * qwin[i % nq] for i in range(n)) / n * proofs   # mixed additions per launch (lane level)
        entry = {"launches_profiled": n, "proofs_per_launch": round(proofs, 2), "points_per_launch": int(points), "avg_ms_profiled": round(dur_ms, 3),
                 "windows_per_point": round(point_windows / points, 3)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch = sum(c["FETCH_SIZE"]) / n * 1024
            write = sum(c["WRITE_SIZE"]) / n * 1024
            entry.update(fetch_bytes_per_launch=int(fetch), write_bytes_per_launch=int(write), hbm_bytes_per_launch=int(fetch + write),
                         bytes_per_point=round((fetch + write) / points, 1), hbm_GBps_profiled=round((fetch + write) / dur_ms / 1e6, 1))
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            insts = sum(c["SQ_INSTS_VALU"]) / n                 # wave-instructions per launch
            cyc = sum(c["GRBM_GUI_ACTIVE"]) / n / N_XCD          # kernel duration in shader cycles
            madds = point_windows / 64                           # wave-level mixed additions per launch
            mad, mul = MADS[key]
            entry.update(valu_insts_per_madd=round(insts / madds, 1), kernel_cycles=int(cyc),
                         effective_clock_GHz=round(cyc / (dur_ms * 1e6), 3),
                         valu_util=round(insts * 4 / (N_SIMD * cyc), 4),
                         mad_issue_frac=round(madds * (mad * c_mad + mul * c_mul) / (N_SIMD * cyc), 4))
        if "TCC_HIT_sum" in c:
            h, m = sum(c["TCC_HIT_sum"]), sum(c["TCC_MISS_sum"])
            entry["l2_hit_rate"] = round(h / (h + m), 4)
        if "SQ_WAVE_CYCLES" in c:
            entry["wave_cycles_per_launch_quads"] = int(sum(c["SQ_WAVE_CYCLES"]) / n)
            entry["wait_inst_any_frac_of_wave_cycles"] = round(sum(c["SQ_WAIT_INST_ANY"]) / sum(c["SQ_WAVE_CYCLES"]), 4)
            entry["active_inst_any_frac_of_wave_cycles"] = round(sum(c["SQ_ACTIVE_INST_ANY"]) / sum(c["SQ_WAVE_CYCLES"]), 4)
        res[key] = entry
    path = os.path.join("profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(path))
        if "sparse" not in doc and "dense" not in doc:
            doc = {}
    except (OSError, ValueError):
        doc = {}
    if total_proofs:
        res["source"] = (f"profiles/{tag}_pmc_summary.txt: rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_* + GRBM_GUI_ACTIVE, separate "
                         f"runs, kernel-trace only) over `bench.py --batch {total_proofs} --steps 1 --warmup 0 --no-cpu --no-legs --no-isolated` ({variant} padding): "
                         "ONE step of the headline itself under the prover's own sub-batch plan, every accumulation launch of it averaged")
    else:
        res["source"] = (f"profiles/{tag}_pmc_summary.txt: rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_* + GRBM_GUI_ACTIVE, separate "
                     f"runs, kernel-trace only) over `OG_SUB_PLAN={proofs_arg or '...'} bench.py --batch {2 * proofs_arg if proofs_arg else 512} --steps 1 --warmup 0 "
                     f"--no-cpu --no-legs` ({variant} padding): every launch covers {proofs_arg or 'one sub-batch of'} proofs, the steady-state "
                     "sub-batch size of the batch-1024 headline")
    doc[variant] = res
    doc["note"] = ("valu_util = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; "
                   "mad_issue_frac = the same with only the v_mad_u64_u32 / v_mul_lo_u32 of the mixed additions at their measured issue cost "
                   "(og_ubench_cycles); FETCH_SIZE taken as reported: the accesses are 64 B / 128 B gathers, for which the raw value matches "
                   "the gather volume (the 2x under-count of MI355X_MICROARCH.md applies to wide coalesced streams)")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
