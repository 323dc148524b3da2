#!/usr/bin/env python
"""Digest of tools/profile_round.sh's output:  python tools/profile_digest.py gpurun_out/<tag> <round>
Prints a markdown summary, copies the kernel-statistics CSVs to profiles/r<round>_kernel_stats_{warm,cold}_<W>.csv and rewrites
profiles/pmc_traffic.json (HBM bytes per launch, gfx950 correction applied)."""
import csv
import glob
import json
import os
import re
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALG = {"C2": 512 * 4096, "C3": 384 * 65536, "C3N": 384 * 65536, "C5": 696 * 65536, "C3F": 600 * 65536, "C5F": 996 * 65536, "C2F": 944 * 4096}
WAVES = {"C2": 4096 // 64, "C3": 1024, "C3N": 1024, "C5": 1024, "C3F": 1024, "C5F": 1024, "C2F": 4096 // 8}   # (C2: one lane per arm since round 4's end; 4096 // 8 before)


def one(pattern):
    f = glob.glob(pattern)
    return f[0] if f else None


def kernel_of(name):
    """`void vfik::(anonymous namespace)::cycle_kernel<float, 7, ...>(vfik::KArgs)` -> `cycle_kernel<float, 7, ...>`"""
    m = re.search(r"(cycle_\w+<[^>]*>)", name)
    return m.group(1) if m else name


def line_of(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None


def main():
    d, rnd = sys.argv[1], int(sys.argv[2])
    print("| workload | state | kernel | dispatches | rocprofv3 mean ns | median | min | frac from the CSV mean | the traced process's own HIP events, us | plain run (no tracer): us per launch, frac |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    ktrace = {}
    for w in ("C2", "C2F", "C3", "C3N", "C5", "C3F", "C5F"):
        plain = line_of(os.path.join(d, "bench_%s.json" % w))
        for state, tdir, under in (("warm", "trace_", "bench_under_rocprof_%s.json"), ("cold", "cold_trace_", "bench_cold_under_rocprof_%s.json")):
            st = one(os.path.join(d, tdir + w, "*", "*kernel_stats.csv"))
            tr = one(os.path.join(d, tdir + w, "*", "*kernel_trace.csv"))
            if not st:
                continue
            shutil.copy(st, os.path.join(ROOT, "profiles", "r%02d_kernel_stats_%s_%s.csv" % (rnd, state, w)))
            row = [r for r in csv.DictReader(open(st)) if "::cycle_" in r["Name"]][0]
            dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr)) if "::cycle_" in r["Kernel_Name"]]
            line = line_of(os.path.join(d, under % w))
            mean = float(row["AverageNs"])
            frac_csv = ALG[w] / (mean * 1e-9) / 1e9 / 8000.0
            if plain:
                pr = plain["roofline"] if state == "warm" else plain["roofline"].get("cold", {})
                pus = pr.get("us_per_launch_hip_events", pr.get("us_per_launch"))
                ptxt = "%.3f, %.3f" % (pus, pr["frac"]) if pus else "-"
            else:
                ptxt = "-"
            ktrace.setdefault(w, {"kernel": kernel_of(row["Name"]), "round": rnd, "algorithmic_bytes_per_launch": ALG[w]})[state] = {
                "mean_ns": mean, "median_ns": statistics.median(dur), "min_ns": int(row["MinNs"]), "dispatches": int(row["Calls"]), "frac": frac_csv,
                "traced_process_us_per_launch": line["roofline"]["us_per_launch_hip_events"] if line else None}
            # the same command launched one by one under the tracer (host-bound there: durations include queueing), kept for the record
            dst = one(os.path.join(d, ("direct_trace_" if state == "warm" else "direct_cold_trace_") + w, "*", "*kernel_stats.csv"))
            if dst:
                shutil.copy(dst, os.path.join(ROOT, "profiles", "r%02d_kernel_stats_direct_%s_%s.csv" % (rnd, state, w)))
                drow = [r for r in csv.DictReader(open(dst)) if "::cycle_" in r["Name"]][0]
                dline = line_of(os.path.join(d, ("bench_direct_under_rocprof_%s.json" if state == "warm" else "bench_direct_cold_under_rocprof_%s.json") % w))
                ktrace[w][state]["launched_one_by_one_under_the_tracer"] = {
                    "mean_ns": float(drow["AverageNs"]), "min_ns": int(drow["MinNs"]), "dispatches": int(drow["Calls"]),
                    "traced_process_us_per_launch": dline["roofline"]["us_per_launch_hip_events"] if dline else None}
            print("| %s | %s | `%s` | %s | %.0f | %.0f | %s | **%.3f** | %s | %s |" % (
                w, state, kernel_of(row["Name"]), row["Calls"], mean, statistics.median(dur), row["MinNs"], frac_csv,
                "%.3f" % line["roofline"]["us_per_launch_hip_events"] if line else "-", ptxt))
    if ktrace:
        with open(os.path.join(ROOT, "profiles", "kernel_trace.json"), "w") as f:
            json.dump(ktrace, f, indent=1)
    traffic = {}
    print()
    for w in ("C3", "C3N", "C5", "C3F", "C5F"):
        per = {}
        kname = None
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = one(os.path.join(d, "pmc_mem_%s_%s" % (w, c), "*", "*counter_collection.csv"))
            if not f:
                continue
            for r in csv.DictReader(open(f)):
                if "::cycle_" not in r["Kernel_Name"]:
                    continue
                kname = r["Kernel_Name"]
                per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        if "FETCH_SIZE" not in per or "WRITE_SIZE" not in per:
            continue
        fetch = statistics.median(per["FETCH_SIZE"].values())
        write = statistics.median(per["WRITE_SIZE"].values())
        hbm = fetch * 1024 * 2 + write * 1024
        traffic[w] = {"hbm_bytes_per_launch": hbm, "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
                      "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane coalesced stream: doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
                      "state": "cold (rotating input sets: the launches' inputs come from HBM)",
                      "dispatches_sampled": len(per["FETCH_SIZE"]), "kernel": kernel_of(kname), "round": rnd,
                      "algorithmic_bytes_per_launch": ALG[w]}
        print("%s: FETCH_SIZE %.1f KB x2 = %.2f MB, WRITE_SIZE %.1f KB = %.2f MB -> %.2f MB per launch vs %.2f MB algorithmic (ratio %.3f)"
              % (w, fetch, fetch * 2048 / 1e6, write, write * 1024 / 1e6, hbm / 1e6, ALG[w] / 1e6, hbm / ALG[w]))
    if traffic:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
    print()
    sqj = {}
    for w in ("C2", "C2F", "C3", "C3N", "C5", "C3F", "C5F"):
        f = one(os.path.join(d, "pmc_sq_" + w, "*", "*counter_collection.csv"))
        if not f:
            continue
        per = {}
        kname = None
        for r in csv.DictReader(open(f)):
            if "::cycle_" not in r["Kernel_Name"]:
                continue
            kname = r["Kernel_Name"]
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        if not per:
            continue
        print("%s per wave (%d waves), median over %d dispatches: " % (w, WAVES[w], len(next(iter(per.values())))) +
              ", ".join("%s %.0f" % (k, statistics.median(v.values()) / WAVES[w]) for k, v in sorted(per.items())))
        # per wave: what bench.py's roofline.valu quotes (in-kernel clock 2.15-2.2 GHz by the s_memtime stamps of tools/stamps.py)
        sqj[w] = {k: statistics.median(v.values()) / WAVES[w] for k, v in sorted(per.items())}
        sqj[w].update({"waves": WAVES[w], "round": rnd, "kernel": kernel_of(kname), "clock_ghz": 2.2,
                       "dispatches_sampled": len(next(iter(per.values())))})
    if sqj:
        with open(os.path.join(ROOT, "profiles", "pmc_sq.json"), "w") as f:
            json.dump(sqj, f, indent=1)


if __name__ == "__main__":
    main()
