#!/usr/bin/env python
"""Digest of tools/profile_round.sh's output:  python tools/profile_digest.py gpurun_out/<tag> <round>
Prints a markdown summary and rewrites profiles/pmc_traffic.json (HBM bytes per launch, gfx950 correction applied)."""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALG = {"C2": 512 * 4096, "C3": 384 * 65536, "C3N": 384 * 65536, "C5": 696 * 65536}


def one(pattern):
    f = glob.glob(pattern)
    return f[0] if f else None


def main():
    d, rnd = sys.argv[1], int(sys.argv[2])
    print("| workload | kernel | dispatches | mean ns (rocprofv3) | median | min | bench line under rocprofv3: us/launch (HIP events) | frac from the CSV | frac the line prints | un-profiled us/launch | un-profiled frac |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for w in ("C2", "C3", "C3N", "C5"):
        st = one(os.path.join(d, "trace_" + w, "*", "*kernel_stats.csv"))
        tr = one(os.path.join(d, "trace_" + w, "*", "*kernel_trace.csv"))
        if not st:
            continue
        row = [r for r in csv.DictReader(open(st)) if "::cycle_" in r["Name"]][0]
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr)) if "::cycle_" in r["Kernel_Name"]]
        line = json.loads(open(os.path.join(d, "bench_under_rocprof_%s.json" % w)).read().strip().splitlines()[-1])
        plain = json.loads(open(os.path.join(d, "bench_%s.json" % w)).read().strip().splitlines()[-1])
        mean = float(row["AverageNs"])
        frac_csv = ALG[w] / (mean * 1e-9) / 1e9 / 8000.0
        print("| %s | `%s` | %s | %.0f | %.0f | %s | %.3f | %.3f | %.3f | %.3f | %.3f |" % (
            w, row["Name"].split("::")[-1].replace("(vfik::KArgs)", ""), row["Calls"], mean, statistics.median(dur), row["MinNs"],
            line["roofline"]["us_per_launch_hip_events"], frac_csv, line["roofline"]["frac"],
            plain["roofline"]["us_per_launch_hip_events"], plain["roofline"]["frac"]))
    traffic = {}
    print()
    for w in ("C3", "C3N", "C5"):
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
                      "dispatches_sampled": len(per["FETCH_SIZE"]), "kernel": kname.split("::")[-1].replace("(vfik::KArgs)", ""), "round": rnd,
                      "algorithmic_bytes_per_launch": ALG[w]}
        print("%s: FETCH_SIZE %.1f KB x2 = %.2f MB, WRITE_SIZE %.1f KB = %.2f MB -> %.2f MB per launch vs %.2f MB algorithmic (ratio %.3f)"
              % (w, fetch, fetch * 2048 / 1e6, write, write * 1024 / 1e6, hbm / 1e6, ALG[w] / 1e6, hbm / ALG[w]))
    if traffic:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
    print()
    for w in ("C3", "C3N", "C5"):
        f = one(os.path.join(d, "pmc_sq_" + w, "*", "*counter_collection.csv"))
        if not f:
            continue
        per = {}
        for r in csv.DictReader(open(f)):
            if "::cycle_" not in r["Kernel_Name"]:
                continue
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        print("%s per wave (1024 waves), median over %d dispatches: " % (w, len(next(iter(per.values())))) +
              ", ".join("%s %.0f" % (k, statistics.median(v.values()) / 1024.0) for k, v in sorted(per.items())))


if __name__ == "__main__":
    main()
