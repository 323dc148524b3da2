#!/usr/bin/env python
"""How the kernel durations of a rocprofv3 --kernel-trace depend on the launch cadence: duration statistics of the cycle kernel by the
idle gap in front of each dispatch, for the trace with graph-replayed launches (trace_<W>) and the one launched one by one
(direct_trace_<W>) of a profile round.      python tools/trace_gaps.py gpurun_out/<tag> C3 [C5 ...]"""
import collections
import csv
import glob
import statistics as st
import sys

d = sys.argv[1]
for w in sys.argv[2:]:
    for tag, what in (("trace_" + w, "launches replayed from a hipGraph"), ("direct_trace_" + w, "launched one by one")):
        fs = glob.glob("%s/%s/*/*kernel_trace.csv" % (d, tag))
        if not fs:
            continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "cycle_" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        gaps = [int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) for i in range(1, len(rows))]
        s2s = [int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["Start_Timestamp"]) for i in range(1, len(rows))]
        q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
        print("%s, warm, %s: %d dispatches; duration mean %.0f median %.0f p10 %.0f p90 %.0f ns; start-to-start median %.0f ns"
              % (w, what, len(dur), st.mean(dur), st.median(dur), q(dur, 0.1), q(dur, 0.9), st.median(s2s)))
        b = collections.defaultdict(list)
        for g, x in zip(gaps, dur[1:]):
            b[0 if g < 500 else 1 if g < 3000 else 2 if g < 6000 else 3].append(x)
        for k in sorted(b):
            print("    idle gap in front %-10s %6d dispatches, duration median %.0f ns" % (["< 0.5 us", "0.5-3 us", "3-6 us", "> 6 us"][k], len(b[k]), st.median(b[k])))
