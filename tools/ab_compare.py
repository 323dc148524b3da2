#!/usr/bin/env python
"""Same-box A/B of two builds of libvfik_hip.so: alternating bench.py runs; per run the p10 over 30 repetitions of the
HIP-event time per launch (200 launches each), then median and best over the rounds (a box's clock wanders by several %).

Between gpurun boxes the same binary varies by +-1.5 %, so only runs on one box compare.  Build the variant out of
tree with tools/build_variant.sh (copy of vfclik_amd/csrc + include under /tmp, flags / patches, `make`; the library lands in
tools/variants/) and run on the GPU box:

    python tools/ab_compare.py tools/variants/<name>.so [--workload C3] [--rounds 3] [--state cold]

The in-tree library is the baseline; variants must have the in-tree ABI version (the binding refuses any other).  (Never
build variants in tree: the source-hash stamp would then describe them.)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, workload, state):
    env = dict(os.environ)
    env.pop("VFIK_HIP_LIB", None)
    if lib:
        env["VFIK_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--no-cpu-baseline", "--host-path", "0",
                          "--rollout", "0", "--steps", "200", "--reps", "30", "--state", state, "--launch", "direct", "--secondary", "0"], env=env, capture_output=True, text=True, timeout=600)
    d = json.loads(out.stdout.strip().split("\n")[-1])
    return d["roofline"]["us_per_launch_p10"], d["max_abs_err_rad_s"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--state", default="warm", choices=["warm", "cold"],
                    help="warm: back-to-back launches over one input set (Infinity-Cache resident); cold: rotating input sets (inputs from HBM)")
    a = ap.parse_args()
    print("workload %s, state %s" % (a.workload, a.state))
    names = ["in-tree"] + a.variants
    res = {n: [] for n in names}
    err = {}
    for _ in range(a.rounds):
        for n in names:
            us, e = run(None if n == "in-tree" else n, a.workload, a.state)
            res[n].append(us)
            err[n] = e
    base = sorted(res["in-tree"])[len(res["in-tree"]) // 2]
    for n in names:
        med = sorted(res[n])[len(res[n]) // 2]
        print("%-40s %s  median %.3f us (%+.1f %%)  best %.3f (%+.1f %%)  max err %.3g" % (n, " ".join("%.3f" % x for x in res[n]), med, 100 * (med / base - 1), min(res[n]), 100 * (min(res[n]) / min(res["in-tree"]) - 1), err[n]))


if __name__ == "__main__":
    main()
