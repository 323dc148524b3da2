#!/usr/bin/env python
"""Same-box A/B of an environment switch of the in-tree library (VFIK_BLOCK, VFIK_TWO_WAVES, ...): alternating bench.py runs,
p10 of the HIP-event time per launch, like tools/ab_compare.py.
    python tools/ab_env.py VFIK_BLOCK=64 VFIK_BLOCK=128 [--workload C3] [--state warm] [--rounds 3]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(setting, workload, state):
    env = dict(os.environ)
    if setting:
        k, v = setting.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--no-cpu-baseline", "--host-path", "0",
                          "--rollout", "0", "--steps", "200", "--reps", "30", "--state", state, "--launch", "direct", "--secondary", "0"], env=env, capture_output=True, text=True, timeout=600)
    d = json.loads(out.stdout.strip().split("\n")[-1])
    return d["roofline"]["us_per_launch_p10"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--state", default="warm", choices=["warm", "cold"])
    a = ap.parse_args()
    names = [""] + a.settings
    res = {n: [] for n in names}
    for _ in range(a.rounds):
        for n in names:
            res[n].append(run(n, a.workload, a.state))
    base = sorted(res[""])[len(res[""]) // 2]
    print("workload %s, state %s" % (a.workload, a.state))
    for n in names:
        med = sorted(res[n])[len(res[n]) // 2]
        print("%-28s %s  median %.3f us (%+.1f %%)" % (n or "default", " ".join("%.3f" % x for x in res[n]), med, 100 * (med / base - 1)))


if __name__ == "__main__":
    main()
