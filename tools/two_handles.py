#!/usr/bin/env python
"""Independent batches in flight on ONE GPU: H handles (own stream each) of 65 536 arms launched round-robin from one host thread
against one handle launching the same number of cycles back to back.  The batches are independent problems (another handle, other
arms): their launches carry no dependency and overlap one launch's boundary with the other's arithmetic.  (Consecutive cycles of ONE
batch depend on each other through the robot and cannot.)     python tools/two_handles.py [--workload C3|C3N|C5]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
a = ap.parse_args()
robot, nobs, flags = {"C3": ("lwr", 8, 0), "C3N": ("lwr", 8, 5), "C5": ("lwr_dual14", 16, 7)}[a.workload]
chain = robots.by_name(robot)
B, K = 65536, 400
print("%s: %d arms per handle, float32 I/O; %d launches per timed region, median of 7; us per launch of ONE batch's cycle" % (a.workload, B, K))
for H in (1, 2, 3, 4):
    sets = []
    for h in range(H):
        w = synth.make_workload(chain, B, nobs, seed=10 + h, io_dtype=np.float32)
        eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=nobs, params=_abi.default_params(flags=flags))
        eng.set_fields(w["fields"], w["nfields"])
        st = torch.cuda.Stream()
        eng.use_stream(st.cuda_stream)
        q = torch.from_numpy(w["q"].astype(np.float32)).cuda()
        out = torch.zeros(B, chain.n, dtype=torch.float32, device="cuda")
        sets.append((eng, eng.stepper(eng.make_io(q, qdot_out=out)), st, q, out))
    steps = [sets[i % H][1] for i in range(K)]
    ts = []
    for rep in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in steps:
            s()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6 / K)
    us = float(np.median(ts[1:]))
    print("  %d handle(s): %.3f us per launch, %.3e cycles/s, %.3f of 8 TB/s by algorithmic bytes" % (H, us, B / us * 1e6, {"C3": 384, "C3N": 384, "C5": 696}[a.workload] * B / us / 1e3 / 8000.0), flush=True)
    for e in sets:
        e[0].close()
