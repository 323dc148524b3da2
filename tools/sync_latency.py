#!/usr/bin/env python
"""Where the fixed cost of a short timed region goes (bench.py --steps 20): launch latency of the first kernel, host enqueue
per launch, and the wake-up latency of the closing synchronize -- torch.cuda.synchronize() alone vs a busy poll on an event."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

chain = robots.lwr()
B = 65536
w = synth.make_workload(chain, B, 8, seed=1, io_dtype=np.float32)
eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params())
eng.set_fields(w["fields"], w["nfields"])
q = torch.from_numpy(w["q"].astype(np.float32)).cuda()
out = torch.zeros(B, 7, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
eng.use_stream(stream.cuda_stream)
io = eng.make_io(q, qdot_out=out)
for _ in range(50):
    eng.step(io)
torch.cuda.synchronize()


def med(f, n=200):
    xs = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        xs.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(xs)), float(np.percentile(xs, 10)), float(np.percentile(xs, 90))


def run(K, how):
    def f():
        for _ in range(K):
            eng.step(io)
        if how == "poll":
            ev = torch.cuda.Event()
            ev.record(stream)
            while not ev.query():
                pass
        elif how == "engsync":
            eng.sync()
        torch.cuda.synchronize()
    return f


print("idle torch.cuda.synchronize(): %.1f us (p10 %.1f, p90 %.1f)" % med(lambda: torch.cuda.synchronize()))
for K in (1, 20, 200):
    for how in ("torch", "engsync", "poll"):
        m, a, b = med(run(K, how), 100 if K < 200 else 30)
        print("K %3d  closing sync: %-8s total %.1f us  per step %.2f us (p10 %.2f p90 %.2f)" % (K, how, m, m / K, a / K, b / K))
