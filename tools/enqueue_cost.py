#!/usr/bin/env python
"""What one launch costs the HOST (the enqueue loop of bench.py runs at this rate; at N = 8 eight such loops share the box's CPUs).

Prints, per call, for a C3-shaped launch (65 536 arms): a trivial ctypes call, Engine.step as bound method, the prebuilt
stepper (Engine.stepper: byref and prototype bound once, no Python-level checks on the hot path), and the kernel's launch period
for reference.  The enqueue time is measured over K launches into an idle stream, before the synchronize (no launch blocks:
the stream's queue holds thousands of packets)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
chain = robots.lwr()
w = synth.make_workload(chain, B, 8, seed=3, io_dtype=np.float32)
eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params())
eng.set_fields(w["fields"], w["nfields"])
q = torch.from_numpy(w["q"].astype(np.float32)).cuda()
out = torch.zeros(B, 7, dtype=torch.float32, device="cuda")
eng.use_stream(torch.cuda.current_stream().cuda_stream)
io = eng.make_io(q, qdot_out=out)
K = 200


def per_call(fn, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
    ts.sort()
    return ts[len(ts) // 2]


ver = eng.lib.vfik_abi_version
print("batch %d, %d launches per repetition, median of 15" % (B, K))
print("trivial ctypes call (vfik_abi_version)      %.2f us" % per_call(ver)[0])
e, t = per_call(lambda: eng.step(io))
print("Engine.step(io)                             %.2f us enqueue, %.2f us per launch incl. drain" % (e, t))
step = eng.stepper(io)
e, t = per_call(step)
print("Engine.stepper(io)()                        %.2f us enqueue, %.2f us per launch incl. drain" % (e, t))
ms = eng.time_steps(io, 20, 2000)
print("vfik_time_steps (C loop, HIP events)        %.2f us per launch" % (ms * 1e3 / 2000))
t0 = time.perf_counter()
ms = eng.time_steps(io, 0, 2000)
print("vfik_time_steps wall / launch               %.2f us" % ((time.perf_counter() - t0) * 1e6 / 2000))
eng.close()
