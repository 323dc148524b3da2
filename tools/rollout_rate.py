#!/usr/bin/env python
"""Closed-loop rollouts (vfik_rollout, 100 cycles per launch, q integrated on the device): microseconds per control cycle of the whole batch.
C3's and C3N's batches (65 536 arms, 8 obstacles, float32 I/O); with VFIK_HIP_LIB set, of that library."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

B, K = 65536, 100
print("library: %s" % os.environ.get("VFIK_HIP_LIB", "in-tree"))
for name, robot, nobs, flags in (("C3", "lwr", 8, 0), ("C3N", "lwr", 8, 5), ("6 joints", "powercube6", 8, 0), ("C5 (stepped: single-cycle launches)", "lwr_dual14", 16, 7)):
    chain = getattr(robots, robot)()
    w = synth.make_workload(chain, B, nobs, seed=1, io_dtype=np.float32)
    eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=max(8, nobs), params=_abi.default_params(flags=flags))
    eng.set_fields(w["fields"], w["nfields"])
    n = chain.n
    dq, do, dqo = eng.dev_alloc(B * n * 4), eng.dev_alloc(B * n * 4), eng.dev_alloc(B * n * 4)
    eng.h2d(dq, w["q"].astype(np.float32))
    io = eng.make_io(dq, qdot_out=do)
    for _ in range(2):
        eng.rollout(io, K, 1e-3, q_out=dqo)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.rollout(io, K, 1e-3, q_out=dqo)
    eng.sync()
    ms = (time.perf_counter() - t0) * 1e3
    print("%-40s %6.2f us per cycle" % (name, ms * 1e3 / (5 * K)))
    eng.close()
