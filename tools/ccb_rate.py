#!/usr/bin/env python
"""Rates of the DROP-IN path: ControlCycleBatch.cycle() -- per-arm port polling in Python, one fused call (cycle kernel +
tracking-error estimator + distance monitor on the device, one synchronisation), per-arm publishing -- against the array path
(step_arrays / Engine.step_host: no bottle anywhere) at B = 1, 64, 4 096 arms.  cycles/s = arm-cycles per second."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vfclik_amd import ports as yarp, robots  # noqa: E402
from vfclik_amd.vf_module import ControlCycleBatch  # noqa: E402


def bottle(values):
    b = yarp.Bottle()
    for v in values:
        b.addDouble(float(v))
    return b


chain = robots.lwr()
print("%6s %28s %28s %28s" % ("arms", "cycle() with ports", "step_arrays, all outputs", "step_arrays, qdot_out only"))
for B in (1, 64, 4096):
    bases = ["/%d/lwr/right" % i for i in range(B)]
    cb = ControlCycleBatch(chain, bases, io_dtype=np.float64, max_fields=8)
    rng = np.random.default_rng(1)
    q = rng.uniform(0.5 * chain.q_lo, 0.5 * chain.q_hi, (B, 7))
    goal = chain.fk(rng.uniform(0.5 * chain.q_lo, 0.5 * chain.q_hi, (B, 7))).reshape(B, 16)
    enc = []
    for a, base in enumerate(bases):
        p = yarp.BufferedPortBottle()
        p.open(base + "/test/param")
        yarp.Network.connect(base + "/test/param", base + "/vectorField/param")
        b = p.prepare(); b.clear(); b.addString("add"); b.addInt(1); b.addDouble(1.0); b.addInt(1)
        lst = b.addList()
        for v in list(goal[a]) + [0.05]:
            lst.addDouble(float(v))
        p.writeStrict()
        o = yarp.BufferedPortBottle()
        o.open(base + "/test/objects")
        yarp.Network.connect(base + "/test/objects", base + "/dmonitor/objectsIn")
        b = o.prepare(); b.clear(); b.addString("add"); b.addInt(0)
        lst = b.addList()
        for v in goal[a]:
            lst.addDouble(float(v))
        o.writeStrict()
        enc.append(cb.ports[a]["encoders"])
    n = 200 if B <= 64 else 12

    def feed():
        for a in range(B):
            b = enc[a].prepare(); b.clear()
            for v in q[a]:
                b.addDouble(float(v))
            enc[a].write()

    feed(); cb.cycle()
    t0 = time.perf_counter()
    for _ in range(n):
        feed()
        got = cb.cycle()
    t_ports = (time.perf_counter() - t0) / n
    assert got.all()
    full = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status", "goal_dist", "track_error", "obj_dist")
    n2 = 300
    cb.step_arrays(q, want=full)
    t0 = time.perf_counter()
    for _ in range(n2):
        cb.step_arrays(q, want=full)
    t_full = (time.perf_counter() - t0) / n2
    t0 = time.perf_counter()
    for _ in range(n2):
        cb.step_arrays(q)
    t_lean = (time.perf_counter() - t0) / n2
    print("%6d %14.1f us %9.3g c/s %14.1f us %9.3g c/s %14.1f us %9.3g c/s" % (B, t_ports * 1e6, B / t_ports, t_full * 1e6, B / t_full, t_lean * 1e6, B / t_lean), flush=True)
    cb.close()
