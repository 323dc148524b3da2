#!/usr/bin/env python
"""Rates of the DROP-IN path: ControlCycleBatch.cycle() -- mail-driven port polling, one fused call (cycle kernel +
tracking-error estimator + distance monitor on the device, one synchronisation), publishing to the ports somebody reads --
against the array path (step_arrays / Engine.step_host: no bottle anywhere) at B = 1, 64, 4 096 arms.  Every arm has a goal
and a monitored object and gets fresh joint angles every cycle: (a) as ONE (B, n) array (write_encoders), nobody listening to
the outputs; (b) the same with a reader on every arm's /qdotOut and /pose (two bottles per arm and cycle are really read);
(c) as B bottles on /bridge/encoders, the reference's way.  cycles/s = arm-cycles per second."""
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
print("%6s %26s %26s %26s %26s %26s" % ("arms", "cycle(), (a) array in", "(b) + 2 readers per arm", "(c) per-arm bottles in", "step_arrays, all outputs", "step_arrays, qdot_out"))
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

    def timed(feeder, reps):
        feeder(); cb.cycle()
        t0 = time.perf_counter()
        for _ in range(reps):
            feeder()
            got = cb.cycle()
        assert got.all()
        return (time.perf_counter() - t0) / reps

    t_array = timed(lambda: cb.write_encoders(q), n)
    readers = []
    for base in bases:       # somebody reads two of every arm's outputs
        for name in ("/vectorField/qdotOut", "/vectorField/pose"):
            r = yarp.BufferedPortBottle()
            r.open(base + "/test/reader" + name)
            yarp.Network.connect(base + name, base + "/test/reader" + name)
            readers.append(r)

    def feed_and_read():
        cb.write_encoders(q)
    t_read = timed(feed_and_read, n)
    for r in readers:
        assert r.read(False).size() in (7, 16)
        r.close()
    t_ports = timed(feed, max(3, n // 4))
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
    print("%6d" % B + "".join(" %12.1f us %9.3g c/s" % (t * 1e6, B / t) for t in (t_array, t_read, t_ports, t_full, t_lean)), flush=True)
    cb.close()
