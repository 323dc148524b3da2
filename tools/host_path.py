#!/usr/bin/env python
"""Host-memory path timing: synchronous vfik_step_host vs the vfik_submit_host pipeline (diagnostic)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

B, N = 65536, 200
chain = robots.lwr()
w = synth.make_workload(chain, B, 8, seed=1, io_dtype=np.float32)
eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params())
eng.set_fields(w["fields"], w["nfields"])
if len(sys.argv) > 1 and sys.argv[1] == "null":
    eng.use_stream(0)
hq = eng.host_array((B, 7))
hq[:] = w["q"]
houts = [{"qdot_out": eng.host_array((B, 7))} for _ in range(3)]
for k in range(3):
    eng.wait(eng.submit_host(hq, houts[k]))
for depth in (1, 2, 3):  # VFIK_HOST_HYBRID=1 switches the input side to the copy engine
    t_sub = t_wait = 0.0
    t0 = time.perf_counter()
    tickets = []
    for k in range(N):
        if len(tickets) == depth:
            a = time.perf_counter()
            eng.wait(tickets.pop(0))
            t_wait += time.perf_counter() - a
        a = time.perf_counter()
        tickets.append(eng.submit_host(hq, houts[k % 3]))
        t_sub += time.perf_counter() - a
    for t in tickets:
        eng.wait(t)
    dt = time.perf_counter() - t0
    print("depth %d: %.1f us/step  (submit %.1f us, wait %.1f us per step)  %.3g cycles/s" % (depth, dt * 1e6 / N, t_sub * 1e6 / N, t_wait * 1e6 / N, B * N / dt))
qn = w["q"].astype(np.float32)
eng.step_host(qn)
t0 = time.perf_counter()
for _ in range(30):
    eng.step_host(qn)
print("sync pageable: %.1f us/step" % ((time.perf_counter() - t0) * 1e6 / 30))
pq = eng.host_array((B, 7)); pq[:] = qn
t0 = time.perf_counter()
for _ in range(30):
    eng.step_host(pq)
print("sync, pinned q in (pageable out): %.1f us/step" % ((time.perf_counter() - t0) * 1e6 / 30))
eng.close()

# zero-copy: the kernel reads q from / writes qdot to pinned host memory itself (no copy engine)
eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params())
eng.set_fields(w["fields"], w["nfields"])
zq = eng.host_array((B, 7)); zq[:] = w["q"]
zo = eng.host_array((B, 7))
io = eng.make_io(zq.ctypes.data, qdot_out=zo.ctypes.data)
eng.step(io); eng.sync()
ref = eng.step_host(w["q"].astype(np.float32))["qdot_out"]
print("zero-copy equals copy path:", np.array_equal(ref, zo))
t0 = time.perf_counter()
for _ in range(N):
    eng.step(io); eng.sync()
print("zero-copy, sync each step: %.1f us/step" % ((time.perf_counter() - t0) * 1e6 / N))
t0 = time.perf_counter()
for _ in range(N):
    eng.step(io)
eng.sync()
print("zero-copy, back to back: %.1f us/step" % ((time.perf_counter() - t0) * 1e6 / N))
eng.close()

# which direction costs what: zero-copy read only / write only
eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params())
eng.set_fields(w["fields"], w["nfields"])
zq = eng.host_array((B, 7)); zq[:] = w["q"]
zo = eng.host_array((B, 7))
dq = eng.dev_alloc(B * 7 * 4); do = eng.dev_alloc(B * 7 * 4)
eng.h2d(dq, w["q"].astype(np.float32))
for name, io in (("read q over PCIe, write qdot to HBM", eng.make_io(zq.ctypes.data, qdot_out=do)),
                 ("read q from HBM, write qdot over PCIe", eng.make_io(dq, qdot_out=zo.ctypes.data)),
                 ("both in HBM", eng.make_io(dq, qdot_out=do))):
    eng.step(io); eng.sync()
    t0 = time.perf_counter()
    for _ in range(N):
        eng.step(io)
    eng.sync()
    print("%-40s %.1f us/step" % (name, (time.perf_counter() - t0) * 1e6 / N))
eng.close()
