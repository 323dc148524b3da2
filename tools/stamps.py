#!/usr/bin/env python
"""Where a wave spends its cycles: runs the diagnostic build (make -C vfclik_amd/csrc
libvfik_hip_stamps.so) on a workload and prints per-section s_memtime differences (median over
waves).  Diagnostic only -- never quote this build's run time."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VFIK_HIP_LIB"] = os.path.join(ROOT, "vfclik_amd", "csrc", "libvfik_hip_stamps.so")
from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
robot, B, nobs, io, flags = {"C3": ("lwr", 65536, 8, np.float32, 0), "C5": ("lwr_dual14", 65536, 16, np.float32, 7),
                             "C2": ("lwr", 4096, 4, np.float64, 0)}[wl]
chain = robots.by_name(robot)
w = synth.make_workload(chain, B, nobs, seed=1, io_dtype=io)
eng = engine.Engine(chain, B, io_dtype=io, max_slots=nobs, params=_abi.default_params(flags=flags))
eng.set_fields(w["fields"], w["nfields"])
dq = eng.dev_alloc(B * chain.n * np.dtype(io).itemsize)
do = eng.dev_alloc(B * chain.n * np.dtype(io).itemsize)
eng.h2d(dq, w["q"].astype(io))
ioo = eng.make_io(dq, qdot_out=do)
for _ in range(5):
    eng.step(ioo)
eng.sync()
nw = (B + 63) // 64
st = np.zeros((nw, 10), dtype=np.uint64)
eng.lib.vfik_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert eng.lib.vfik_debug_read_stamps(eng.h, st.ctypes.data) == 0
d = np.diff(st[:, :8].astype(np.int64), axis=1)
names = ["issue loads", "wait for q", "sincos+FK+J", "tool+goal attractor", "slots", "normCart+RefPt+IK", "nullspace+mixer+stores"]
print("workload", wl, "waves", nw)
for i in range(7):
    print("  %-22s median %7.0f  p10 %7.0f  p90 %7.0f ticks" % (names[i], np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
tot = (st[:, 7] - st[:, 0]).astype(np.int64)
print("  %-16s median %7.0f  p10 %7.0f  p90 %7.0f ticks" % ("whole wave", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
rt = (st[:, 9] - st[:, 8]).astype(np.int64)  # s_memrealtime ticks (100 MHz) over the same interval
ok = rt > 0
print("  shader clock from s_memtime / s_memrealtime: %.3f GHz (median over waves)" % np.median(tot[ok] / rt[ok] * 0.1))
print("  kernel span by s_memrealtime: first wave start -> last wave end %.2f us" % ((int(st[:, 9].max()) - int(st[:, 8].min())) / 100.0))
span = int(st[:, 7].max() - st[:, 0].min())
print("  first start -> last end: %d ticks; wave start spread %d ticks" % (span, int(st[:, 0].max() - st[:, 0].min())))
ms = eng.time_steps(ioo, 5, 50)
print("  diagnostic build: %.2f us per launch (do not quote)" % (ms * 1e3 / 50))
