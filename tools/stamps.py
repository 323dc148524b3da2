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
COLD = "--cold" in sys.argv  # rotate over 34 input sets: the stamped launch finds its inputs in HBM, not in the Infinity Cache
robot, B, nobs, io, flags = {"C3": ("lwr", 65536, 8, np.float32, 0), "C5": ("lwr_dual14", 65536, 16, np.float32, 7),
                             "C2": ("lwr", 4096, 4, np.float64, 0), "C3N": ("lwr", 65536, 8, np.float32, 5),
                             "C3G": ("lwr", 65536, 8, np.float32, 0), "GAN": ("lwr", 65536, 5, np.float32, 0),
                             "C3D": ("lwr", 65536, 8, np.float64, 0),
                             "C3x2": ("lwr", 131072, 8, np.float32, 0),   # two waves per SIMD's worth of arms: the launch runs in rounds
                             # the full-output cycle (bench.py C3F / C5F): everything vf / nullspace / debug publish
                             "C3F": ("lwr", 65536, 8, np.float32, 5), "C5F": ("lwr_dual14", 65536, 16, np.float32, 7),
                             # ... and single extra outputs, to price them one by one
                             "C3N+qdist": ("lwr", 65536, 8, np.float32, 5), "C3N+pose": ("lwr", 65536, 8, np.float32, 5)}[wl]
EXTRA = {"C3F": ("pose", "pose_nt", "qdot_vf", "qdot_null", "qdist", "status"), "C5F": ("pose", "pose_nt", "qdot_vf", "qdot_null", "qdist", "status"),
         "C3N+qdist": ("qdist",), "C3N+pose": ("pose",)}.get(wl, ())
chain = robots.by_name(robot)
w = synth.make_workload(chain, B, nobs, seed=1, io_dtype=io, max_fields=8 if wl == "GAN" else None)
if wl == "GAN":  # goalAndNormal scene (object_feeder:248-303): attractor + funnel + near-goal repeller + 5 obstacles
    F = w["fields"]
    F["id"][:, 6], F["type"][:, 6], F["force"][:, 6] = 2, 5, 30.0
    F["p"][:, 6, 0:3] = F["p"][:, 0, [3, 7, 11]]
    F["p"][:, 6, 3:6] = F["p"][:, 0, [2, 6, 10]]
    F["p"][:, 6, 6:10] = [0.15, 10.0, 0.15, 2.0]
    F["id"][:, 7], F["type"][:, 7], F["force"][:, 7] = 3, 2, -10.0
    F["p"][:, 7, 0:3] = F["p"][:, 0, [3, 7, 11]] - 0.05 * F["p"][:, 0, [2, 6, 10]]
    F["p"][:, 7, 3:6] = [0.05, 0.001, 5.0]
    w["nfields"][:] = 8
    nobs = 10
if wl == "C3G":  # one arm with another decay order: the whole batch takes the general per-slot path
    w["fields"]["p"][B // 2, 1, 5] = 2.0
def one_set():
    e = engine.Engine(chain, B, io_dtype=io, max_slots=nobs, params=_abi.default_params(flags=flags))
    e.set_fields(w["fields"], w["nfields"])
    dq = e.dev_alloc(B * chain.n * np.dtype(io).itemsize)
    do = e.dev_alloc(B * chain.n * np.dtype(io).itemsize)
    e.h2d(dq, w["q"].astype(io))
    outs = {}
    for k in EXTRA:
        cols = {"pose": 16, "pose_nt": 16, "status": 1}.get(k, chain.n)
        outs[k] = e.dev_alloc(B * cols * 4)
    return e, e.make_io(dq, qdot_out=do, **outs)


eng, ioo = one_set()
others = [one_set() for _ in range(33)] if COLD else []
if COLD:  # all handles launch on ONE stream, in order, set 0 last: its launch is as cold as bench.py --state cold makes them
    import torch
    st_ = torch.cuda.current_stream().cuda_stream
    for e, _ in [(eng, ioo)] + others:
        e.use_stream(st_)
    for _ in range(3):
        for e, i_ in others:
            e.step(i_)
        eng.step(ioo)
    torch.cuda.synchronize()
    print("COLD state: 34 input sets round-robin, stamps of the last launch of set 0")
else:
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
# when do the waves start and end (10 ns resolution), and does it follow the workgroup index?
s0 = (st[:, 8].astype(np.int64) - int(st[:, 8].min())) / 100.0
e0 = (st[:, 9].astype(np.int64) - int(st[:, 8].min())) / 100.0
print("  wave START after the first wave [us]: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(s0, [10, 50, 90, 100])))
print("  wave END   after the first wave [us]: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(e0, [10, 50, 90, 100])))
idx = np.arange(nw)
print("  start vs workgroup index: corr %.2f; mean start of workgroups 0-127 %.2f, 448-575 %.2f, 896-1023 %.2f us"
      % (np.corrcoef(idx, s0)[0, 1], s0[:128].mean(), s0[448:576].mean(), s0[896:].mean()))
print("  mean start by (workgroup index mod 8): " + " ".join("%.2f" % s0[k::8].mean() for k in range(8)))
# the slowest waves: which phase is long, and where do they sit?
order = np.argsort(-tot)[:12]
print("  slowest waves (index, xcd = index mod 8, total ticks, phase ticks):")
for i in order:
    print("    %5d  xcd %d  %6d  %s" % (i, i % 8, tot[i], " ".join("%5d" % x for x in d[i])))
print("  p99 %d  p99.9 %d  max %d ticks; waves above median + 10%%: %d" % (np.percentile(tot, 99), np.percentile(tot, 99.9), tot.max(), int((tot > 1.1 * np.median(tot)).sum())))
late = np.argsort(-e0)[:12]
print("  last waves to end (index, start us, end us, duration ticks): " + "; ".join("%d %.2f %.2f %d" % (i, s0[i], e0[i], tot[i]) for i in late))
