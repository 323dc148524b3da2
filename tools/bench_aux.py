#!/usr/bin/env python
"""Throughput of the small kernels beside the control cycle (mixer sum, tracking-error estimator, distance
monitor, field probe) at the C3 batch size, against their algorithmic bytes."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (HIP events)
from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

B, O, REP = 65536, 8, 200
chain = robots.lwr()
w = synth.make_workload(chain, B, 8, seed=1, io_dtype=np.float32)
eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params())
eng.set_fields(w["fields"], w["nfields"])
stream = torch.cuda.current_stream()
eng.use_stream(stream.cuda_stream)
dev = torch.device("cuda", 0)
pose = torch.from_numpy(chain.fk(w["q"]).reshape(B, 16).astype(np.float32)).to(dev)
v6 = torch.randn(B, 6, device=dev)
frames = torch.from_numpy(np.tile(np.eye(4, dtype=np.float32).reshape(16), (B, O, 1))).to(dev)
cmds = torch.randn(6, B, 7, device=dev)
out8, outd, outv, outm = (torch.empty(B, 8, device=dev), torch.empty(B, O, 2, device=dev), torch.empty(B, 6, device=dev),
                          torch.empty(B, 7, device=dev))
wts = np.array([1, 1, 0.5, 0, 0, 0.25])
cases = {
    "mix_kernel (vfik_mix, K=6)": (lambda: eng.mix(cmds, wts, outm), B * 7 * (6 + 1) * 4),
    "track_kernel (vfik_track_error)": (lambda: eng.track_error(pose, v6, out8), B * ((16 + 6 + 8) * 4 + 2 * 38 * 8)),
    "monitor_kernel (8 objects)": (lambda: eng.object_distances(pose, frames, O, outd), B * O * (32 + 2) * 4),
    "probe_kernel (goal + 8 repellers)": (lambda: eng.probe_field(pose, outv), B * (16 + 16 + 64 + 6) * 4),
}
res = {}
for name, (fn, nbytes) in cases.items():
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(REP):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REP
    res[name] = {"us_per_launch": us, "algorithmic_bytes": nbytes, "GBps": nbytes / us / 1e3, "frac_of_8TBps": nbytes / us / 1e3 / 8000}
    print("%-36s %7.2f us  %8.1f GB/s  (%.3f of 8 TB/s, %.2f MB)" % (name, us, res[name]["GBps"], res[name]["frac_of_8TBps"], nbytes / 1e6))
print(json.dumps(res))
eng.close()
