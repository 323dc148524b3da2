#!/usr/bin/env python
"""Launch time and throughput of the lean C3 launch (7 joints, goal + 8 repellers, float32 I/O) over the batch size: what a
second wave per SIMD is worth once the batch exceeds one wave per SIMD (65 536 arms = 1 024 waves)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

chain = robots.lwr()
for B in (16384, 32768, 65536, 81920, 98304, 131072, 196608, 262144, 524288):
    w = synth.make_workload(chain, B, 8, seed=3, io_dtype=np.float32)
    eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params())
    eng.set_small_batch_kernel(0)
    eng.set_fields(w["fields"], w["nfields"])
    q = torch.from_numpy(w["q"].astype(np.float32)).cuda()
    out = torch.zeros(B, 7, dtype=torch.float32, device="cuda")
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    io = eng.make_io(q, qdot_out=out)
    ts = [eng.time_steps(io, 30, 300) * 1e3 / 300 for _ in range(5)]
    us = float(np.median(ts))
    print("B %7d  waves %5d  %.3f us per launch  %.3e cycles/s  %.3f of the HBM roofline (384 B per cycle)" % (B, (B + 63) // 64, us, B / us * 1e6, 384 * B / us / 1e3 / 8000.0), flush=True)
    eng.close()
