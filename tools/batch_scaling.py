#!/usr/bin/env python
"""Launch time and throughput of the lean C3 launch (7 joints, goal + 8 repellers, float32 I/O) over the batch size, beyond
one wave per SIMD (65 536 arms = 1 024 waves): the launch in rounds of one wave per SIMD (VFIK_TWO_WAVES=0), the two-waves-per-SIMD
build (the default beyond 1 024 waves) and the persistent launch (VFIK_PERSISTENT=1: one wave per SIMD striding over the chunks, next
chunk's inputs prefetched), same box, alternating.  `--flags 5` = the default process set."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--sizes", default="32768,65536,81920,98304,131072,196608,262144,393216,524288")
a = ap.parse_args()
chain = robots.lwr()
print("lean launch, 7 joints, goal + 8 repellers, float32 I/O, flags 0x%x; us per launch (median of 5 x 300 launches, HIP events)" % a.flags)
print("%8s %6s %12s %12s %12s %8s %14s %10s" % ("arms", "waves", "rounds us", "two waves", "persistent", "ratio", "cycles/s", "of 8 TB/s"))
for B in [int(x) for x in a.sizes.split(",")]:
    w = synth.make_workload(chain, B, 8, seed=3, io_dtype=np.float32)
    res = {}
    for pers in (0, 2, 1):   # 0 rounds, 2 two waves per SIMD, 1 persistent
        os.environ["VFIK_PERSISTENT"] = "1" if pers == 1 else "0"
        os.environ["VFIK_TWO_WAVES"] = "1" if pers == 2 else "0"
        eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params(flags=a.flags))
        eng.set_small_batch_kernel(0)
        eng.set_fields(w["fields"], w["nfields"])
        q = torch.from_numpy(w["q"].astype(np.float32)).cuda()
        out = torch.zeros(B, 7, dtype=torch.float32, device="cuda")
        eng.use_stream(torch.cuda.current_stream().cuda_stream)
        io = eng.make_io(q, qdot_out=out)
        ts = [eng.time_steps(io, 30, 300) * 1e3 / 300 for _ in range(5)]
        res[pers] = float(np.median(ts))
        eng.close()
    us = res[2]
    print("%8d %6d %12.3f %12.3f %12.3f %8.3f %14.3e %10.3f" % (B, (B + 63) // 64, res[0], res[2], res[1], res[2] / res[0], B / us * 1e6, 384 * B / us / 1e3 / 8000.0),
          flush=True)
