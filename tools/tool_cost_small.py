#!/usr/bin/env python
"""What a hand on the arm costs at vfclik's own sizes (diagnostic): 1 / 2 / 64 / 4 096 arms of the LWR, float64 I/O, nullspace module + mixer,
every per-cycle row published -- with and without `set tool 0 0 0.2` (old/README.old:84).  Microseconds per launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

chain = robots.lwr()
tool = np.eye(4)
tool[:3, 3] = [0.0, 0.0, 0.2]
for dt in (np.float64, np.float32):
    for B in (1, 2, 64, 4096):
        row = []
        for with_tool in (False, True):
            w = synth.make_workload(chain, B, 4, seed=1, io_dtype=dt)
            eng = engine.Engine(chain, B, io_dtype=dt, max_slots=8, params=_abi.default_params(flags=5))
            eng.set_fields(w["fields"], w["nfields"])
            if with_tool:
                eng.set_tool(tool.reshape(16))
            es = np.dtype(dt).itemsize
            bufs = {k: eng.dev_alloc(B * n * es) for k, n in (("q", 7), ("qdot_out", 7), ("qdot_vf", 7), ("qdot_null", 7), ("pose", 16), ("pose_nt", 16), ("qdist", 7))}
            eng.h2d(bufs["q"], w["q"].astype(dt))
            io = eng.make_io(bufs["q"], **{k: v for k, v in bufs.items() if k != "q"})
            ms = eng.time_steps(io, 20, 200)
            row.append(ms * 1e3 / 200)
            eng.close()
        print("%-8s %5d arms: %6.2f us without a tool, %6.2f us with" % (np.dtype(dt).name, B, row[0], row[1]))
