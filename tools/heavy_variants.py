#!/usr/bin/env python
"""Timing of the NON-LEAN single-cycle variants of the 14-joint chain (tool, float64 I/O, general field path): the kernels that are
built as an object of their own (csrc/Makefile, HEAVY).  65 536 arms, 16 obstacles, nullspace module + joint-limit task + mixer,
microseconds per launch; with VFIK_HIP_LIB set, of that library (same-box before / after).

    python tools/heavy_variants.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

B = 65536
chain = robots.lwr_dual14()
tool = np.eye(4)
tool[:3, 3] = [0.0, 0.0, 0.2]


def run(name, dt, flags, with_tool, general):
    w = synth.make_workload(chain, B, 16, seed=1, io_dtype=dt)
    if general:
        w["fields"]["p"][0, 12, 5] = 2.5   # one fractional decay order: the whole batch on the general path
    eng = engine.Engine(chain, B, io_dtype=dt, max_slots=16, params=_abi.default_params(flags=flags))
    eng.set_fields(w["fields"], w["nfields"])
    if with_tool:
        eng.set_tool(tool.reshape(16))
    es = np.dtype(dt).itemsize
    dq, do = eng.dev_alloc(B * 14 * es), eng.dev_alloc(B * 14 * es)
    eng.h2d(dq, w["q"].astype(dt))
    io = eng.make_io(dq, qdot_out=do)
    ms = eng.time_steps(io, 20, 200)
    print("%-64s %7.2f us per launch (field path %d)" % (name, ms * 1e3 / 200, eng.field_path))
    eng.close()


print("library: %s" % os.environ.get("VFIK_HIP_LIB", "in-tree"))
run("float32, nullspace, tool, general path", np.float32, 7, True, True)
run("float32, nullspace, tool, straight-line path", np.float32, 7, True, False)
run("float32, nullspace, no tool, general path (lean)", np.float32, 7, False, True)
run("float64, nullspace, no tool, general path", np.float64, 7, False, True)
run("float64, nullspace, no tool, straight-line path", np.float64, 7, False, False)
run("float64, nullspace, tool, straight-line path", np.float64, 7, True, False)
run("float64, nullspace, tool, general path", np.float64, 7, True, True)
run("float64, no module, tool, straight-line path", np.float64, 0, True, False)
run("float64, no module, tool, general path", np.float64, 0, True, True)
