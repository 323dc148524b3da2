#!/usr/bin/env python
"""Cost of the general per-slot field path against the straight-line repeller path (diagnostic):
the C3 workload with ONE arm given a different decay order, which sends the whole batch down the general path;
and a 'goalAndNormal' scene (attractor + funnel + near-goal repeller + 5 obstacles, object_feeder:248-303)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vfclik_amd import _abi, engine, robots, synth  # noqa: E402

B = 65536
chain = robots.lwr()


def run(name, w, slots):
    eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=slots, params=_abi.default_params())
    eng.set_fields(w["fields"], w["nfields"])
    dq = eng.dev_alloc(B * 7 * 4)
    do = eng.dev_alloc(B * 7 * 4)
    eng.h2d(dq, w["q"].astype(np.float32))
    io = eng.make_io(dq, qdot_out=do)
    ms = eng.time_steps(io, 20, 200)
    print("%-44s %.2f us per step (slots in use %d, field path %d%s)" % (name, ms * 1e3 / 200, eng.lib.vfik_slots_in_use(eng.h), eng.field_path,
                                                                             ", mixed orders" if eng.mixed_orders else ""))
    eng.close()


w = synth.make_workload(chain, B, 8, seed=1, io_dtype=np.float32)
run("C3, straight-line path", w, 8)
w["fields"]["p"][0, 1, 5] = 2.0
run("C3, one arm with another order", w, 8)
w["fields"]["p"][:, 1:9, 5] = [5, 20, 20, 5, 2, 20, 20, 3]
run("C3, orders that differ by obstacle", w, 8)
w["fields"]["p"][:, 1:9, 5] = np.random.default_rng(0).integers(1, 21, (B, 8))
run("C3, every arm its own random orders", w, 8)
w = synth.make_workload(chain, B, 5, seed=1, io_dtype=np.float32, max_fields=8)
F = w["fields"]
F["id"][:, 6], F["type"][:, 6], F["force"][:, 6] = 2, 5, 30.0      # funnel at the goal along its z axis
F["p"][:, 6, 0:3] = F["p"][:, 0, [3, 7, 11]]
F["p"][:, 6, 3:6] = F["p"][:, 0, [2, 6, 10]]
F["p"][:, 6, 6:10] = [0.15, 10.0, 0.15, 2.0]
F["id"][:, 7], F["type"][:, 7], F["force"][:, 7] = 3, 2, -10.0     # near-goal repeller
F["p"][:, 7, 0:3] = F["p"][:, 0, [3, 7, 11]] - 0.05 * F["p"][:, 0, [2, 6, 10]]
F["p"][:, 7, 3:6] = [0.05, 0.001, 5.0]
w["nfields"][:] = 8
run("goalAndNormal + 5 obstacles", w, 10)
F["p"][:, 1:6, 5] = 20.0    # old/README.old:75: `ObstacleP ... 0.05 20` beside the feeder's order-5 near-goal repeller
F["p"][:, 1:6, 3] = 0.05
run("README scene: goalAndNormal + 5 x order 20", w, 10)
F["p"][:, 1:6, 5] = 5.0
# the same scene over a table (ObstacleH, object_feeder:344-353): one hemisphere repeller more, in place of the fifth obstacle
F["id"][:, 5], F["type"][:, 5], F["force"][:, 5] = 40, 4, -50.0
F["p"][:, 5] = 0.0
F["p"][:, 5, 0:8] = [0.0, 0.0, -0.3, 0.02, -0.01, 1.0, 0.05, 5.0]
run("goalAndNormal + 4 obst. + table", w, 11)
F["p"][0, 5, 7] = 4.5      # one arm's table with a fractional order: the whole batch on the general path
run("... on the general path", w, 11)
chain = robots.lwr_dual14()
w = synth.make_workload(chain, B, 16, seed=1, io_dtype=np.float32)


def run14(name, w):
    eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=16, params=_abi.default_params(flags=7))
    eng.set_fields(w["fields"], w["nfields"])
    dq = eng.dev_alloc(B * 14 * 4)
    do = eng.dev_alloc(B * 14 * 4)
    eng.h2d(dq, w["q"].astype(np.float32))
    io = eng.make_io(dq, qdot_out=do)
    ms = eng.time_steps(io, 20, 200)
    print("%-44s %.2f us per step (field path %d%s)" % (name, ms * 1e3 / 200, eng.field_path, ", mixed orders" if eng.mixed_orders else ""))
    eng.close()


run14("C5, straight-line path", w)
w["fields"]["p"][0, 12, 5] = 2.0
run14("C5, one arm with another order", w)
w["fields"]["p"][:, 1:17, 5] = [5, 20, 20, 5, 2, 20, 20, 3, 5, 5, 20, 20, 20, 20, 5, 5]
run14("C5, orders that differ by obstacle", w)
