import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from vfclik_amd import _abi, engine, robots, synth
B = 65536
chain = robots.lwr()
tool = np.eye(4); tool[:3, 3] = [0.0, 0.0, 0.2]
def run(name, flags, with_tool, wq=None):
    w = synth.make_workload(chain, B, 8, seed=1, io_dtype=np.float32)
    kw = {}
    if wq is not None: kw["wq"] = wq
    eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=_abi.default_params(flags=flags, **kw))
    eng.set_fields(w["fields"], w["nfields"])
    if with_tool: eng.set_tool(tool.reshape(16))
    dq, do = eng.dev_alloc(B * 7 * 4), eng.dev_alloc(B * 7 * 4)
    eng.h2d(dq, w["q"].astype(np.float32))
    io = eng.make_io(dq, qdot_out=do)
    ms = eng.time_steps(io, 20, 200)
    print("%-50s %7.2f us per launch (field path %d)" % (name, ms * 1e3 / 200, eng.field_path))
    eng.close()
run("C3 (no module), no tool", 0, False)
run("C3 (no module), tool 0 0 0.2", 0, True)
run("C3N (nullspace+mixer), no tool", 5, False)
run("C3N, tool 0 0 0.2", 5, True)
run("C3N, joint weights", 5, False, wq=[0.5]*7 + [1.0]*9)
run("C3N, joint weights and tool", 5, True, wq=[0.5]*7 + [1.0]*9)
