import sys, numpy as np
sys.path.insert(0, ".")
import torch
from oracle import oracle_c
from vfclik_amd import _abi, engine, robots, synth
chain = robots.lwr()
for B, nobs in ((1000, 6), (1000, 8), (65536, 8), (65536, 4), (4096, 8)):
    w = synth.make_workload(chain, B, nobs, seed=2, io_dtype=np.float64)
    params = _abi.default_params()
    eng = engine.Engine(chain, B, io_dtype=np.float64, max_slots=nobs, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    ref = oracle_c.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out",))["qdot_out"]
    a = eng.step_host(w["q"], want=("qdot_out",))["qdot_out"]
    b = eng.step_host(w["q"], want=("qdot_out", "status"))["qdot_out"]
    q = torch.from_numpy(w["q"]).cuda(); o = torch.zeros(B, 7, dtype=torch.float64, device="cuda")
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    io = eng.make_io(q, qdot_out=o)
    for _ in range(3): eng.step(io)
    torch.cuda.synchronize()
    c = o.cpu().numpy()
    e = np.abs(c - ref).max(axis=1)
    print(B, nobs, "lean host %.2e  full host %.2e  lean dev %.2e  bad arms %d first %s" % (np.abs(a - ref).max(), np.abs(b - ref).max(), e.max(), (e > 1e-6).sum(), np.nonzero(e > 1e-6)[0][:8]))
    eng.close()
