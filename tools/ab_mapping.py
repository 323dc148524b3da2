#!/usr/bin/env python
"""Same-box A/B of the two arm-to-lane mappings on small lean batches: one lane per arm (cycle_kernel) against eight lanes
per arm (cycle_sub8_kernel).  HIP-event time per launch, median over rounds of 500 launches, per batch size.

    python tools/ab_mapping.py [--dtype float64] [--nobs 4]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float64")
    ap.add_argument("--nobs", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    import torch
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    dt = np.dtype(a.dtype).type
    tdt = torch.float32 if dt == np.float32 else torch.float64
    rows = []
    for B in (1, 64, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        w = synth.make_workload(chain, B, a.nobs, seed=3, io_dtype=dt)
        eng = engine.Engine(chain, B, io_dtype=dt, max_slots=max(1, a.nobs), params=_abi.default_params())
        eng.set_fields(w["fields"], w["nfields"])
        q = torch.from_numpy(w["q"].astype(dt)).cuda()
        out = torch.zeros(B, 7, dtype=tdt, device="cuda")
        eng.use_stream(torch.cuda.current_stream().cuda_stream)
        io = eng.make_io(q, qdot_out=out)
        res = {}
        for name, mb in (("lane", 0), ("sub8", 1 << 30)):
            eng.set_small_batch_kernel(mb)
            ts = []
            for _ in range(a.rounds):
                ts.append(eng.time_steps(io, 50, 500) * 1e3 / 500)
            res[name] = float(np.median(ts))
            res[name + "_out"] = out.cpu().numpy().copy()
        err = float(np.abs(res["lane_out"] - res["sub8_out"]).max())
        rows.append({"batch": B, "lane_us": res["lane"], "sub8_us": res["sub8"], "sub8_over_lane": res["sub8"] / res["lane"], "max_abs_diff": err})
        print("B %6d  lane-per-arm %.3f us   eight-lanes-per-arm %.3f us   ratio %.3f   max |diff| %.2e" % (B, res["lane"], res["sub8"], res["sub8"] / res["lane"], err), flush=True)
        eng.close()
    print(json.dumps({"dtype": a.dtype, "nobs": a.nobs, "rows": rows}))


if __name__ == "__main__":
    main()
