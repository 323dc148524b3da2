#!/usr/bin/env python
"""Latency of one control cycle at the batch sizes vfclik itself runs at (a handful of arms, scripts/vfclik:88-105) up to 4 096:
us per launch (HIP events around 500 back-to-back launches, median of 7) for
  (a) lean       q -> qdot_out, no module                                    (BASELINE C2's shape)
  (b) default    nullspace + mixer (vfclik:95-97), qdot_out only
  (c) full       nullspace + mixer, everything vf / nullspace / debug publish every cycle:
                 pose, pose_no_tool, qdotOut, qdotout, qdist, status         (scripts/vf:341-342,462-466; nullspace:180-184; debug_jointlimits:69-73)
with one lane per arm (cycle_kernel) and with eight lanes per arm (cycle_sub8_kernel) where that kernel serves the launch.
The floor under any of them is tools/ubench_launch: ~1.5 us for an empty kernel, ~3.0 us for one dependent load -> store."""
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
    ap.add_argument("--sync", action="store_true", help="one launch + one synchronisation per cycle (what a control loop with fresh joint angles every cycle does) instead of back-to-back launches")
    a = ap.parse_args()
    import torch
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    dt = np.dtype(a.dtype).type
    tdt = torch.float32 if dt == np.float32 else torch.float64
    rows = []
    print("7 joints, goal + %d repellers, %s I/O; us per %s" % (a.nobs, a.dtype, "cycle (launch + synchronisation, host clock)" if a.sync else "launch (back to back, HIP events)"))
    print("%6s %-8s %14s %16s %8s" % ("arms", "mode", "lane per arm", "8 lanes per arm", "ratio"))
    for B in (1, 8, 64, 512, 4096):
        w = synth.make_workload(chain, B, a.nobs, seed=3, io_dtype=dt)
        for mode, flags, full in (("lean", 0, False), ("default", 5, False), ("full", 5, True)):
            eng = engine.Engine(chain, B, io_dtype=dt, max_slots=max(1, a.nobs), params=_abi.default_params(flags=flags))
            eng.set_fields(w["fields"], w["nfields"])
            q = torch.from_numpy(w["q"].astype(dt)).cuda()
            outs = {"qdot_out": torch.zeros(B, 7, dtype=tdt, device="cuda")}
            if full:
                for k, c in (("pose", 16), ("pose_nt", 16), ("qdot_vf", 7), ("qdot_null", 7), ("qdist", 7)):
                    outs[k] = torch.zeros(B, c, dtype=tdt, device="cuda")
                outs["status"] = torch.zeros(B, dtype=torch.int32, device="cuda")
            eng.use_stream(torch.cuda.current_stream().cuda_stream)
            io = eng.make_io(q, **outs)
            res = {}
            for name, mb in (("lane", 0), ("sub8", 1 << 30)):
                eng.set_small_batch_kernel(mb)
                n0 = eng.small_batch_launches
                if a.sync:
                    import time
                    step = eng.stepper(io)
                    ts = []
                    for _ in range(a.rounds):
                        for _ in range(50):
                            step(); eng.sync()
                        t0 = time.perf_counter()
                        for _ in range(500):
                            step(); eng.sync()
                        ts.append((time.perf_counter() - t0) * 1e6 / 500)
                else:
                    ts = [eng.time_steps(io, 50, 500) * 1e3 / 500 for _ in range(a.rounds)]
                took = eng.small_batch_launches > n0
                res[name] = float(np.median(ts)) if (name == "lane" or took) else None
                res[name + "_out"] = outs["qdot_out"].cpu().numpy().copy()
            diff = float(np.abs(res["lane_out"] - res["sub8_out"]).max())
            rows.append({"batch": B, "mode": mode, "lane_us": res["lane"], "sub8_us": res["sub8"], "max_abs_diff": diff})
            print("%6d %-8s %14.3f %16s %8s" % (B, mode, res["lane"], "%.3f" % res["sub8"] if res["sub8"] else "(not served)",
                                              "%.3f" % (res["sub8"] / res["lane"]) if res["sub8"] else "-"), flush=True)
            eng.close()
    print(json.dumps({"dtype": a.dtype, "nobs": a.nobs, "rows": rows}))


if __name__ == "__main__":
    main()
