#!/usr/bin/env python
"""Benchmark of the batched control cycle (BASELINE.json metric: 7-DOF IK cycles/s at batch 65 536).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one control cycle of the whole batch = ONE launch of the fused kernel (vfik_step) over
synthetic inputs that are already resident in HBM.  Workload at every N: BASELINE config C3 per GPU
(65 536 arms x 7 joints, goal + 8 decay repellers, float32 I/O, float64 arithmetic) -- C4 is exactly
this with 8 ranks, so scaling is weak and there is no collective on the data path.

Rank 0 prints one JSON line.  `roofline` prices the kernel against HBM with the ALGORITHMIC bytes of
SURVEY 8d (384 B per cycle at C3); `cpu_baseline` times the CPU oracle (oracle/, the build's port of
the reference loop -- the reference itself cannot run) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (robot, batch per GPU, obstacles, io dtype, flags, algorithmic bytes per cycle [SURVEY 8d])
    "C2": ("lwr", 4096, 4, "float64", 0, 512),
    "C3": ("lwr", 65536, 8, "float32", 0, 384),
    "C5": ("lwr_dual14", 65536, 16, "float32", 1 | 2 | 4, 696),
    # C3 with the nullspace module and the mixer on, as `vfclik` starts them by default (vfclik:72-79); not a BASELINE config
    "C3N": ("lwr", 65536, 8, "float32", 1 | 4, 384),
    "C3D": ("lwr", 65536, 8, "float64", 0, 768),  # C3 with float64 I/O (not a BASELINE config)
}


def cpu_baseline(chain, params, w, budget_s=12.0):
    """CPU oracle (C, OpenMP over arms) on repeated passes over the same batch, ~budget_s seconds."""
    from oracle import oracle_c
    threads = oracle_c.max_threads()
    B = w["q"].shape[0]
    oracle_c.cycle_batch(chain, params, w["q"][:1024], w["fields"][:1024], w["nfields"][:1024], want=("qdot_out",))
    t0 = time.perf_counter()
    passes = 0
    while True:
        oracle_c.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out",))
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and passes >= 2:
            break
    return {"value": passes * B / dt, "unit": "cycles/s", "cores": threads, "kind": "port",
            "sample": "%d passes over the %d-arm batch in %.1f s, C oracle (oracle/vfik_oracle.c), OpenMP" % (passes, B, dt)}


def numpy_loop_rate(chain, params, w, arms=150):
    """The reference-style per-arm Python/NumPy loop (oracle/vfik_numpy.py), one process."""
    from oracle import vfik_numpy as vn
    from vfclik_amd import _abi
    pd = _abi.params_to_dict(params)
    cycles = []
    for b in range(arms):
        arm = vn.ArmCycle(chain.B, chain.jtype, chain.q_lo, chain.q_hi, pd)
        arm.set_fields({int(f["id"]): [float(f["force"]), int(f["type"]), f["p"][:_abi.FIELD_NPARAMS[int(f["type"])]].tolist()]
                        for f in w["fields"][b][: w["nfields"][b]]})
        cycles.append((arm, w["q"][b].tolist()))
    t0 = time.perf_counter()
    for arm, q in cycles:
        arm.cycle(q)
    return arms / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rollout", type=int, default=100, help="also time vfik_rollout with this many cycles per launch (0 = skip)")
    ap.add_argument("--host-path", type=int, default=100, help="also time this many steps with q/qdot in host memory (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the real thing; gloo + --single-device rehearses the N>1 control flow on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--gather", action="store_true", help="collate qdot of all ranks with one RCCL all_gather after the timed region")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from vfclik_amd import _abi, engine, robots, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    robot, B, nobs, io_name, flags, bytes_per_cycle = WORKLOADS[args.workload]
    io_dtype = np.dtype(io_name)
    chain = robots.by_name(robot)
    params = _abi.default_params(flags=flags)
    w = synth.make_workload(chain, B, nobs, seed=1 + rank, io_dtype=io_dtype.type)  # seeds 1.. for timing

    eng = engine.Engine(chain, B, io_dtype=io_dtype.type, max_slots=nobs, device=local_rank, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    tdt = torch.float32 if io_dtype == np.float32 else torch.float64
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where the timing reductions run
    q = torch.from_numpy(w["q"].astype(io_dtype)).to(dev)
    qdot = torch.zeros(B, chain.n, dtype=tdt, device=dev)
    stream = torch.cuda.current_stream()
    eng.use_stream(stream.cuda_stream)
    io = eng.make_io(q, qdot_out=qdot)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(io)
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        eng.step(io)
    enqueue_s = time.perf_counter() - t0  # host time to issue the launches (must stay below the kernel time)
    ev1.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)  # same stream as the launches
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # secondary figure (never `value`): closed-loop rollout, K control cycles per launch with q integrated in
    # registers -- what the cycle costs once the per-launch boundary is amortised
    rollout = None
    if args.rollout > 0:
        q_end = torch.empty_like(q)
        qd2 = torch.empty_like(qdot)
        io_r = eng.make_io(q, qdot_out=qd2)
        eng.rollout(io_r, args.rollout, 1e-3, q_out=q_end)
        sync_all()
        launches = 5
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(launches):
            eng.rollout(io_r, args.rollout, 1e-3, q_out=q_end)
        r1.record(stream)
        sync_all()
        r_ms = r0.elapsed_time(r1)
        if world > 1:
            t = torch.tensor([r_ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r_ms = float(t.item())
        rollout = {"cycles_per_launch": args.rollout, "launches": launches, "dt": 1e-3,
                   "us_per_cycle": r_ms * 1e3 / (launches * args.rollout),
                   "cycles_per_s": world * B * launches * args.rollout / (r_ms * 1e-3)}

    # secondary figure (never `value`): the PCIe-inclusive rate when q and qdot live in HOST memory, as they do
    # for a host that talks to robots -- synchronous pageable copies vs the three-stream pipeline on pinned buffers
    host_path = None
    if world == 1 and args.host_path > 0:
        hq = eng.host_array((B, chain.n))
        hq[:] = w["q"].astype(io_dtype)
        houts = [{"qdot_out": eng.host_array((B, chain.n))} for _ in range(3)]
        for k in range(3):
            eng.wait(eng.submit_host(hq, houts[k]))
        t1 = time.perf_counter()
        tickets = []
        for k in range(args.host_path):
            if len(tickets) == 3:
                eng.wait(tickets.pop(0))
            tickets.append(eng.submit_host(hq, houts[k % 3]))
        for t in tickets:
            eng.wait(t)
        piped = time.perf_counter() - t1
        qn = w["q"].astype(io_dtype)
        eng.step_host(qn)
        reps = max(3, args.host_path // 10)
        t1 = time.perf_counter()
        for _ in range(reps):
            eng.step_host(qn)
        synced = time.perf_counter() - t1
        host_path = {"pipelined_pinned_cycles_per_s": B * args.host_path / piped, "pipelined_us_per_step": piped * 1e6 / args.host_path,
                     "sync_pageable_cycles_per_s": B * reps / synced, "sync_us_per_step": synced * 1e6 / reps,
                     "bytes_per_step_each_way": int(B * chain.n * io_dtype.itemsize), "in_flight": 3}

    gathered = None
    if args.gather and world > 1:
        # optional collation of the per-rank results (NOT part of the control path): one all_gather
        src = qdot if args.dist_backend == "nccl" else qdot.cpu()
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        gathered = torch.cat(parts)

    if rank == 0:
        got = qdot.cpu().numpy().astype(np.float64)
        from oracle import oracle_c  # checker only: accuracy half of the metric + CPU baseline
        ref = oracle_c.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out",))
        max_err = float(np.abs(got - ref["qdot_out"]).max())
        us_per_launch = ev_ms * 1e3 / args.steps
        achieved = bytes_per_cycle * B / (us_per_launch * 1e-6) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "7-DOF IK cycles/sec (whole node), batch=65536; max |qdot-qdot_ref|",
            "value": world * B * args.steps / elapsed,
            "unit": "cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: batch=%d/GPU x %d-DOF %s, goal + %d decay repellers, %s I/O, flags=0x%x"
                                   % (args.workload, B, chain.n, chain.name, nobs, io_name, flags),
                       "parallelism": "arm batch sharded over %d GPU(s), no collective" % world,
                       "lambda": params.lambda_, "launches_per_step": 1,
                       "host_enqueue_us_per_step": enqueue_s * 1e6 / args.steps},
            "max_abs_err_rad_s": max_err,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # <io type, joints, nullspace module, PLAIN, rollout, straight-line field path, LEAN>: the bench
                         # workloads (revolute chain, identity tool, unit weights, integer-order repellers, qdot_out only)
                         "kernel": "vfik::cycle_kernel<%s,%d,%s,true,false,true,true>" % ("float" if io_name == "float32" else "double", chain.n,
                                                                                   "true" if flags & 1 else "false"),
                         "algorithmic_bytes_per_cycle": bytes_per_cycle, "us_per_launch_hip_events": us_per_launch},
        }
        if rollout is not None:
            line["rollout"] = rollout
        if host_path is not None:
            line["host_path"] = host_path
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(chain, params, w)
            line["cpu_baseline"]["numpy_ref_style_loop_cycles_per_s_1proc"] = numpy_loop_rate(chain, params, w)
        if gathered is not None:
            line["config"]["gathered_rows"] = int(gathered.shape[0])
        print(json.dumps(line), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
