#!/usr/bin/env python
"""Benchmark of the batched control cycle (BASELINE.json metric: 7-DOF IK cycles/s at batch 65 536).

    python bench.py --gpus N --steps K --warmup W            (N > 1: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one control cycle of the whole batch = ONE launch of the fused kernel (vfik_step) over
synthetic inputs that are already resident in HBM.  Workload at every N: BASELINE config C3 per GPU
(65 536 arms x 7 joints, goal + 8 decay repellers, float32 I/O, float64 arithmetic) -- C4 is exactly
this with 8 ranks, so scaling is weak and there is no collective on the data path.

Timing (median and p10/p90): the timed region -- barrier, synchronize, stamp, EXACTLY K
launches, synchronize, stamp -- is repeated R times (--reps); the collective that takes the maximum over
ranks runs after the last repetition, outside every stamped interval.  `value` comes from the MEDIAN
repetition; p10/p90 are printed next to it.  R further repetitions, interleaved with those, bracket the same K
launches with HIP events on the launch stream and give the kernel's launch period for `roofline`.

Rank 0 prints one JSON line.  `roofline` prices the kernel against HBM with the ALGORITHMIC bytes of
SURVEY 8d (384 B per cycle at C3); `cpu_baseline` (N = 1 only) times the CPU oracle (oracle/, the build's
port of the reference loop -- the reference itself cannot run) on this box's host cores, BEFORE
this process touches the GPU: the reference-style per-arm NumPy loop as one process per core over disjoint
arm slices (kind "port"), and the C/OpenMP port as `best_cpu`.
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

# What vf / nullspace / debug_jointlimits publish every cycle besides the mixed command (vf:341-342,462-466;
# nullspace:180-184; debug_jointlimits:69-73), in scalars per arm: pose 16, pose_no_tool 16, qdotOut n, qdotout n, qdist n, status 1
FULL_OUTS = ("pose", "pose_nt", "qdot_vf", "qdot_null", "qdist", "status")

WORKLOADS = {
    # name: (robot, batch per GPU, obstacles, io dtype, flags, algorithmic bytes per cycle [SURVEY 8d], extra outputs)
    "C2": ("lwr", 4096, 4, "float64", 0, 512, ()),
    "C3": ("lwr", 65536, 8, "float32", 0, 384, ()),
    "C5": ("lwr_dual14", 65536, 16, "float32", 1 | 2 | 4, 696, ()),
    # C3 with the nullspace module and the mixer on, as `vfclik` starts them by default (vfclik:72-79); not a BASELINE config
    "C3N": ("lwr", 65536, 8, "float32", 1 | 4, 384, ()),
    "C3D": ("lwr", 65536, 8, "float64", 0, 768, ()),  # C3 with float64 I/O (not a BASELINE config)
    # the reference-faithful cycle: C3N / C5 publishing everything the per-arm processes publish (SURVEY 8d "optional outputs":
    # + (16 + 16 + 3 n + 1) scalars): 384 + 54 x 4 = 600 B, 696 + 75 x 4 = 996 B
    "C3F": ("lwr", 65536, 8, "float32", 1 | 4, 600, FULL_OUTS),
    "C5F": ("lwr_dual14", 65536, 16, "float32", 1 | 2 | 4, 996, FULL_OUTS),
    # C2's batch with vfclik's default process set publishing every per-cycle row: the eight-lanes-per-arm kernel <double, 7, NS>
    "C2F": ("lwr", 4096, 4, "float64", 1 | 4, 512 + 54 * 8, FULL_OUTS),
}

MIB = 1 << 20


def host_cores(cap=16):
    """Worker processes / threads for the CPU baseline: the cores this process may use -- its cpuset, the
    cgroup's CPU quota when one is set -- and at most `cap`: a 1-GPU box is a 16-core share of a host whose
    other cores (os.cpu_count() reports them all) belong to other tenants; oversubscribing them measured
    5e5 cycles/s for the C port on 256 threads against 8.5e6 on an idle 128."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, (quota + period // 2) // period))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap) if cap else n)


# ---------------------------------------------------------------------------------------------------
# CPU baseline (checker code timed as a baseline: the only place besides tests/ and smoke() that uses oracle/)
# ---------------------------------------------------------------------------------------------------
def _numpy_slice_worker(args):
    """One process of the reference-style loop: one arm per Python iteration, exactly like the loop of
    scripts/vf:193-521 minus sleeps and ports (oracle/vfik_numpy.ArmCycle), over its own slice of arms,
    pass after pass until the budget is used up."""
    chain, pd, q, fields, nfields, budget_s = args
    from oracle import vfik_numpy as vn
    from vfclik_amd import _abi
    try:  # one core per process: the 6 x 6 solves must not start a BLAS thread pool each
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except ImportError:
        pass
    arms = []
    for b in range(q.shape[0]):
        arm = vn.ArmCycle(chain.B, chain.jtype, chain.q_lo, chain.q_hi, pd)
        arm.set_fields({int(f["id"]): [float(f["force"]), int(f["type"]), f["p"][:_abi.FIELD_NPARAMS[int(f["type"])]].tolist()]
                        for f in fields[b][: nfields[b]]})
        arms.append((arm, q[b].tolist()))
    cycles = 0
    t0 = time.perf_counter()
    while True:
        for arm, qb in arms:
            arm.cycle(qb)
        cycles += len(arms)
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            return cycles, dt


def numpy_loop_node_rate(chain, params, w, procs, arms_per_proc=64, budget_s=6.0):
    """SURVEY 8d CPU baseline (1): `procs` independent processes over disjoint arm slices -> whole-node
    cycles/s = all cycles / the slowest process's time.  Runs before the GPU is initialised (fork is safe)."""
    import multiprocessing as mp
    from vfclik_amd import _abi
    pd = _abi.params_to_dict(params)
    B = w["q"].shape[0]
    per = max(1, min(arms_per_proc, B // procs))
    jobs = [(chain, pd, w["q"][p * per:(p + 1) * per], w["fields"][p * per:(p + 1) * per], w["nfields"][p * per:(p + 1) * per], budget_s)
            for p in range(procs)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_numpy_slice_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    cycles = sum(c for c, _ in res)
    slowest = max(t for _, t in res)
    return {"value": cycles / slowest, "unit": "cycles/s", "cores": procs, "kind": "port",
            "sample": "reference-style per-arm NumPy loop (oracle/vfik_numpy.ArmCycle = scripts/vf:193-521 minus sleeps and ports): "
                      "%d processes x %d arms each (disjoint slices of the bench batch), repeated passes for %.1f s; %d cycles, wall %.1f s incl. process start"
                      % (procs, per, slowest, cycles, wall),
            "per_process_cycles_per_s": cycles / slowest / procs}


def c_oracle_rate(chain, params, w, threads, budget_s=8.0):
    """CPU baseline (2), "best CPU": the C port of the oracle, OpenMP over arms, repeated passes over the batch."""
    from oracle import oracle_c
    threads = max(1, min(threads, oracle_c.max_threads()))
    B = w["q"].shape[0]
    oracle_c.cycle_batch(chain, params, w["q"][:1024], w["fields"][:1024], w["nfields"][:1024], want=("qdot_out",), nthreads=threads)
    t0 = time.perf_counter()
    passes = 0
    while True:
        oracle_c.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out",), nthreads=threads)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and passes >= 2:
            break
    return {"value": passes * B / dt, "unit": "cycles/s", "cores": threads, "kind": "port",
            "sample": "%d passes over the %d-arm batch in %.1f s, C oracle (oracle/vfik_oracle.c), OpenMP" % (passes, B, dt)}


def kernel_name(io_name, n, flags, batch, sub8, full=False, uni=True, dhp=0):
    """The kernel a lean bench launch takes (vfik_kernel.hip, launch_v) -- as rocprofv3 names it, without spaces.
    <io type, joints, nullspace module, PLAIN, rollout, straight-line field path, LEAN, compile-time flags, persistent, aux block,
    waves per SIMD, uniform repeller image, [order planes,] DH pattern>: the bench workloads are revolute chains with identity tool,
    unit weights, integer-order repellers and qdot_out only."""
    t = "float" if io_name == "float32" else "double"
    ns = "true" if flags & 1 else "false"
    if sub8:
        return "vfik::cycle_sub8_kernel%s<%s,%d,%s,%d>" % ("" if flags & 1 else "_x", t, n, ns, dhp)
    d = dhp if (t == "float" or n == 7) else 0   # (the lane-per-arm pattern variants: float32 I/O; float64 I/O for the 7-joint chain)
    u = "true" if uni else "false"
    cf = flags if (flags & 1 and n <= 7 and flags in (5, 7)) else -1
    if full:  # the per-cycle rows asked for, no per-arm option: the publishing lean variant (LEAN 3)
        return "vfik::cycle_kernel_x<%s,%d,%s,true,false,true,3,%d,false,false,1,%s,false,%d>" % (t, n, ns, cf, u, d)
    # (the lean variants take their ten arguments as scalars, preloaded into SGPRs: cycle_kernel_s)
    return "vfik::cycle_kernel_s<%s,%d,%s,true,false,true,1,%d,false,false,1,%s,%d>" % (t, n, ns, cf, u, d)


def pctl(xs, p):
    xs = sorted(xs)
    if len(xs) == 1:
        return xs[0]
    k = (len(xs) - 1) * p / 100.0
    lo = int(k)
    hi = min(lo + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=30, help="repetitions of the timed region (K steps each); value = median repetition")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--secondary", type=int, default=1, help="C3 at N = 1: also time the batch with per-arm (safe distance, force) pairs and with "
                    "integer orders that differ (never `value`); 0: skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes / threads of the CPU baseline (0 = this box's CPU share, at most 16)")
    ap.add_argument("--rollout", type=int, default=100, help="also time vfik_rollout with this many cycles per launch (0 = skip)")
    ap.add_argument("--host-path", type=int, default=100, help="also time this many steps with q/qdot in host memory (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the real thing; gloo + --single-device rehearses the N>1 control flow on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--gather", action="store_true", help="collate qdot of all ranks with one RCCL all_gather after the timed region")
    ap.add_argument("--rotate", type=int, default=-1,
                    help="input sets of the COLD measurement: that many independent handles + q / qdot buffers launched round-robin, so that "
                         "no byte survives in the 256 MiB Infinity Cache between two uses (-1 = as many as make the bytes touched between "
                         "two uses of a set exceed 640 MiB, at least 24; 0 = skip)")
    ap.add_argument("--state", default="both", choices=["both", "warm", "cold"],
                    help="both: `value` and `roofline` from back-to-back launches of ONE input set (cache-resident), `roofline.cold` from the "
                         "rotating sets; cold: every launch of the run rotates (what a rocprofv3 trace of the cold state needs); warm: no rotation")
    ap.add_argument("--launch", default="auto", choices=["auto", "direct", "graph"],
                    help="how the K launches of a timed region reach the GPU: direct = K vfik_step calls; graph = one replay of a hipGraph that "
                         "captured those K vfik_step calls (the host then costs ~0.3 us per launch instead of 3-5); auto = whichever the warm-up "
                         "measures faster on this box (graph only when it wins by > 3 %: it pays when the enqueue loop, not the GPU, sets the pace)")
    ap.add_argument("--dump-reps", default=None, help="diagnostic: write the per-repetition times (us per launch, HIP events) to this file")
    ap.add_argument("--sync-each", action="store_true",
                    help="diagnostic: synchronize after every launch (un-overlapped kernel durations for a rocprofv3 kernel trace); "
                         "the line is then not a throughput measurement")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1 or args.steps < 1 or args.reps < 1 or args.warmup < 0:
        raise SystemExit("--gpus, --steps, --reps must be >= 1 and --warmup >= 0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no external launcher: this process becomes the parent of one fresh process per GPU.  It has made
        # no GPU call (torch is not even imported yet) and makes none afterwards.
        from vfclik_amd import launcher
        raise SystemExit(launcher.main_spawn(os.path.abspath(__file__), sys.argv[1:] if argv is None else argv, args.gpus))
    return worker(args)


def worker(args):
    from vfclik_amd import _abi, robots, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    robot, B, nobs, io_name, flags, bytes_per_cycle, extra_outs = WORKLOADS[args.workload]
    io_dtype = np.dtype(io_name)
    chain = robots.by_name(robot)
    params = _abi.default_params(flags=flags)
    w = synth.make_workload(chain, B, nobs, seed=1 + rank, io_dtype=io_dtype.type)  # seeds 1.. for timing

    # CPU baseline first, while this process has not initialised the GPU (rank 0 at N = 1 only)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cores = args.cpu_procs if args.cpu_procs > 0 else host_cores()
        cpu = numpy_loop_node_rate(chain, params, w, procs=cores)
        cpu["os_cpu_count"] = os.cpu_count()
        cpu["best_cpu"] = c_oracle_rate(chain, params, w, cores)

    import torch
    import torch.distributed as dist

    # each rank on its own cores (near its GPU's NUMA node where sysfs tells), before the first GPU call: the HIP runtime's threads
    # inherit the mask; printed on stderr.  (device_count() initialises nothing.)
    binding = None
    if world > 1:
        from vfclik_amd import launcher
        try:
            binding = launcher.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), visible_gpus=torch.cuda.device_count())
        except Exception as e:  # noqa: BLE001 -- pinning is an optimisation: a host whose sysfs reads differently must not cost the run
            print("[bench] rank %d: not pinned (%s)" % (rank, e), file=sys.stderr)

    from vfclik_amd import engine, sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = args.dist_backend
    backend_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The control path has no collective: the process group only carries the barrier in front of each timed region, the max over
        # ranks of the timings and the optional gather.  One store for the job; RCCL is initialised eagerly (device_id), so that a
        # rank that cannot bring its communicator up says so HERE -- then EVERY rank falls back to gloo for those three (decided
        # through the store), instead of the job dying without a line.
        # (the store of the env:// rendezvous: under torch.distributed.run the launcher's agent hosts it, under bench.py's own
        # launcher rank 0 does)
        store, _, _ = next(dist.rendezvous("env://", rank=rank, world_size=world))
        if backend == "nccl":
            from vfclik_amd import launcher
            import datetime
            bound = float(os.environ.get("VFIK_BENCH_PG_TIMEOUT", "120"))
            # (1) agree BEFORE anyone enters the collective communicator start (launcher.agree_on_backend): a rank whose local
            # pre-check fails -- RCCL missing, its device unusable, a device it shares with another rank -- takes everybody to gloo
            ok, why, dev_id = True, "", ""
            try:
                forced = os.environ.get("VFIK_BENCH_FAIL_NCCL") == "1" or os.environ.get("VFIK_BENCH_FAIL_NCCL_RANK") == str(rank)
                if forced:
                    raise RuntimeError("forced by VFIK_BENCH_FAIL_NCCL%s (rehearsal of the fallback)" % ("" if os.environ.get("VFIK_BENCH_FAIL_NCCL") == "1" else "_RANK"))
                if not dist.is_nccl_available():
                    raise RuntimeError("torch.distributed has no nccl (RCCL) backend")
                torch.zeros(1, device=torch.device("cuda", local_rank)).add_(1)
                torch.cuda.synchronize()
                prop = torch.cuda.get_device_properties(local_rank)
                dev_id = str(getattr(prop, "uuid", "")) or "%s/%s" % (getattr(prop, "pci_bus_id", local_rank), getattr(prop, "pci_device_id", ""))
            except Exception as e:  # noqa: BLE001
                ok, why = False, (str(e).splitlines()[0][:160] if str(e) else type(e).__name__)
            backend, why, _ = launcher.agree_on_backend(store, rank, world, (ok, why, dev_id), timeout_s=bound)
            if backend == "nccl":
                # (2) every pre-check passed: bring RCCL up eagerly (device_id) and prove it with one all-reduce -- under a watchdog,
                # because a peer that dies inside ncclCommInitRank would otherwise leave this rank there for good
                ok = 1
                with launcher.Watchdog(bound, "rank %d: RCCL communicator start" % rank):
                    try:
                        dist.init_process_group("nccl", store=dist.PrefixStore("nccl", store), rank=rank, world_size=world,
                                                device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=bound))
                        t = torch.ones(1, device=torch.device("cuda", local_rank))
                        dist.all_reduce(t)           # the communicator really works
                        torch.cuda.synchronize()
                        ok = 1 if int(t.item()) == world else 0
                    except Exception as e:  # noqa: BLE001 -- whatever RCCL / c10d raise
                        ok, why = 0, (str(e).splitlines()[0][:160] if str(e) else type(e).__name__)
                    store.set("nccl_ok_%d" % rank, str(ok))
                    all_ok = all(store.get("nccl_ok_%d" % r) == b"1" for r in range(world))   # (bounded: the store's timeout is `bound`)
                if not all_ok:
                    if dist.is_initialized():
                        try:
                            dist.destroy_process_group()
                        except Exception:  # noqa: BLE001
                            pass
                    backend = "gloo"
                    why = why or "another rank's communicator did not come up"
            if backend == "gloo":
                backend_note = "gloo (RCCL communicator not available on every rank%s): barrier / timing reductions on the CPU" % (": " + why if why else "")
                print("[bench] rank %d: gloo for the barrier and the timing reductions%s" % (rank, ": " + why if why else ""), file=sys.stderr)
                dist.init_process_group("gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world)

    tdt = torch.float32 if io_dtype == np.float32 else torch.float64
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the timing reductions run
    # the launch stream: a stream of its own (not the legacy default stream), so that the K launches of a region can also be captured into a hipGraph
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    out_cols = {"pose": 16, "pose_nt": 16, "qdot_vf": chain.n, "qdot_null": chain.n, "qdist": chain.n}

    sharded = []

    def make_set(shift):
        """One independent input set: its own handle (field sets, nullspace state), q and output buffers.  Sets other than the
        first hold the same workload with the arms rolled by `shift` rows: other bytes at other addresses."""
        if shift == 0 and args.gather:
            # the product's sharded driver: this rank's shard of the world * B arms on its device, local rows in
            sharded.append(sharding.ShardedEngine(chain, world * B, rank=rank, world=world, devices=[local_rank],
                                                  io_dtype=io_dtype.type, max_slots=nobs, params=params))
            e = sharded[0].engines[0]
        else:
            e = engine.Engine(chain, B, io_dtype=io_dtype.type, max_slots=nobs, device=local_rank, params=params)
        e.set_fields(np.roll(w["fields"], shift, axis=0), np.roll(w["nfields"], shift))
        e.use_stream(stream.cuda_stream)
        qt = torch.from_numpy(np.roll(w["q"], shift, axis=0).astype(io_dtype)).to(dev)
        outs = {"qdot_out": torch.zeros(B, chain.n, dtype=tdt, device=dev)}
        for k in extra_outs:
            outs[k] = torch.zeros(B, dtype=torch.int32, device=dev) if k == "status" else torch.zeros(B, out_cols[k], dtype=tdt, device=dev)
        return e, e.make_io(qt, **outs), qt, outs

    eng, io, q, outs0 = make_set(0)
    qdot = outs0["qdot_out"]
    # bytes one launch really moves (PMC pass of an earlier run; the compact repeller image makes them fewer than the algorithmic ones)
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get(args.workload, {})
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc pass of this command, round %s; not measured by this run)" % rec.get("round")
        except Exception:
            traffic = None
    # what rocprofv3 measured for this workload's kernel (profiles/kernel_trace.json, written by tools/profile_digest.py from the round's
    # --kernel-trace --stats runs of this command): quoted beside the live HIP-event figure, never in place of it
    ktrace = None
    kt = os.path.join(ROOT, "profiles", "kernel_trace.json")
    if os.path.exists(kt):
        try:
            ktrace = json.load(open(kt)).get(args.workload)
        except Exception:
            ktrace = None
    # what the wave executed (profiles/pmc_sq.json, written by tools/profile_digest.py from the round's SQ counter pass of this command):
    # the VALU floor beside the HBM figure -- with one wave per SIMD the launch cannot be shorter than its wave's vector instructions
    # at 4 cycles each (wave64 on a 16-lane SIMD; float64 FMA is full rate on gfx950)
    sq = None
    sqf = os.path.join(ROOT, "profiles", "pmc_sq.json")
    if os.path.exists(sqf):
        try:
            sq = json.load(open(sqf)).get(args.workload)
        except Exception:
            sq = None
    # rotating input sets (the COLD state): enough of them that the bytes touched between two uses of a set exceed 640 MiB
    per_launch = float(traffic) if traffic else float(bytes_per_cycle * B)
    n_sets = 0
    if args.state != "warm" and args.rotate != 0:
        n_sets = args.rotate if args.rotate > 0 else max(24, int(640 * MIB / per_launch) + 2)
    sets = [(eng, io)]
    keep = [(q, outs0)]
    for k in range(1, n_sets):
        e_k, io_k, q_k, o_k = make_set(k * 2039)
        sets.append((e_k, io_k))
        keep.append((q_k, o_k))

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(values):
        """element-wise maximum over the ranks of a list of floats (one collective, outside every stamped interval)"""
        if world == 1:
            return list(values)
        t = torch.tensor(values, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]

    def capture(active_sets, K):
        """The K launches of a region as ONE hipGraph (torch.cuda.CUDAGraph captures the vfik_step calls on the launch stream)."""
        ns = len(active_sets)
        steppers = [e_i.stepper(io_i) for e_i, io_i in active_sets]
        g = torch.cuda.CUDAGraph()
        # (thread-local capture mode: a collective library's watchdog thread querying its events must not invalidate the capture)
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for i in range(K):
                steppers[i % ns]()
        torch.cuda.synchronize()
        return g

    def timed(active_sets, K, R, wall=True, graph=None):
        """R repetitions of the timed region; returns (wall seconds per WALL repetition, enqueue seconds, HIP-event ms per EVENT
        repetition).  Repetitions alternate between two kinds: WALL repetitions (even) carry nothing but the K launches between
        the stamps and give `value`; EVENT repetitions (odd) bracket the same K launches with HIP events on the launch stream
        and give the kernel's launch period for `roofline` (with one priming launch in front of the first event, so that all K
        timed launches run back to back whatever K is).  (Recording two timing events costs a 20-launch region ~20 us --
        1 us a step -- which is why they are kept out of the interval that `value` comes from.)  Launch i of a repetition
        goes to input set i mod len(active_sets): one set = back-to-back launches over the same bytes (cache-resident),
        many = every launch finds its inputs in HBM."""
        ns = len(active_sets)
        steppers = [e_i.stepper(io_i) for e_i, io_i in active_sets]   # byref / prototype bound once: the hot enqueue
        steps = [steppers[i % ns] for i in range(K)]
        wall_s, ev_pairs, enqueue_s = [], [], []
        for r in range(2 * R if wall else R):
            with_events = (r % 2 == 1) or not wall
            if with_events:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()                      # ranks start together; the barrier itself is NOT inside the interval
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if with_events:
                # one priming launch in front of the first event: the K launches between the events then run back to back, the
                # first of them included (a kernel that starts on an idle queue takes 0.7-1.6 us longer, profiles/r03_trace_gaps.txt:
                # at 20 launches per region that bias was 0.4 us per launch).  EVENT repetitions only; `value` never sees it.
                steppers[(ns - 1) % ns]()
                ev0.record(stream)
            if graph is not None:
                graph.replay()
            elif args.sync_each:
                for step in steps:
                    step()
                    torch.cuda.synchronize()
            else:
                for step in steps:
                    step()
            t_enq = time.perf_counter()
            if with_events:
                ev1.record(stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if with_events:
                ev_pairs.append((ev0, ev1))
            else:
                wall_s.append(t1 - t0)
                enqueue_s.append(t_enq - t0)
        return wall_s, enqueue_s, [a.elapsed_time(b) for a, b in ev_pairs]  # events: same stream as the launches

    K, R = args.steps, args.reps
    primary = sets if args.state == "cold" else sets[:1]
    for i in range(max(args.warmup, len(primary))):
        e_i, io_i = primary[i % len(primary)]
        e_i.step(io_i)
    torch.cuda.synchronize()
    # direct launches or a captured graph of them: decided in the (untimed) warm-up, per --launch
    graph, launch_note = None, "direct: K vfik_step calls per region"
    if args.launch != "direct" and not args.sync_each:
        g, why = None, ""
        try:
            g = capture(primary, K)
        except Exception as e:  # capture not possible on this stack: stay direct, say so
            why = str(e).splitlines()[0][:120] if str(e) else type(e).__name__
        # every rank takes the same path from here on (the trial regions carry barriers): graphs only if EVERY rank captured one
        all_ok = reduce_max([0.0 if g is not None else 1.0])[0] == 0.0
        if not all_ok:
            launch_note = "direct (graph capture failed%s)" % (": " + why if why else " on another rank")
        elif args.launch == "graph":
            graph = g
        else:
            d_s, _, _ = timed(primary, K, 5)
            g_s, _, _ = timed(primary, K, 5, graph=g)
            picks = reduce_max([pctl(d_s, 50), pctl(g_s, 50)])   # one decision for all ranks: on the slowest rank's times
            if picks[1] < 0.97 * picks[0]:
                graph = g
            launch_note += " (auto; warm-up trial, us per step: direct %.3f, graph %.3f)" % (picks[0] * 1e6 / K, picks[1] * 1e6 / K)
        if graph is not None:
            launch_note = launch_note.replace("direct: K vfik_step calls per region", "hipGraph: one replay of K captured vfik_step launches per region")
    wall_s, enqueue_s, ev_ms = timed(primary, K, R, graph=graph)
    own_wall_s = list(wall_s)
    wall_s = reduce_max(wall_s)        # per repetition: the slowest rank
    per_rank_ms = None
    if world > 1:
        mine = torch.tensor([pctl(own_wall_s, 50) * 1e3 / K], dtype=torch.float64, device=red_dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank_ms = [float(p.item()) for p in parts]

    # the COLD state beside the warm one (`--state both`): the same launches rotating over the input sets, HIP events only
    cold = None
    if args.state == "both" and n_sets > 1:
        for e_i, io_i in sets:
            e_i.step(io_i)
        torch.cuda.synchronize()
        cold_graph = None
        if graph is not None:
            try:
                cold_graph = capture(sets, K)
            except Exception:
                cold_graph = None
        _, _, cold_ms = timed(sets, K, R, wall=False, graph=cold_graph)
        cold_ms = reduce_max(cold_ms)
        cold_us = [m * 1e3 / K for m in cold_ms]
        cu = pctl(cold_us, 50)
        cold = {"state": "cold: %d input sets (own handle, q and output buffers each) launched round-robin; %.0f MiB touched between two uses "
                         "of a set, against 256 MiB of Infinity Cache" % (n_sets, (n_sets - 1) * per_launch / MIB),
                "input_sets": n_sets, "us_per_launch": cu, "us_per_launch_p10": pctl(cold_us, 10), "us_per_launch_p90": pctl(cold_us, 90),
                "achieved": bytes_per_cycle * B / (cu * 1e-6) / 1e9, "frac": bytes_per_cycle * B / (cu * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "real_bytes_frac": (traffic / (cu * 1e-6) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                "cycles_per_s": world * B / (cu * 1e-6), "launches_timed": R * K}

    # secondary figures (never `value`): the headline scene is the friendliest one -- every decay repeller with the feeder's safe
    # distance, force and order (uniform image, 16 B a slot, order 5 straight-line).  The same batch (a) with each arm's own safe
    # distance and force (compact image, 24 B a slot) and (b) with integer orders that differ by obstacle as in old/README.old:75
    # (order-20 obstacles beside an order-5 near-goal repeller: the order planes, MIXO kernel variants), warm, HIP events only.
    secondary = None
    if world == 1 and args.workload == "C3" and args.secondary:
        secondary = {}
        rng2 = np.random.default_rng(77)
        variants = {}
        f1 = w["fields"].copy()
        f1["p"][:, 1:1 + nobs, 4] = rng2.uniform(0.001, 0.004, (B, 1)).astype(io_dtype)
        f1["force"][:, 1:1 + nobs] = rng2.uniform(-12.0, -8.0, (B, 1)).astype(io_dtype)
        variants["per_arm_safe_distance_and_force"] = f1
        f2 = w["fields"].copy()
        f2["p"][:, 1:1 + nobs, 5] = [5.0] + [20.0] * (nobs - 1)
        variants["orders_5_and_20_by_obstacle"] = f2
        f3 = w["fields"].copy()
        f3["p"][:, 1:1 + nobs, 5] = rng2.integers(1, 21, (B, nobs)).astype(np.float64)
        variants["orders_of_each_arm_its_own"] = f3
        from oracle import oracle_c as _oc   # checker only
        for name, fv in variants.items():
            e2 = engine.Engine(chain, B, io_dtype=io_dtype.type, max_slots=nobs, device=local_rank, params=params)
            e2.set_fields(fv, w["nfields"])
            e2.use_stream(stream.cuda_stream)
            o2 = torch.zeros(B, chain.n, dtype=tdt, device=dev)
            io2 = e2.make_io(q, qdot_out=o2)
            for _ in range(max(3, args.warmup)):
                e2.step(io2)
            torch.cuda.synchronize()
            _, _, ms2 = timed([(e2, io2)], K, min(R, 10), wall=False)
            us2 = pctl([m * 1e3 / K for m in ms2], 50)
            ref2 = _oc.cycle_batch(chain, params, w["q"][:4096], fv[:4096], w["nfields"][:4096], want=("qdot_out",))
            secondary[name] = {"us_per_launch": us2, "cycles_per_s": B / (us2 * 1e-6), "frac": bytes_per_cycle * B / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                               "field_path": e2.field_path, "uniform_repellers": e2.uniform_repellers, "mixed_orders": e2.mixed_orders,
                               "max_abs_err_rad_s_first_4096_arms": float(np.abs(o2[:4096].cpu().numpy().astype(np.float64) - ref2["qdot_out"]).max())}
            e2.close()

    # secondary figure (never `value`): closed-loop rollout, K control cycles per launch with q integrated in
    # registers -- what the cycle costs once the per-launch boundary is amortised
    rollout = None
    if args.rollout > 0:
        q_end = torch.empty_like(q)
        qd2 = torch.empty_like(qdot)
        io_r = eng.make_io(q, qdot_out=qd2)
        eng.rollout(io_r, args.rollout, 1e-3, q_out=q_end)
        torch.cuda.synchronize()
        launches = 5
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        r0.record(stream)
        for _ in range(launches):
            eng.rollout(io_r, args.rollout, 1e-3, q_out=q_end)
        r1.record(stream)
        torch.cuda.synchronize()
        r_ms = reduce_max([r0.elapsed_time(r1)])[0]
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
        src = qdot if backend == "nccl" else qdot.cpu()
        gathered = sharded[0].gather(src)   # sharding.collate: one all_gather over xGMI (RCCL), gloo in the rehearsal

    if rank == 0:
        got = qdot.cpu().numpy().astype(np.float64)
        from oracle import oracle_c  # checker only: accuracy half of the metric
        ref = oracle_c.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out",))
        max_err = float(np.abs(got - ref["qdot_out"]).max())
        med_s, p10_s, p90_s = pctl(wall_s, 50), pctl(wall_s, 10), pctl(wall_s, 90)
        us_launch = [m * 1e3 / K for m in ev_ms]
        us_med = pctl(us_launch, 50)
        if args.dump_reps:
            with open(args.dump_reps, "w") as f:
                f.write("\n".join("%.4f %.4f" % (u, ws * 1e6 / K) for u, ws in zip(us_launch, wall_s)) + "\n")
        achieved = bytes_per_cycle * B / (us_med * 1e-6) / 1e9
        total = world * B * K
        if len(primary) > 1:
            state_txt = ("cold: every launch of the run rotates over %d input sets (own handle, q and output buffers each), %.0f MiB touched "
                         "between two uses of a set, against 256 MiB of Infinity Cache: inputs come from HBM" % (len(primary), (len(primary) - 1) * per_launch / MIB))
        else:
            state_txt = ("warm: back-to-back launches over ONE input set (%.1f MB), which stays in the 256 MiB Infinity Cache: this is "
                         "the fraction of the HBM roofline at cache-resident inputs; `cold` = the same launches with inputs from HBM" % (per_launch / 1e6))
        line = {
            "metric": "7-DOF IK cycles/sec (whole node), batch=65536; max |qdot-qdot_ref|",
            "value": total / med_s,
            "unit": "cycles/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": med_s * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: batch=%d/GPU x %d-DOF %s, goal + %d decay repellers, %s I/O, flags=0x%x"
                                   % (args.workload, B, chain.n, chain.name, nobs, io_name, flags),
                       "parallelism": "arm batch sharded over %d GPU(s), no collective" % world,
                       "lambda": params.lambda_, "launches_per_step": 1, "launch": launch_note,
                       "host_enqueue_us_per_step": pctl(enqueue_s, 50) * 1e6 / K},
            "repetitions": {"count": R, "steps_each": K, "value_p10": total / p90_s, "value_median": total / med_s, "value_p90": total / p10_s,
                            "ms_per_step_p10": p10_s * 1e3 / K, "ms_per_step_median": med_s * 1e3 / K, "ms_per_step_p90": p90_s * 1e3 / K,
                            "timed": "per repetition: barrier, synchronize, stamp, K launches, synchronize, stamp; max over ranks per repetition; "
                                     "no collective inside a stamped interval"},
            "max_abs_err_rad_s": max_err,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         # which memory the launches of the timed region found their inputs in
                         "state": state_txt,
                         "real_bytes_frac": (traffic / (us_med * 1e-6) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         # <io type, joints, nullspace module, PLAIN, rollout, straight-line field path, LEAN>: the bench
                         # workloads (revolute chain, identity tool, unit weights, integer-order repellers, qdot_out only)
                         "kernel": kernel_name(io_name, chain.n, flags, B, eng.small_batch_launches > 0, bool(extra_outs), eng.uniform_repellers, eng.dh_pattern),
                         "algorithmic_bytes_per_cycle": bytes_per_cycle,
                         "us_per_launch_hip_events": us_med,
                         "us_per_launch_p10": pctl(us_launch, 10), "us_per_launch_p90": pctl(us_launch, 90),
                         "launches_timed": R * K, "timed": "R repetitions of K launches between two HIP events on the launch stream (one untimed priming launch in front of the first event), interleaved with the wall-clock repetitions"},
        }
        if args.sync_each:
            line["diagnostic"] = "--sync-each: every launch was followed by a synchronize; not a throughput measurement"
        if sq and sq.get("SQ_INSTS_VALU"):
            clock = float(sq.get("clock_ghz", 2.2))
            floor_us = sq["SQ_INSTS_VALU"] * 4.0 / (clock * 1e3)
            line["roofline"]["valu"] = {
                "insts_per_wave": sq["SQ_INSTS_VALU"], "floor_us": floor_us, "frac": floor_us / us_med, "clock_ghz": clock,
                "salu_per_wave": sq.get("SQ_INSTS_SALU"), "active_quad_cycles_per_wave": sq.get("SQ_ACTIVE_INST_ANY"),
                "issue_stall_quad_cycles_per_wave": sq.get("SQ_WAIT_INST_ANY"),
                "meaning": "the launch's waves are one per SIMD: floor_us = VALU instructions of a wave x 4 cycles / clock is the time the SIMD "
                           "needs to issue them if nothing ever stalled; frac = floor_us / launch period.  This, not HBM, is the roof that binds "
                           "(DESIGN.md 5.3): the kernel computes in float64, one instruction per 4 cycles per SIMD",
                "source": "profiles/pmc_sq.json (rocprofv3 --pmc SQ_* pass of this command, round %s; not measured by this run)" % sq.get("round")}
        if secondary is not None:
            line["secondary"] = secondary
        if cold is not None:
            line["roofline"]["cold"] = cold
        if ktrace:
            # (traced with the launches replayed from a hipGraph, back to back as in the untraced run; under the tracer every dispatch
            # carries profiling work: the traced means run 1-5 % above the untraced launch period)
            line["roofline"]["kernel_trace"] = {
                "source": "profiles/kernel_trace.json <- profiles/r%02d_kernel_stats_{warm,cold}_%s.csv (rocprofv3 --kernel-trace --stats of this command "
                          "with --launch graph; not measured by this run)" % (ktrace.get("round", 0), args.workload),
                "kernel": ktrace.get("kernel"),
                **{st: {k: ktrace[st].get(k) for k in ("mean_ns", "median_ns", "min_ns", "dispatches", "frac")} for st in ("warm", "cold") if st in ktrace}}
        if extra_outs:
            line["config"]["outputs"] = ["qdot_out"] + list(extra_outs)
        if world > 1:
            line["config"]["process_group"] = backend_note or ("nccl (RCCL): barrier and timing reductions only" if backend == "nccl" else "gloo")
        if per_rank_ms is not None:
            line["per_rank_ms_per_step"] = per_rank_ms
            line["config"]["rank0_cpu_binding"] = binding
        if rollout is not None:
            line["rollout"] = rollout
        if host_path is not None:
            line["host_path"] = host_path
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if gathered is not None:
            line["config"]["gathered_rows"] = int(gathered.shape[0])
        print(json.dumps(line), flush=True)

    for e_k, _ in sets:
        e_k.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
