"""`python bench.py --gpus N` without an external launcher: the parent starts one fresh process per GPU
before making any GPU call, relays rank 0's line and fails when a rank fails (vfclik_amd/launcher.py; the
reference's counterpart is the process set of scripts/vfclik:88-121).  CPU only: gloo, world size 2."""
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys, time
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    mode = sys.argv[1]
    if mode == "fail" and rank == 1:
        sys.exit(3)
    dist.init_process_group("gloo")
    if mode == "fail":
        time.sleep(60)   # rank 0 would hang in a collective without its peer: the parent must end it
    # the bench's timing protocol: barrier, stamp, work, stamp, THEN the collective
    reps = []
    for r in range(3):
        dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))
        reps.append(time.perf_counter() - t0)
    t = torch.tensor(reps, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mine = torch.tensor([float(rank)])
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "max_s": t.tolist(), "ranks": [float(p) for p in parts]}))
    else:
        print("noise from rank", rank)   # must not reach the relayed output
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.fixture()
def rank_script(tmp_path):
    p = tmp_path / "rank.py"
    p.write_text(RANK_SCRIPT)
    return str(p)


def test_spawn_two_ranks_and_collect(rank_script):
    from vfclik_amd import launcher
    rc, out = launcher.spawn_ranks([sys.executable, rank_script, "ok"], 2, timeout=150)
    assert rc == 0
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "noise" not in out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == [0.0, 1.0]
    assert all(s >= 0.019 for s in d["max_s"])  # the slower rank (20 ms) sets every repetition's time


def test_a_failed_rank_fails_the_run(rank_script):
    from vfclik_amd import launcher
    rc, _ = launcher.spawn_ranks([sys.executable, rank_script, "fail"], 2, timeout=150)
    assert rc == 3


def test_rank_env():
    from vfclik_amd import launcher
    e = launcher.rank_env(1, 4, 2345, base={})
    assert e["RANK"] == "1" and e["LOCAL_RANK"] == "1" and e["WORLD_SIZE"] == "4"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "2345" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    with pytest.raises(ValueError):
        launcher.spawn_ranks(["true"], 0)


def test_bench_parent_spawns_before_any_gpu_call(monkeypatch):
    """--gpus 2 with no WORLD_SIZE: bench.main hands the command line to the launcher and never imports torch."""
    sys.path.insert(0, ROOT)
    import bench
    from vfclik_amd import launcher
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake(script, argv, world):
        seen.update(script=script, argv=list(argv), world=world)
        return 0

    monkeypatch.setattr(launcher, "main_spawn", fake)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "20", "--warmup", "5", "--single-device", "--dist-backend", "gloo"])
    assert e.value.code == 0
    assert seen["world"] == 2 and os.path.basename(seen["script"]) == "bench.py"
    assert seen["argv"] == ["--gpus", "2", "--steps", "20", "--warmup", "5", "--single-device", "--dist-backend", "gloo"]


def test_percentiles():
    sys.path.insert(0, ROOT)
    import bench
    xs = list(range(1, 12))
    assert bench.pctl(xs, 50) == 6 and bench.pctl(xs, 10) == 2 and bench.pctl(xs, 90) == 10 and bench.pctl([7.0], 90) == 7.0


BIG_SCRIPT = textwrap.dedent("""
    import os, sys, time
    rank = int(os.environ["RANK"])
    if rank == 0:
        line = "x" * 1023
        for i in range(512):          # 512 KiB: eight times what a pipe holds
            print(line)
        print("END-OF-RANK-0")
    else:
        time.sleep(0.5)
""")


def test_rank0_output_beyond_the_pipe_capacity_does_not_deadlock(tmp_path):
    """rank 0's stdout is drained while the parent waits (RCCL logs at NCCL_DEBUG=INFO go to stdout)."""
    from vfclik_amd import launcher
    p = tmp_path / "big.py"
    p.write_text(BIG_SCRIPT)
    rc, out = launcher.spawn_ranks([sys.executable, str(p)], 2, timeout=60)
    assert rc == 0
    assert out.rstrip().endswith("END-OF-RANK-0") and len(out) > 512 * 1024


def test_a_stuck_rank_times_out_and_everything_is_terminated(tmp_path):
    from vfclik_amd import launcher
    p = tmp_path / "stuck.py"
    p.write_text("import os, time\nprint('hello from', os.environ['RANK'], flush=True)\ntime.sleep(300)\n")
    import time
    t0 = time.monotonic()
    rc, out = launcher.spawn_ranks([sys.executable, str(p)], 2, timeout=2.0)
    assert rc == 124 and "hello from 0" in out
    assert time.monotonic() - t0 < 30
    # main_spawn has a finite default and reports the timeout as its exit code
    assert launcher.DEFAULT_SPAWN_TIMEOUT_S > 0
    assert launcher.main_spawn(str(p), [], 2, timeout=1.5) == 124


def test_rank_cpu_shares_are_disjoint_and_cover_every_rank():
    from vfclik_amd import launcher
    cpus = list(range(16, 48))
    solo = [[c] for c in cpus]                      # no SMT: every CPU its own core
    shares = [launcher.rank_cpus(r, 8, cpus, groups=solo) for r in range(8)]
    assert all(len(s) == 4 for s in shares)
    flat = [c for s in shares for c in s]
    assert len(set(flat)) == len(flat) and set(flat) <= set(cpus)
    assert launcher.rank_cpus(0, 2, list(range(64)), groups=[[c] for c in range(64)]) == list(range(8))          # capped at 8 per rank
    assert launcher.rank_cpus(1, 2, list(range(64)), max_per_rank=0, groups=[[c] for c in range(64)]) == list(range(32, 64))
    assert launcher.rank_cpus(5, 8, [3, 4], groups=[[3], [4]]) == [4]                              # fewer cores than ranks: shared
    with pytest.raises(ValueError):
        launcher.rank_cpus(2, 2, cpus)
    # SMT: CPUs c and c + 128 are the two threads of core c (a 128-core, 256-thread host): eight ranks get eight DIFFERENT sets
    # of physical cores -- rank 4 must not land on the siblings of rank 0's cores
    smt = [[c, c + 128] for c in range(128)]
    shares = [launcher.rank_cpus(r, 8, list(range(256)), groups=smt) for r in range(8)]
    cores = [{c % 128 for c in s} for s in shares]
    assert all(len(s) == 8 and len(k) == 4 for s, k in zip(shares, cores))       # 4 cores x 2 threads
    assert all(cores[a].isdisjoint(cores[b]) for a in range(8) for b in range(a + 1, 8))
    assert shares[0] == [0, 1, 2, 3, 128, 129, 130, 131] and shares[4][0] == 64
    # the topology of THIS machine, whatever it is: shares of different ranks never share a core
    groups = launcher._sibling_groups(set(launcher.allowed_cpus()))
    assert sorted(c for g in groups for c in g) == launcher.allowed_cpus()


def test_pin_rank_narrows_the_affinity_of_a_fresh_process(tmp_path):
    """A rank pins itself before its first GPU call; done in a child so that the test runner keeps its own mask."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from vfclik_amd import launcher\n"
            "before = sorted(os.sched_getaffinity(0))\n"
            "mine = launcher.pin_rank(1, 2, log=sys.stdout)\n"
            "after = sorted(os.sched_getaffinity(0))\n"
            "assert mine == after and set(after) <= set(before), (mine, before, after)\n"
            "assert len(before) < 2 or after != before\n"
            "print('OK', after)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "pinned to CPUs" in r.stdout and "OK" in r.stdout
    env = dict(os.environ, VFIK_NO_PIN="1")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nfrom vfclik_amd import launcher\nassert launcher.pin_rank(0, 2) is None" % ROOT],
                       capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0, r.stderr


def test_rank_cpus_near_the_gpu_numa_node():
    """Two sockets of 64 cores x 2 threads, GPUs 0-3 on node 0 and 4-7 on node 1: every rank gets cores of ITS GPU's node, the four ranks
    of a node split that node's cores, nobody shares a core; unknown or short node lists fall back to the plain split."""
    from vfclik_amd import launcher
    cpus = list(range(256))
    smt = [[c, c + 128] for c in range(128)]
    node0 = set(range(0, 64)) | set(range(128, 192))
    node1 = set(range(64, 128)) | set(range(192, 256))
    lists = [node0] * 4 + [node1] * 4
    shares = [launcher.rank_cpus_near_gpu(r, 8, cpus, lists, groups=smt) for r in range(8)]
    for r, s in enumerate(shares):
        assert set(s) <= (node0 if r < 4 else node1) and len(s) == 8
    cores = [{c % 128 for c in s} for s in shares]
    assert all(cores[a].isdisjoint(cores[b]) for a in range(8) for b in range(a + 1, 8))
    assert launcher.rank_cpus_near_gpu(1, 2, cpus, None if launcher.gpu_local_cpus() is None else [node0], groups=smt) == launcher.rank_cpus(1, 2, cpus, groups=smt)
    assert launcher.rank_cpus_near_gpu(0, 2, cpus, [set(), set()], groups=smt) == launcher.rank_cpus(0, 2, cpus, groups=smt)
    assert launcher._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}


# ---- which backend carries the barrier: agreed through the store BEFORE the collective communicator start ----
def test_decide_backend_rules():
    from vfclik_amd import launcher
    ok = (True, "", "GPU-a")
    assert launcher.decide_backend([ok, (True, "", "GPU-b")]) == ("nccl", "")
    b, why = launcher.decide_backend([ok, (False, "no RCCL", "")])
    assert b == "gloo" and why.startswith("rank 1: no RCCL")
    b, why = launcher.decide_backend([ok, (True, "", "GPU-b"), (True, "", "GPU-a")])
    assert b == "gloo" and "ranks 0 and 2 share device GPU-a" in why
    assert launcher.decide_backend([(True, "", ""), (True, "", "")])[0] == "nccl"   # unknown device ids: RCCL is tried (under the watchdog)


AGREE_SCRIPT = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from vfclik_amd import launcher
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[1]
    store, _, _ = next(dist.rendezvous("env://", rank=rank, world_size=world))
    t0 = time.time()
    if mode == "one_fails":          # ONLY rank 1 fails its pre-check: every rank must still reach gloo, nobody enters a collective first
        pre = (rank != 1, "device lost" if rank == 1 else "", "GPU-%%d" %% rank)
    elif mode == "shared":
        pre = (True, "", "GPU-0")
    elif mode == "silent":           # rank 1 never publishes: rank 0 must give up after the bound, not hang
        if rank == 1:
            time.sleep(30)
            sys.exit(0)
        pre = (True, "", "GPU-0")
    else:
        pre = (True, "", "GPU-%%d" %% rank)
    backend, why, got = launcher.agree_on_backend(store, rank, world, pre, timeout_s=3.0)
    if backend == "gloo":
        dist.init_process_group("gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"backend": backend, "why": why, "seconds": time.time() - t0}))
""") % ROOT


@pytest.mark.parametrize("mode,backend,why", [("one_fails", "gloo", "rank 1: device lost"), ("shared", "gloo", "share device GPU-0"), ("fine", "nccl", "")])
def test_ranks_agree_on_the_backend_through_the_store(tmp_path, mode, backend, why):
    from vfclik_amd import launcher
    script = tmp_path / "agree.py"
    script.write_text(AGREE_SCRIPT)
    rc, out = launcher.spawn_ranks([sys.executable, str(script), mode], 2, timeout=120)
    assert rc == 0, out
    line = json.loads(out.strip().splitlines()[-1])
    assert line["backend"] == backend and why in line["why"] and line["seconds"] < 30


def test_a_rank_that_never_publishes_costs_a_bounded_wait(tmp_path):
    from vfclik_amd import launcher
    import time
    script = tmp_path / "agree.py"
    script.write_text(AGREE_SCRIPT)
    t0 = time.time()
    rc, out = launcher.spawn_ranks([sys.executable, str(script), "silent"], 2, timeout=120)
    assert rc != 0 and time.time() - t0 < 25     # rank 0 raised after its 3-s bound; the parent ended rank 1


def test_watchdog_ends_a_process_that_is_stuck(tmp_path):
    import subprocess
    import time
    code = "import sys, time; sys.path.insert(0, %r)\nfrom vfclik_amd import launcher\nwith launcher.Watchdog(0.5, 'test', code=7):\n    time.sleep(30)\n" % ROOT
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 7 and time.time() - t0 < 20 and "did not finish within" in p.stderr
    code = "import sys; sys.path.insert(0, %r)\nfrom vfclik_amd import launcher\nwith launcher.Watchdog(5, 'test', code=7):\n    pass\nprint('ok')\n" % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout.strip() == "ok"
