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
