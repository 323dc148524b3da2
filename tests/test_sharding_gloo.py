"""N > 1 host logic on the CPU: world-size-2 gloo processes shard a batch, each computes its slice
(the CPU oracle stands in for the GPU step here -- tests may use it), results are collated with
one all_gather and compared with the un-sharded computation."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, q):
    import torch
    import torch.distributed as dist
    from oracle import oracle_c
    from vfclik_amd import _abi, robots, sharding, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chain = robots.lwr()
    w = synth.make_workload(chain, batch, 3, seed=7, io_dtype=np.float64)  # same seed: same global batch
    lo, hi = sharding.shard_range(batch, rank, world)
    params = _abi.default_params()
    loc = oracle_c.cycle_batch(chain, params, w["q"][lo:hi], w["fields"][lo:hi], w["nfields"][lo:hi], want=("qdot_out",))
    full = sharding.collate(torch.from_numpy(loc["qdot_out"]), batch)
    # the bench's timing reduction: max over ranks
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((full.numpy(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [64, 101])
def test_two_rank_shard_and_collate(batch):
    import torch.multiprocessing as mp
    from oracle import oracle_c
    from vfclik_amd import _abi, robots, synth
    oracle_c.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chain = robots.lwr()
    w = synth.make_workload(chain, batch, 3, seed=7, io_dtype=np.float64)
    ref = oracle_c.cycle_batch(chain, _abi.default_params(), w["q"], w["fields"], w["nfields"], want=("qdot_out",))
    assert full.shape == (batch, 7)
    assert np.array_equal(full, ref["qdot_out"])  # same code, same inputs, only the slicing differs
    assert tmax == 2.0


def test_shard_ranges_cover_the_batch():
    from vfclik_amd.sharding import shard_range, shard_sizes
    for batch in (1, 7, 64, 65536, 524288, 1000003):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(batch, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == batch
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            s = shard_sizes(batch, world)
            assert max(s) - min(s) <= 1 and sum(s) == batch
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


class _OracleEngine:
    """Stand-in for engine.Engine in CPU tests of the sharding layer (the product has no CPU path): same
    set_fields / step_host surface, computed by the oracle."""

    def __init__(self, chain, batch, device=0, params=None, **kw):
        from vfclik_amd import _abi
        self.chain, self.batch, self.device = chain, batch, device
        self.params = params if params is not None else _abi.default_params()

    def set_fields(self, fields, counts, first_arm=0):
        assert fields.shape[0] == self.batch and counts.shape[0] == self.batch
        self.fields, self.counts = fields, counts

    io_dtype = np.float64

    def step_host(self, q, null_control=None, want=("qdot_out",), into=None, active=None, **kw):
        from oracle import oracle_c
        assert q.shape[0] == self.batch
        out = oracle_c.cycle_batch(self.chain, self.params, q, self.fields, self.counts, null_control=null_control, want=tuple(want),
                                   active=active, into=None if into is None else {k: v.copy() for k, v in into.items()})
        if into is not None:
            for k in want:
                into[k][...] = out[k]
            return into
        return out

    def close(self):
        pass


def _sharded_worker(rank, world, port, batch, devices, q):
    import torch.distributed as dist
    from vfclik_amd import robots, sharding, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chain = robots.lwr()
    w = synth.make_workload(chain, batch, 3, seed=11, io_dtype=np.float64)
    sh = sharding.ShardedEngine(chain, batch, devices=devices, engine_factory=_OracleEngine)   # rank / world from the group
    assert (sh.rank, sh.world) == (rank, world) and (sh.lo, sh.hi) == sharding.shard_range(batch, rank, world)
    sh.set_fields(w["fields"], w["nfields"])                                                     # global arrays in
    out = sh.step_global(w["q"], want=("qdot_out", "pose"))                                      # global rows out
    # the rank-local form used by bench.py --gather: local field rows in, local rows out, one collate
    sh2 = sharding.ShardedEngine(chain, batch, rank=rank, world=world, devices=[0], engine_factory=_OracleEngine)
    sh2.set_fields(sh2.local(w["fields"]), sh2.local(w["nfields"]), global_rows=False)
    loc = sh2.step_host(sh2.local(w["q"]), global_rows=False)
    import torch
    full2 = sh2.gather(torch.from_numpy(loc["qdot_out"])).numpy()
    if rank == 0:
        q.put((out["qdot_out"], out["pose"], full2, [(a, b) for a, b, _ in sh.parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch,devices", [(101, [0]), (64, [0, 1, 2]), (3, [0, 1, 2, 3])])
def test_sharded_engine_global_batch_in_global_rows_out(batch, devices):
    import torch.multiprocessing as mp
    from oracle import oracle_c
    from vfclik_amd import _abi, robots, synth
    oracle_c.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, batch, devices, q)) for r in range(2)]
    for p in procs:
        p.start()
    qdot, pose, full2, parts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chain = robots.lwr()
    w = synth.make_workload(chain, batch, 3, seed=11, io_dtype=np.float64)
    ref = oracle_c.cycle_batch(chain, _abi.default_params(), w["q"], w["fields"], w["nfields"], want=("qdot_out", "pose"))
    assert np.array_equal(qdot, ref["qdot_out"]) and np.array_equal(pose, ref["pose"]) and np.array_equal(full2, ref["qdot_out"])
    # rank 0's rows are split over its devices contiguously, empty parts dropped
    assert parts[0][0] == 0 and all(parts[i][1] == parts[i + 1][0] for i in range(len(parts) - 1))
    assert len(parts) == min(len(devices), (batch + 1) // 2)


def test_sharded_engine_single_process_over_several_devices():
    """world = 1: one process, one handle per device, the batch split over them (no torch.distributed needed)."""
    from oracle import oracle_c
    from vfclik_amd import _abi, robots, sharding, synth
    oracle_c.build()
    chain = robots.lwr()
    w = synth.make_workload(chain, 37, 2, seed=5, io_dtype=np.float64)
    sh = sharding.ShardedEngine(chain, 37, rank=0, world=1, devices=[0, 1, 2, 3], engine_factory=_OracleEngine)
    assert [e.device for e in sh.engines] == [0, 1, 2, 3] and sum(e.batch for e in sh.engines) == 37
    sh.set_fields(w["fields"], w["nfields"])
    out = sh.step_global(w["q"])
    ref = oracle_c.cycle_batch(chain, _abi.default_params(), w["q"], w["fields"], w["nfields"], want=("qdot_out",))
    assert np.array_equal(out["qdot_out"], ref["qdot_out"])
    with pytest.raises(ValueError):
        sh.step_host(w["q"][:5])
    sh.close()


class _PipelinedEngine(_OracleEngine):
    """... with the pipelined host path (submit_host / wait) and a log of the calls."""
    log = []

    def host_array(self, shape, dtype=None):
        return np.zeros(shape, dtype=np.float64 if dtype is None else dtype)

    def submit_host(self, q, outs, null_control=None, q_ref=None, q_cmded=None, active=None, q_lo=None, q_hi=None):
        _PipelinedEngine.log.append(("submit", self.device))
        self._pending = (q.copy(), outs, null_control, active)
        return 17 + self.device

    def wait(self, ticket):
        from oracle import oracle_c
        assert ticket == 17 + self.device
        _PipelinedEngine.log.append(("wait", self.device))
        q, outs, nc, active = self._pending
        res = oracle_c.cycle_batch(self.chain, self.params, q, self.fields, self.counts, null_control=nc, want=tuple(outs), active=active,
                                   into={k: v.copy() for k, v in outs.items()})
        for k in outs:
            outs[k][...] = res[k]


def test_every_device_gets_its_work_before_the_first_wait():
    """ShardedEngine.step_host in one process over several devices: all submits precede the first wait (the devices run
    concurrently, as the reference's per-arm process sets do, vfclik:88-105); results equal the un-sharded computation; `into`
    is honoured; the gated form (no pipelined path) goes through one thread per device and gives the same rows."""
    from oracle import oracle_c
    from vfclik_amd import _abi, robots, sharding, synth
    oracle_c.build()
    chain = robots.lwr()
    w = synth.make_workload(chain, 41, 2, seed=9, io_dtype=np.float64)
    ref = oracle_c.cycle_batch(chain, _abi.default_params(), w["q"], w["fields"], w["nfields"], want=("qdot_out", "pose", "status"))
    _PipelinedEngine.log = []
    sh = sharding.ShardedEngine(chain, 41, rank=0, world=1, devices=[0, 1, 2], engine_factory=_PipelinedEngine)
    sh.set_fields(w["fields"], w["nfields"])
    out = sh.step_host(w["q"], want=("qdot_out", "pose", "status"))
    kinds = [k for k, _ in _PipelinedEngine.log]
    assert kinds == ["submit"] * 3 + ["wait"] * 3, _PipelinedEngine.log
    for k in ("qdot_out", "pose", "status"):
        assert np.array_equal(out[k], ref[k])
    # into: the same arrays come back, written in place
    again = sh.step_host(w["q"], want=("qdot_out", "pose", "status"), into=out)
    assert again["qdot_out"] is out["qdot_out"] and np.array_equal(out["qdot_out"], ref["qdot_out"])
    # the fresh-q gate travels the same way; gated arms keep their rows of `into`
    _PipelinedEngine.log = []
    gate = np.arange(41) % 3 != 0
    marked = {"qdot_out": np.full((41, 7), 9.0)}
    got = sh.step_host(w["q"], want=("qdot_out",), active=gate, into=marked)
    assert [k for k, _ in _PipelinedEngine.log] == ["submit"] * 3 + ["wait"] * 3
    assert np.array_equal(got["qdot_out"][gate], ref["qdot_out"][gate]) and np.all(got["qdot_out"][~gate] == 9.0)
    sh.close()
    # handles without the pipelined path: one thread per device around step_host, same rows
    sh = sharding.ShardedEngine(chain, 41, rank=0, world=1, devices=[0, 1, 2], engine_factory=_OracleEngine)
    sh.set_fields(w["fields"], w["nfields"])
    marked = {"qdot_out": np.full((41, 7), 9.0)}
    got = sh.step_host(w["q"], want=("qdot_out",), active=gate, into=marked)
    assert np.array_equal(got["qdot_out"][gate], ref["qdot_out"][gate]) and np.all(got["qdot_out"][~gate] == 9.0)
    sh.close()


def test_a_rank_without_arms_returns_empty_rows():
    """More ranks than arms: the rank owns no part; its step returns (0, columns) arrays instead of raising."""
    from vfclik_amd import robots, sharding
    chain = robots.lwr()
    sh = sharding.ShardedEngine(chain, 2, rank=3, world=4, devices=[0], engine_factory=_OracleEngine)
    assert sh.local_rows == 0 and not sh.parts
    out = sh.step_host(np.zeros((0, 7)), want=("qdot_out", "pose", "status"), global_rows=False)
    assert out["qdot_out"].shape == (0, 7) and out["pose"].shape == (0, 16) and out["status"].shape == (0,)
