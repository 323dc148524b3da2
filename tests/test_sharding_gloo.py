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
