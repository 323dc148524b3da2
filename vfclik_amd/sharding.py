"""Batch sharding across the GPUs of a node.

Every arm is an independent problem (the reference runs one process set per arm,
/root/reference/scripts/vfclik:88-105), so the batch splits into contiguous slices, one per rank,
and the control path needs no collective.  ``collate`` is the optional gather of per-rank results
into one array (one all_gather; RCCL over xGMI on GPUs, gloo in the CPU tests).
"""


def shard_range(batch, rank, world):
    """Contiguous slice [lo, hi) of rank ``rank``: sizes differ by at most one, lower ranks first."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(int(batch), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(batch, world):
    return [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]


def collate(local, batch, dist=None):
    """All-gather the per-rank rows (torch tensor (b_r, ...)) into the full (batch, ...) tensor on
    every rank.  Ragged shards are padded to the largest shard for the collective."""
    import torch
    if dist is None:
        import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = shard_sizes(batch, world)
    assert local.shape[0] == sizes[rank], "rank %d holds %d rows, expected %d" % (rank, local.shape[0], sizes[rank])
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
