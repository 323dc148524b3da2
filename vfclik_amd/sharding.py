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


def _rank_world(rank, world):
    """(rank, world) of this process: explicit arguments, else an initialised torch.distributed group, else the
    launcher's environment (RANK / WORLD_SIZE), else a single process."""
    import os
    if rank is not None and world is not None:
        return int(rank), int(world)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


class ShardedEngine:
    """A global batch of arms over the GPUs of a node: the batched analogue of scripts/vfclik:88-105, which starts
    one process set per arm.  Rank r of ``world`` (one process per GPU under torch.distributed / bench.py's own
    launcher) owns the contiguous arms ``shard_range(batch, r, world)``; inside a rank the shard is split once more over
    ``devices`` (default: the rank's LOCAL_RANK device when world > 1, every visible device in a single process), one
    ``Engine`` handle and one stream per device.  The devices of a rank work CONCURRENTLY, as the reference's per-arm
    process sets do (vfclik:88-105): a cycle is submitted to every device before the first one is waited for
    (:meth:`step_host`).  Global arrays in, the rank's rows out; ``gather`` collates the rows of all ranks (one
    all_gather, never on the control path).

    engine_factory(chain, batch, device=..., **kw) builds a handle (default ``engine.Engine``; CPU tests inject a
    stand-in -- the product has no CPU path)."""

    def __init__(self, chain, batch, rank=None, world=None, devices=None, engine_factory=None, **engine_kw):
        import os
        self.chain = chain
        self.batch = int(batch)
        self.rank, self.world = _rank_world(rank, world)
        self.lo, self.hi = shard_range(self.batch, self.rank, self.world)
        if engine_factory is None:
            from .engine import Engine as engine_factory
        if devices is None:
            if self.world > 1:
                devices = [int(os.environ.get("LOCAL_RANK", self.rank))]
            else:
                from .engine import load_library
                devices = list(range(max(1, load_library().vfik_device_count())))
        self.devices = list(devices)
        # the rank's rows over its devices: again contiguous, sizes differing by at most one; devices left without an
        # arm (more devices than arms) get no handle
        self.parts = []
        for k, dev in enumerate(self.devices):
            a, b = shard_range(self.hi - self.lo, k, len(self.devices))
            if b > a:
                self.parts.append((self.lo + a, self.lo + b, engine_factory(chain, b - a, device=dev, **engine_kw)))
        self.engines = [e for _, _, e in self.parts]
        self._pin = {}

    @property
    def local_rows(self):
        return self.hi - self.lo

    def close(self):
        for e in self.engines:
            e.close()
        self.parts, self.engines = [], []

    def local(self, arr):
        """This rank's rows of a global (batch, ...) array."""
        if arr.shape[0] != self.batch:
            raise ValueError("expected a global array of %d rows, got %d" % (self.batch, arr.shape[0]))
        return arr[self.lo:self.hi]

    def _rows(self, arr, name, global_rows):
        if arr is None:
            return [None] * len(self.parts)
        want = self.batch if global_rows else self.local_rows
        if arr.shape[0] != want:
            raise ValueError("%s: expected %d rows (%s), got %d" % (name, want, "global" if global_rows else "this rank's", arr.shape[0]))
        off = 0 if global_rows else self.lo
        return [arr[a - off:b - off] for a, b, _ in self.parts]

    def set_fields(self, fields, counts, global_rows=True):
        """Field sets of the whole batch (global_rows) or of this rank's rows only (a caller that never holds the
        global arrays, e.g. bench.py's per-rank synthetic workload)."""
        for (a, b, e), f, c in zip(self.parts, self._rows(fields, "fields", global_rows), self._rows(counts, "counts", global_rows)):
            e.set_fields(f, c)

    # the columns of every output row (Engine._OUT_SHAPES): what a rank without a single arm still has to return
    _COLS = {"qdot_vf": "n", "qdot_null": "n", "qdot_out": "n", "pose": 16, "pose_nt": 16, "v6": 6, "qdist": "n", "goal_dist": 2,
             "q_ref_out": "n", "track_error": 8}

    def _out_arrays(self, want, into, dtype):
        """This rank's output arrays (local_rows x columns), freshly zeroed or the caller's `into` arrays, and per part the views
        of its rows: a part writes straight into its rows, nothing is concatenated afterwards."""
        import numpy as np
        full = {}
        for k in want:
            if into is not None and k in into:
                full[k] = into[k]
                if full[k].shape[0] != self.local_rows:
                    raise ValueError("into[%s]: expected %d rows (this rank's), got %d" % (k, self.local_rows, full[k].shape[0]))
            elif k == "status":
                full[k] = np.zeros(self.local_rows, dtype=np.int32)
            else:
                c = self._COLS[k]
                full[k] = np.zeros((self.local_rows, self.chain.n if c == "n" else c), dtype=dtype)
        views = [{k: v[a - self.lo:b - self.lo] for k, v in full.items()} for a, b, _ in self.parts]
        return full, views

    def _pinned(self, i, key, shape, dtype):
        """Pinned staging array of part i (Engine.host_array), kept while its shape fits: the pipelined host path is asynchronous
        only from / to pinned memory -- a copy from pageable memory blocks the submitting thread until it is done."""
        slot = self._pin.setdefault(i, {})
        arr = slot.get(key)
        if arr is None or arr.shape != tuple(shape) or arr.dtype != dtype:
            arr = slot[key] = self.engines[i].host_array(tuple(shape), dtype)
        return arr

    def step_host(self, q, null_control=None, want=("qdot_out",), global_rows=True, into=None, **kw):
        """One control cycle of this rank's arms: host arrays in (global or local rows), this rank's rows out.
        Per-arm keyword arrays of Engine.step_host (q_ref, q_cmded, active, q_lo, q_hi) are sliced the same way; `into`
        (a dict of this rank's output arrays from an earlier call) is written in place, so the rows of arms gated off by
        `active` keep their content.

        The rank's devices run concurrently, from this one thread: every part's inputs are staged into ITS pinned arrays and
        the cycle is SUBMITTED (Engine.submit_host: copy engine / kernel / copy engine on the part's own streams) to every
        device before the first one is waited for; then the outputs are copied out of the pinned arrays.  (Handles without the
        pipelined path -- the CPU tests' stand-in -- go through one thread per device around the blocking step_host.)  A rank
        that owns no arm (more ranks than arms) returns arrays of zero rows."""
        import numpy as np
        qs = self._rows(q, "q", global_rows)
        ncs = self._rows(null_control, "null_control", global_rows)
        kws = {k: self._rows(v, k, global_rows) for k, v in kw.items() if v is not None}
        dtype = getattr(self.engines[0], "io_dtype", np.float64) if self.engines else (q.dtype if hasattr(q, "dtype") else np.float64)
        full, views = self._out_arrays(tuple(want), into, dtype)
        if not self.parts:
            return full
        pipelined = all(hasattr(e, "submit_host") and hasattr(e, "host_array") for e in self.engines) and \
            set(kws) <= {"q_ref", "q_cmded", "active", "q_lo", "q_hi"} and all(k != "obj_dist" and k != "track_error" for k in want)
        if pipelined:
            tickets, staged = [], []
            gated = "active" in kws
            for i, (a, b, e) in enumerate(self.parts):           # every device gets its work ...
                args = {}
                for key, rows in [("q", qs[i]), ("null_control", ncs[i])] + [(k, v[i]) for k, v in kws.items()]:
                    if rows is None:
                        continue
                    dt = np.int32 if key == "active" else e.io_dtype
                    rows = (np.asarray(rows) != 0) if key == "active" else rows
                    pin = self._pinned(i, "in_" + key, np.shape(rows), np.dtype(dt))
                    pin[...] = rows
                    args[key] = pin
                outs = {}
                for k, v in views[i].items():
                    outs[k] = self._pinned(i, "out_" + k, v.shape, v.dtype)
                    if gated:
                        outs[k][...] = v          # rows of gated arms come back as they went in
                staged.append(outs)
                tickets.append(e.submit_host(args.pop("q"), outs, **args))
            for (a, b, e), t in zip(self.parts, tickets):        # ... before the first one is waited for
                e.wait(t)
            for outs, view in zip(staged, views):
                for k, v in view.items():
                    v[...] = outs[k]
            return full
        if len(self.parts) == 1:
            (a, b, e), = self.parts
            e.step_host(qs[0], null_control=ncs[0], want=want, into=views[0], **{k: v[0] for k, v in kws.items()})
            return full
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(self.parts)) as pool:
            futs = [pool.submit(e.step_host, qs[i], null_control=ncs[i], want=want, into=views[i], **{k: v[i] for k, v in kws.items()})
                    for i, (a, b, e) in enumerate(self.parts)]
            for f in futs:
                f.result()
        return full

    def gather(self, local_rows, dist=None):
        """The rows of every rank, in arm order, on every rank (torch tensor in, torch tensor out)."""
        if self.world == 1:
            return local_rows
        return collate(local_rows, self.batch, dist)

    def step_global(self, q, want=("qdot_out",), dist=None, **kw):
        """step_host on this rank's rows of the global q, then the collated global rows of every output."""
        import torch
        out = self.step_host(q, want=want, **kw)
        return {k: self.gather(torch.from_numpy(v), dist).numpy() for k, v in out.items()}
