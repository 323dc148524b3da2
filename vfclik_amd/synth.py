"""Seeded synthetic workloads of BASELINE.json's configs (distributions: SURVEY.md section 8d).

Values come from the reference where it holds any: slow-down distance 0.05
(old/system_start.sh.old:346), obstacle radius around 0.05 (old/README.old:75), safeDist 0.001 and
decay order 5 (scripts/object_feeder:301-302,331), forces +1 / -10 (object_feeder:235,323), field ids
1 and 4+k (object_feeder:234,322), identity tool (scripts/vf:154).
"""
import numpy as np

from . import _abi


def make_workload(chain, batch, n_obstacles, seed=0, io_dtype=np.float32, max_fields=None):
    """Returns dict(q, fields, nfields, tool) with inputs already rounded to ``io_dtype``."""
    rng = np.random.default_rng(seed)
    n = chain.n
    lo, hi = 0.8 * chain.q_lo, 0.8 * chain.q_hi
    q = rng.uniform(lo, hi, size=(batch, n)).astype(io_dtype).astype(np.float64)
    qg = rng.uniform(lo, hi, size=(batch, n))
    goal = chain.fk(qg).reshape(batch, 16).astype(io_dtype).astype(np.float64)
    M = 1 + n_obstacles if max_fields is None else max_fields
    fields = np.zeros((batch, M), dtype=_abi.FIELD_DTYPE)
    fields["id"][:, 0] = 1
    fields["type"][:, 0] = _abi.FIELD_ATTRACTOR
    fields["force"][:, 0] = 1.0
    fields["p"][:, 0, :16] = goal
    fields["p"][:, 0, 16] = np.float64(io_dtype(0.05))
    for k in range(n_obstacles):
        pos = np.stack([rng.uniform(-0.8, 0.8, batch), rng.uniform(-0.8, 0.8, batch), rng.uniform(0.0, 1.2, batch)], 1)
        rad = rng.uniform(0.03, 0.10, batch)
        f = fields[:, 1 + k]
        f["id"] = 4 + k
        f["type"] = _abi.FIELD_REPELLER
        f["force"] = -10.0
        f["p"][:, 0:3] = pos.astype(io_dtype)
        f["p"][:, 3] = rad.astype(io_dtype)
        f["p"][:, 4] = np.float64(io_dtype(0.001))
        f["p"][:, 5] = 5.0
    nfields = np.full(batch, 1 + n_obstacles, dtype=np.int32)
    tool = np.eye(4).reshape(16)
    return dict(q=q, fields=fields, nfields=nfields, tool=tool)
