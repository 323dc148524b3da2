#!/usr/bin/env python
"""Error budget of a float32 distance law for the decay repellers (Next 3: "float32 for the repellers' distance law
with d = o - p kept in float64 ... accept only if max_abs_err stays < 1e-6").  CPU model, no GPU: for the C3 workload, the twist
v = speed * S0 * normCart(attractor + sum_k force_k m_k d_k / D_k) with the magnitude chain m_k / D_k = ((r + s) / D)^5 / D evaluated
(a) in float64 and (b) in float32 (d = o - p and the sum in float64, as proposed); the joint velocity error follows through the
damped least-squares map qdot = J^T (J J^T + lambda^2 I)^-1 v of every arm.  Prints max |dv| and max |dqdot|."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vfclik_amd import robots, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
chain = robots.lwr()
w = synth.make_workload(chain, B, 8, seed=1, io_dtype=np.float32)
T = chain.fk(w["q"])
p = T[:, :3, 3]
F = w["fields"]
goal = F["p"][:, 0, :16].reshape(B, 4, 4)
dg = goal[:, :3, 3] - p
Dg = np.linalg.norm(dg, axis=1)
tot0 = dg / Dg[:, None]          # attractor, force 1
S0 = np.minimum(1.0, Dg / F["p"][:, 0, 16])


def field(dtype):
    tot = tot0.copy()
    for k in range(1, 9):
        d = F["p"][:, k, 0:3] - p                       # float64, as proposed
        D2 = (d * d).sum(1)
        rs = (F["p"][:, k, 3] + F["p"][:, k, 4]).astype(dtype)
        di = (1.0 / np.sqrt(D2.astype(dtype))).astype(dtype)    # the law in `dtype`
        rb = (rs * di).astype(dtype)
        b2 = (rb * rb).astype(dtype)
        rp = ((b2 * b2).astype(dtype) * rb).astype(dtype)
        kk = (F["force"][:, k].astype(dtype) * np.minimum(rp, dtype(1e6)) * di).astype(dtype)
        tot += d * kk.astype(np.float64)[:, None]
    n = np.linalg.norm(tot, axis=1)
    return tot / n[:, None] * S0[:, None]


v64, v32 = field(np.float64), field(np.float32)
dv = np.abs(v64 - v32).max()
print("arms %d: max |dv| = %.3e m/s" % (B, dv))
print("bound through the DLS map: |dqdot| <= |dv| / (2 lambda) = %.3e rad/s (lambda = 0.1)" % (dv / 0.2))
# ... and the map itself, for the arms with the largest twist errors (the oracle's Jacobian: test infrastructure, fine in a tool)
from oracle import vfik_numpy as vn  # noqa: E402
lam2, worst = 0.01, 0.0
order = np.argsort(-np.abs(v64 - v32).max(axis=1))[:2000]
for b in order:
    rob = vn.Lafik(chain.B, chain.jtype, chain.q_lo, chain.q_hi)
    rob.jntsList = w["q"][b].tolist()
    Jb = np.array(rob.jac_list()).reshape(6, chain.n)
    A = Jb @ Jb.T + lam2 * np.eye(6)
    dq = Jb.T @ np.linalg.solve(A, np.concatenate([v64[b] - v32[b], np.zeros(3)]))
    worst = max(worst, float(np.abs(dq).max()))
print("max |dqdot| through the DLS map of the 2 000 arms with the largest twist error = %.3e rad/s" % worst)
print("budget: 1e-6 rad/s in all, of which the float32 store of qdot already takes up to 2.4e-7 (measured max on C3)")
