"""Kinematic chain description -- the part of ``Lafik(config)`` (reference scripts/vf:153) that the
reference hides in arcospyu and a per-robot config file (neither is in the reference tree).

A chain is stored in *z-normal form* (include/vfik_types.h):

    T_ee(q) = B[0] * Jz(q_1) * B[1] * ... * Jz(q_n) * B[n]

which is what the HIP kernel evaluates.  KDL-style segment lists (joint about an arbitrary axis,
then a fixed tip frame: ``Segment(Joint(axis), Frame)``) and DH tables are converted here, on the
host, once.
"""
import math

import numpy as np

from . import _abi

REVOLUTE, PRISMATIC = 0, 1


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=float)


def _rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=float)


def _hom(R=None, p=None):
    T = np.eye(4)
    if R is not None:
        T[:3, :3] = R
    if p is not None:
        T[:3, 3] = p
    return T


def _align_z_to(axis):
    """Rotation C with C @ e_z = axis/|axis|."""
    a = np.asarray(axis, dtype=float)
    a = a / np.linalg.norm(a)
    ez = np.array([0.0, 0.0, 1.0])
    v = np.cross(ez, a)
    s, c = np.linalg.norm(v), float(ez @ a)
    if s < 1e-15:
        return np.eye(3) if c > 0 else np.diag([1.0, -1.0, -1.0])
    vx = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + vx + vx @ vx * ((1 - c) / (s * s))


class Chain:
    def __init__(self, B, jtype, q_lo, q_hi, name="chain"):
        self.B = np.ascontiguousarray(np.asarray(B, dtype=np.float64).reshape(-1, 3, 4))
        self.n = self.B.shape[0] - 1
        if not 1 <= self.n <= _abi.MAX_JOINTS:
            raise ValueError("chain must have 1..%d joints, got %d" % (_abi.MAX_JOINTS, self.n))
        self.jtype = [int(t) for t in jtype]
        self.q_lo = np.asarray(q_lo, dtype=np.float64).copy()
        self.q_hi = np.asarray(q_hi, dtype=np.float64).copy()
        if not (len(self.jtype) == len(self.q_lo) == len(self.q_hi) == self.n):
            raise ValueError("jtype / q_lo / q_hi must have n entries")
        if not np.all(self.q_hi > self.q_lo):
            raise ValueError("every joint needs q_hi > q_lo")
        self.name = name

    # -- constructors ------------------------------------------------------------------------
    @classmethod
    def from_dh(cls, dh, q_lo, q_hi, base=None, name="dh"):
        """Standard DH rows (a, alpha, d, theta_offset): T_i = Rz(q+off) Tz(d) Tx(a) Rx(alpha)."""
        Bs = [np.eye(4) if base is None else np.asarray(base, dtype=float).reshape(4, 4)]
        for (a, alpha, d, off) in dh:
            Bs[-1] = Bs[-1] @ _hom(_rot_z(off))
            Bs.append(_hom(None, [0, 0, d]) @ _hom(None, [a, 0, 0]) @ _hom(_rot_x(alpha)))
        return cls([T[:3, :] for T in Bs], [REVOLUTE] * len(dh), q_lo, q_hi, name)

    @classmethod
    def from_segments(cls, segments, q_lo, q_hi, base=None, name="segments"):
        """KDL-style: segments = [(joint_type, axis3, f_tip4x4)], pose_i(q) = Joint(axis, q) * f_tip."""
        Bs = [np.eye(4) if base is None else np.asarray(base, dtype=float).reshape(4, 4)]
        jt = []
        for (jtype, axis, f_tip) in segments:
            Cm = _hom(_align_z_to(axis))
            Bs[-1] = Bs[-1] @ Cm
            Bs.append(Cm.T @ np.asarray(f_tip, dtype=float).reshape(4, 4))
            jt.append(int(jtype))
        return cls([T[:3, :] for T in Bs], jt, q_lo, q_hi, name)

    def concat(self, other, name=None):
        """Serial composition: self's flange carries other's base (C5: two 7-DOF chains, one flange)."""
        A = np.vstack([self.B[-1], [0, 0, 0, 1]]) @ np.vstack([other.B[0], [0, 0, 0, 1]])
        B = list(self.B[:-1]) + [A[:3, :]] + list(other.B[1:])
        return Chain(B, self.jtype + other.jtype, np.concatenate([self.q_lo, other.q_lo]),
                     np.concatenate([self.q_hi, other.q_hi]), name or (self.name + "+" + other.name))

    # -- C view -----------------------------------------------------------------------------
    def to_struct(self):
        s = _abi.Chain()
        s.n = self.n
        for i in range(self.n):
            s.jtype[i] = self.jtype[i]
            s.q_lo[i] = float(self.q_lo[i])
            s.q_hi[i] = float(self.q_hi[i])
        for i in range(self.n + 1):
            flat = self.B[i].reshape(12)
            for k in range(12):
                s.B[i][k] = float(flat[k])
        return s

    # -- host-side forward kinematics (setup utilities only; the control path runs on the GPU) --
    def fk(self, q):
        """Batched FK on the host for building synthetic goals / initial poses: q (B,n) -> (B,4,4)."""
        q = np.atleast_2d(np.asarray(q, dtype=np.float64))
        Bn = q.shape[0]
        X = np.tile(np.vstack([self.B[0], [0, 0, 0, 1]]), (Bn, 1, 1))
        for i in range(self.n):
            Jz = np.tile(np.eye(4), (Bn, 1, 1))
            if self.jtype[i] == REVOLUTE:
                c, s = np.cos(q[:, i]), np.sin(q[:, i])
                Jz[:, 0, 0], Jz[:, 0, 1], Jz[:, 1, 0], Jz[:, 1, 1] = c, -s, s, c
            else:
                Jz[:, 2, 3] = q[:, i]
            X = X @ Jz @ np.vstack([self.B[i + 1], [0, 0, 0, 1]])
        return X
