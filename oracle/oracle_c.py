"""ctypes binding of oracle/libvfik_oracle.so.  TEST INFRASTRUCTURE (see vfik_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from vfclik_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class State(C.Structure):
    _fields_ = [("lastvec", C.c_double * (_abi.MAX_JOINTS * _abi.MAX_JOINTS)), ("sig", C.c_int * _abi.MAX_JOINTS)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libvfik_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvfik_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.vfo_max_threads.restype = C.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def new_states(B, n):
    st = (State * B)()
    for s in st:
        lib().vfo_state_init(C.byref(s), n)
    return st


def cycle_batch(chain, params, q, fields, nfields, tool=None, null_control=None, ext_cmd=None, states=None,
                want=("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status"), nthreads=0,
                q_lo=None, q_hi=None, active=None, into=None):
    """chain: vfclik_amd.chain.Chain; params: _abi.Params; q (B,n) f64; fields (B,M) FIELD_DTYPE;
    nfields (B,) i32; tool (16,) or (B,16); ext_cmd (4,B,n); q_lo / q_hi (B,n): this cycle's limits per arm;
    active (B,): arms with 0 are skipped (rows of `into`, a dict from an earlier call, stay).  Returns dict of arrays."""
    L = lib()
    q = np.ascontiguousarray(q, dtype=np.float64)
    B, n = q.shape
    assert n == chain.n
    fields = np.ascontiguousarray(fields, dtype=_abi.FIELD_DTYPE)
    M = fields.shape[1]
    nfields = np.ascontiguousarray(nfields, dtype=np.int32)
    if tool is None:
        tool = np.eye(4).reshape(16)
    tool = np.ascontiguousarray(tool, dtype=np.float64)
    tstride = 16 if tool.ndim == 2 else 0
    nc = None if null_control is None else np.ascontiguousarray(null_control, dtype=np.float64)
    ec = None if ext_cmd is None else np.ascontiguousarray(ext_cmd, dtype=np.float64)
    if (params.flags & _abi.F_NULLSPACE) and states is None:
        states = new_states(B, n)
    shapes = {"qdot_vf": (B, n), "qdot_null": (B, n), "qdot_out": (B, n), "pose": (B, 16), "pose_nt": (B, 16),
              "v6": (B, 6), "qdist": (B, n)}
    out = {k: (into[k] if into is not None and k in into else np.zeros(s)) for k, s in shapes.items() if k in want}
    if "status" in want:
        out["status"] = into["status"] if into is not None and "status" in into else np.zeros(B, dtype=np.int32)
    lo = None if q_lo is None else np.ascontiguousarray(q_lo, dtype=np.float64)
    hi = None if q_hi is None else np.ascontiguousarray(q_hi, dtype=np.float64)
    assert (lo is None) == (hi is None) and (lo is None or (lo.shape == (B, n) and hi.shape == (B, n)))
    act = None if active is None else np.ascontiguousarray(np.asarray(active) != 0, dtype=np.int32)
    cs = chain.to_struct()
    L.vfo_cycle_batch(C.byref(cs), C.byref(params), C.c_int(B), _p(tool), C.c_int(tstride), _p(fields), C.c_int(M),
                      _p(nfields), _p(q), _p(nc), _p(ec), states if states is not None else None,
                      _p(out.get("qdot_vf")), _p(out.get("qdot_null")), _p(out.get("qdot_out")), _p(out.get("pose")),
                      _p(out.get("pose_nt")), _p(out.get("v6")), _p(out.get("qdist")), _p(out.get("status")),
                      C.c_int(nthreads), _p(lo), _p(hi), _p(act))
    out["states"] = states
    return out


def mix(cmd, w):
    cmd = np.ascontiguousarray(cmd, dtype=np.float64)
    K, n = cmd.shape
    w = np.ascontiguousarray(w, dtype=np.float64)
    out = np.zeros(n)
    lib().vfo_mix(_p(cmd), _p(w), C.c_int(K), C.c_int(n), _p(out))
    return out


def probe_field(params, fields, nfields, poses, speed=None):
    """vf:469-503: speedScale * scalars * normCart(sum) of every arm's field set at the given poses (B,16)."""
    fields = np.ascontiguousarray(fields, dtype=_abi.FIELD_DTYPE)
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    B, M = fields.shape
    out = np.zeros((B, 6))
    vec6, sc = np.zeros(6), np.zeros(2)
    for b in range(B):
        lib().vfo_field_eval(_p(fields[b]), C.c_int(int(nfields[b])), _p(poses[b, :12].copy()), C.c_double(params.rot_slowdown), _p(vec6), _p(sc))
        s = params.speed_scale if speed is None else speed[b]
        out[b, :3] = s * sc[0] * vec6[:3]
        out[b, 3:] = s * sc[1] * vec6[3:]
    return out


def joint_p(ref, q, lo, hi, kp, delta):
    """(B,n) references and angles -> (kp*(clamp(ref)-q), at_goal flags)."""
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    B, n = q.shape
    out = np.zeros((B, n))
    flags = np.zeros(B, dtype=np.int32)
    fn = lib().vfo_joint_p
    fn.restype = C.c_int
    for b in range(B):
        flags[b] = fn(_p(ref[b]), _p(q[b]), _p(lo), _p(hi), C.c_int(n), C.c_double(kp), C.c_double(delta), _p(out[b]))
    return out, flags


def lwr_cmd(qdot_lim, q, q_cmded, direct):
    qdot_lim = np.ascontiguousarray(qdot_lim, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    q_cmded = np.ascontiguousarray(q_cmded, dtype=np.float64)
    B, n = q.shape
    out = np.zeros((B, n))
    for b in range(B):
        lib().vfo_lwr_cmd(_p(qdot_lim[b]), _p(q[b]), _p(q_cmded[b]), C.c_int(n), C.c_int(int(direct[b])), _p(out[b]))
    return out


def restrict(J):
    J = np.ascontiguousarray(J, dtype=np.float64)
    n = J.shape[1]
    Bm = np.zeros((n, n))
    lib().vfo_restrict(_p(J), C.c_int(n), _p(Bm))
    return Bm


class NullspaceC:
    """Stateful wrapper of vfo_nullspace_basis / vfo_move_in_nullspace."""

    def __init__(self, n):
        self.n = n
        self.lastvec = np.zeros((n, n))
        self.sig = np.ones(n, dtype=np.int32)

    def basis(self, J):
        J = np.ascontiguousarray(J, dtype=np.float64)
        out = np.zeros((self.n, self.n))
        lib().vfo_nullspace_basis.restype = C.c_int
        r = lib().vfo_nullspace_basis(_p(J), C.c_int(self.n), _p(self.lastvec), _p(self.sig), _p(out))
        return out[:r]

    def move(self, J, control):
        J = np.ascontiguousarray(J, dtype=np.float64)
        control = np.ascontiguousarray(control, dtype=np.float64)
        qd = np.zeros(self.n)
        rk = C.c_int(0)
        lib().vfo_move_in_nullspace(_p(J), C.c_int(self.n), _p(control), C.c_int(len(control)), _p(self.lastvec),
                                    _p(self.sig), _p(qd), C.byref(rk))
        return qd, rk.value


def check_limits(q, qdot, lo, hi, scale=0.3):
    q = np.ascontiguousarray(q, dtype=np.float64)
    qd = np.array(qdot, dtype=np.float64)
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    lib().vfo_check_limits.restype = C.c_int
    t = lib().vfo_check_limits(_p(q), _p(qd), _p(lo), _p(hi), C.c_int(len(q)), C.c_double(scale))
    return qd, bool(t)


def jacobian(chain, q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    J = np.zeros((6, chain.n))
    T = np.zeros(12)
    cs = chain.to_struct()
    lib().vfo_jacobian(C.byref(cs), _p(q), _p(J), _p(T))
    return J, T.reshape(3, 4)


def max_threads():
    return lib().vfo_max_threads()
