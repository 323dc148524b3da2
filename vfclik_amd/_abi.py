"""ctypes mirror of include/vfik_types.h and loader of the HIP shared library.

The product path has no CPU fallback: if ``libvfik_hip.so`` is missing or does not export a
symbol declared in include/vfik.h, importing the engine raises.
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 5  # include/vfik.h: VFIK_ABI_VERSION
MAX_JOINTS = 16
MAX_PARAMS = 17
MIX_CHANNELS = 6
NULL_CONTROLS = 4

FIELD_NULL, FIELD_ATTRACTOR, FIELD_REPELLER, FIELD_HEMISPHERE, FIELD_FUNNEL = 0, 1, 2, 4, 5
FIELD_NPARAMS = {FIELD_NULL: 0, FIELD_ATTRACTOR: 17, FIELD_REPELLER: 6, FIELD_HEMISPHERE: 8, FIELD_FUNNEL: 10}

F_NULLSPACE, F_JOINT_LIMIT_TASK, F_MIXER, F_LIMITER = 1, 2, 4, 8
ST_NAN, ST_LIMIT_STOP, ST_NULL_AMBIGUOUS, ST_LIMITED, ST_JOINT_AT_GOAL = 1, 2, 4, 8, 16

#: numpy view of ``struct vfik_field`` (152 bytes)
FIELD_DTYPE = np.dtype([("id", "<i4"), ("type", "<i4"), ("force", "<f8"), ("p", "<f8", (MAX_PARAMS,))])
assert FIELD_DTYPE.itemsize == 152


class Field(C.Structure):
    _fields_ = [("id", C.c_int32), ("type", C.c_int32), ("force", C.c_double), ("p", C.c_double * MAX_PARAMS)]


class Chain(C.Structure):
    _fields_ = [("n", C.c_int32), ("jtype", C.c_int32 * MAX_JOINTS),
                ("B", (C.c_double * 12) * (MAX_JOINTS + 1)),
                ("q_lo", C.c_double * MAX_JOINTS), ("q_hi", C.c_double * MAX_JOINTS)]


class Params(C.Structure):
    _fields_ = [("speed_scale", C.c_double), ("lambda_", C.c_double), ("rot_slowdown", C.c_double),
                ("null_gain", C.c_double), ("lookahead", C.c_double), ("jl_gain", C.c_double),
                ("max_vel", C.c_double), ("wy", C.c_double * 6), ("wq", C.c_double * MAX_JOINTS),
                ("mix_w", C.c_double * MIX_CHANNELS), ("flags", C.c_uint32), ("reserved", C.c_uint32),
                ("jp_kp", C.c_double), ("jp_delta", C.c_double)]


def default_params(**kw):
    """Defaults: speedScale 1.0 (vf:136), nullspace gain 0.5 / look-ahead 0.3 (nullspace:62,121),
    mixer weights [1,1,0,0,0,0] (bridge:596); lambda / rot_slowdown / jl_gain are build-defined."""
    p = Params()
    p.speed_scale, p.lambda_, p.rot_slowdown = 1.0, 0.1, 0.3
    p.null_gain, p.lookahead, p.jl_gain, p.max_vel = 0.5, 0.3, 0.5, 1.0
    p.jp_kp, p.jp_delta = 1.5, 0.087  # joint_p_controller:55,57
    for i in range(6):
        p.wy[i] = 1.0
    for i in range(MAX_JOINTS):
        p.wq[i] = 1.0
    for i, w in enumerate([1.0, 1.0, 0.0, 0.0, 0.0, 0.0]):
        p.mix_w[i] = w
    p.flags = 0
    for k, v in kw.items():
        if k == "lambda":
            k = "lambda_"
        if k in ("wy", "wq", "mix_w"):
            arr = getattr(p, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
        else:
            setattr(p, k, v)
    return p


def params_to_dict(p):
    return {"speed_scale": p.speed_scale, "lambda": p.lambda_, "rot_slowdown": p.rot_slowdown,
            "null_gain": p.null_gain, "lookahead": p.lookahead, "jl_gain": p.jl_gain, "max_vel": p.max_vel,
            "wy": list(p.wy), "wq": list(p.wq), "mix_w": list(p.mix_w), "flags": int(p.flags)}


def pkg_dir():
    return os.path.dirname(os.path.abspath(__file__))


HIP_LIB_PATH = os.path.join(pkg_dir(), "csrc", "libvfik_hip.so")
