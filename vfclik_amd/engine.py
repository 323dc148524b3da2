"""ctypes binding of libvfik_hip.so (include/vfik.h) -- the only compute path of this package.

``Engine`` owns one ``vfik_handle``: the batched state of the reference's per-arm vf / nullspace /
debug_jointlimits / bridge-mixer processes on one GPU.  It fails loudly when the HIP library is
missing or no GPU is visible; nothing here computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from .chain import Chain


class VfikError(RuntimeError):
    pass


class IO(C.Structure):
    _fields_ = [("q", C.c_void_p), ("null_control", C.c_void_p), ("qdot_vf", C.c_void_p), ("qdot_null", C.c_void_p),
                ("qdot_out", C.c_void_p), ("pose", C.c_void_p), ("pose_nt", C.c_void_p), ("v6", C.c_void_p),
                ("qdist", C.c_void_p), ("status", C.c_void_p), ("goal_dist", C.c_void_p),
                ("q_ref", C.c_void_p), ("q_cmded", C.c_void_p),
                ("active", C.c_void_p), ("q_lo", C.c_void_p), ("q_hi", C.c_void_p), ("q_ref_out", C.c_void_p),
                ("track_error", C.c_void_p), ("obj_dist", C.c_void_p)]


_OUT_SHAPES = {"qdot_vf": "n", "qdot_null": "n", "qdot_out": "n", "pose": 16, "pose_nt": 16, "v6": 6, "qdist": "n",
               "goal_dist": 2, "q_ref_out": "n", "track_error": 8}
_lib = None


def load_library(path=None):
    """Load libvfik_hip.so and declare the prototypes of include/vfik.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("VFIK_HIP_LIB") or _abi.HIP_LIB_PATH
    try:
        # torch ships its own libamdhip64; let it load first so that this process holds ONE HIP
        # runtime (loading /opt/rocm's copy first makes a later torch.cuda init fail)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise VfikError("HIP library %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    H = C.c_void_p
    protos = {
        "vfik_abi_version": (C.c_int, []),
        "vfik_struct_sizes": (None, [C.POINTER(C.c_size_t)]),
        "vfik_last_error": (C.c_char_p, []),
        "vfik_device_count": (C.c_int, []),
        "vfik_supported_joints": (C.c_uint32, []),
        "vfik_create": (H, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "vfik_destroy": (None, [H]),
        "vfik_set_stream": (C.c_int, [H, C.c_void_p]),
        "vfik_set_chain": (C.c_int, [H, C.POINTER(_abi.Chain)]),
        "vfik_set_params": (C.c_int, [H, C.POINTER(_abi.Params)]),
        "vfik_set_tool": (C.c_int, [H, C.c_void_p, C.c_int]),
        "vfik_set_speed_scale": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p]),
        "vfik_set_fields": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
        "vfik_set_mixer_weights": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p]),
        "vfik_set_ext_cmd": (C.c_int, [H, C.c_int, C.c_void_p]),
        "vfik_reset_state": (C.c_int, [H]),
        "vfik_step": (C.c_int, [H, C.POINTER(IO)]),
        "vfik_step_host": (C.c_int, [H, C.POINTER(IO)]),
        "vfik_sync": (C.c_int, [H]),
        "vfik_rollout": (C.c_int, [H, C.POINTER(IO), C.c_int, C.c_double, C.c_int, C.c_void_p]),
        "vfik_rollout_host": (C.c_int, [H, C.POINTER(IO), C.c_int, C.c_double, C.c_int, C.c_void_p]),
        "vfik_mix": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
        "vfik_track_error": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "vfik_track_reset": (C.c_int, [H]),
        "vfik_dev_alloc": (C.c_void_p, [H, C.c_size_t]),
        "vfik_dev_free": (C.c_int, [H, C.c_void_p]),
        "vfik_memcpy_h2d": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_size_t]),
        "vfik_memcpy_d2h": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_size_t]),
        "vfik_time_steps": (C.c_int, [H, C.POINTER(IO), C.c_int, C.c_int, C.POINTER(C.c_float)]),
        "vfik_slots_in_use": (C.c_int, [H]),
        "vfik_field_path": (C.c_int, [H]),
        "vfik_uniform_repellers": (C.c_int, [H]),
        "vfik_mixed_orders": (C.c_int, [H]),
        "vfik_launch_epoch": (C.c_long, [H]),
        "vfik_dh_pattern": (C.c_int, [H]),
        "vfik_device_bytes": (C.c_size_t, [H]),
        "vfik_host_alloc": (C.c_void_p, [H, C.c_size_t]),
        "vfik_host_free": (C.c_int, [H, C.c_void_p]),
        "vfik_submit_host": (C.c_int, [H, C.POINTER(IO), C.POINTER(C.c_long)]),
        "vfik_wait": (C.c_int, [H, C.c_long]),
        "vfik_set_arm_weights": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
        "vfik_set_max_vel": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p]),
        "vfik_probe_field": (C.c_int, [H, C.c_void_p, C.c_void_p]),
        "vfik_object_distances": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
        "vfik_set_objects": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p, C.c_int]),
        "vfik_set_small_batch_kernel": (C.c_int, [H, C.c_int]),
        "vfik_small_batch_launches": (C.c_long, [H]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not match include/vfik.h
        fn.restype = res
        fn.argtypes = args
    if lib.vfik_abi_version() != _abi.ABI_VERSION:
        raise VfikError("ABI version mismatch: library %d, binding %d" % (lib.vfik_abi_version(), _abi.ABI_VERSION))
    sizes = (C.c_size_t * 4)()
    lib.vfik_struct_sizes(sizes)
    mine = [C.sizeof(_abi.Field), C.sizeof(_abi.Chain), C.sizeof(_abi.Params), C.sizeof(IO)]
    if list(sizes) != mine:
        raise VfikError("struct layout mismatch: library %s, Python mirrors %s" % (list(sizes), mine))
    if path == _abi.HIP_LIB_PATH or _lib is None:
        _lib = lib
    return lib


def _ptr(x):
    """Device / host address of a torch tensor, numpy array or raw int."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()  # torch.Tensor


class Engine:
    def __init__(self, chain, batch, io_dtype=np.float32, max_slots=16, device=0, params=None):
        if not isinstance(chain, Chain):
            raise TypeError("chain must be a vfclik_amd.chain.Chain")
        self.lib = load_library()
        self.chain = chain
        self.n = chain.n
        self.batch = int(batch)
        self.io_dtype = np.dtype(io_dtype)
        if self.io_dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("io_dtype must be float32 or float64")
        self.max_slots = int(max_slots)
        self.device = int(device)
        bits = 32 if self.io_dtype == np.float32 else 64
        self._pinned = []
        self.h = self.lib.vfik_create(self.device, bits, self.n, self.max_slots, self.batch)
        if not self.h:
            raise VfikError("vfik_create: " + self.lib.vfik_last_error().decode())
        self._chk(self.lib.vfik_set_chain(self.h, C.byref(chain.to_struct())))
        self.params = params if params is not None else _abi.default_params()
        self._chk(self.lib.vfik_set_params(self.h, C.byref(self.params)))
        self._torch_out = {}
        self.n_objects = 0  # object frames of the distance monitor held by the handle (set_objects)

    # -- plumbing -------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise VfikError("vfik error %d: %s" % (rc, self.lib.vfik_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.vfik_sync(self.h)
            for p in self._pinned:  # arrays handed out by host_array() die with the engine
                self.lib.vfik_host_free(self.h, C.c_void_p(p))
            self._pinned = []
            self.lib.vfik_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_stream(self, stream_ptr):
        """Launch on the caller's HIP stream (``torch.cuda.current_stream().cuda_stream``)."""
        self._chk(self.lib.vfik_set_stream(self.h, C.c_void_p(stream_ptr)))

    # -- slow-changing state (event-driven in the reference) --------------------------------------
    def set_params(self, **kw):
        for k, v in kw.items():
            k = "lambda_" if k == "lambda" else k
            if k in ("wy", "wq", "mix_w"):
                arr = getattr(self.params, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                setattr(self.params, k, v)
        self._chk(self.lib.vfik_set_params(self.h, C.byref(self.params)))

    def set_tool(self, tool16, per_arm=False):
        t = np.ascontiguousarray(tool16, dtype=np.float64)
        if t.size != (16 * self.batch if per_arm else 16):
            raise ValueError("tool must hold 16 doubles%s" % (" per arm" if per_arm else ""))
        self._chk(self.lib.vfik_set_tool(self.h, t.ctypes.data, 1 if per_arm else 0))

    def set_speed_scale(self, values, first_arm=0):
        """Per-arm speedScale (the /max_vel value each arm's vf keeps, vf:197-207)."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        self._chk(self.lib.vfik_set_speed_scale(self.h, int(first_arm), len(v), v.ctypes.data))

    def set_fields(self, fields, counts, first_arm=0):
        f = np.ascontiguousarray(fields, dtype=_abi.FIELD_DTYPE)
        if f.ndim != 2:
            raise ValueError("fields must be (n_arms, max_fields)")
        c = np.ascontiguousarray(counts, dtype=np.int32)
        self._chk(self.lib.vfik_set_fields(self.h, int(first_arm), f.shape[0], f.ctypes.data, f.shape[1], c.ctypes.data))

    def set_objects(self, frames, first_arm=0):
        """Object frames of the distance monitor (monitor_distance:72,111-129): (n_arms, n_objects, 16) doubles, kept on the
        device; a cycle call that asks for ``obj_dist`` then gets every arm's distance / angle to its objects."""
        f = np.ascontiguousarray(frames, dtype=np.float64)
        if f.ndim != 3 or f.shape[2] != 16:
            raise ValueError("frames must be (n_arms, n_objects, 16)")
        self._chk(self.lib.vfik_set_objects(self.h, int(first_arm), f.shape[0], f.ctypes.data, f.shape[1]))
        self.n_objects = f.shape[1]

    def set_mixer_weights(self, weights, first_arm=0):
        """Per-arm mixer weights (n_arms, 6); ``None`` returns to the batch-wide ``params.mix_w``."""
        if weights is None:
            self._chk(self.lib.vfik_set_mixer_weights(self.h, 0, 0, None))
            return
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1, _abi.MIX_CHANNELS)
        self._chk(self.lib.vfik_set_mixer_weights(self.h, int(first_arm), w.shape[0], w.ctypes.data))

    def set_ext_cmd(self, channel, cmd):
        if cmd is None:
            self._chk(self.lib.vfik_set_ext_cmd(self.h, int(channel), None))
            return
        a = np.ascontiguousarray(cmd, dtype=self.io_dtype)
        if a.shape != (self.batch, self.n):
            raise ValueError("cmd must be (batch, n)")
        self._chk(self.lib.vfik_set_ext_cmd(self.h, int(channel), a.ctypes.data))

    def reset_state(self):
        self._chk(self.lib.vfik_reset_state(self.h))

    @property
    def slots_in_use(self):
        return self.lib.vfik_slots_in_use(self.h)

    @property
    def field_path(self):
        """0 general, 1 straight-line (goal + decay repellers of one integer order), 2 straight-line with an aux block (one funnel and / or one hemisphere per arm)."""
        return self.lib.vfik_field_path(self.h)

    @property
    def uniform_repellers(self):
        """True when every decay repeller of the batch shares one safe distance and one force (the uniform repeller image is read)."""
        return bool(self.lib.vfik_uniform_repellers(self.h))

    @property
    def dh_pattern(self):
        """1 when the chain matches a DH pattern the lean kernels are specialised for (include/vfik.h: vfik_dh_pattern), else 0."""
        return int(self.lib.vfik_dh_pattern(self.h))

    @property
    def launch_epoch(self):
        """Moves with every call that can change what a launch bakes in: a captured hipGraph of steps is valid for the epoch it was
        captured under (include/vfik.h: vfik_launch_epoch)."""
        return int(self.lib.vfik_launch_epoch(self.h))

    @property
    def mixed_orders(self):
        """True when the batch's decay repellers have integer orders that differ (the order planes are read; ABI 5)."""
        return bool(self.lib.vfik_mixed_orders(self.h))

    @property
    def device_bytes(self):
        return self.lib.vfik_device_bytes(self.h)

    # -- one control cycle -----------------------------------------------------------------------
    def _shape(self, key):
        if key == "obj_dist":  # [B][n_objects][2]: one /dmonitor/distOut entry per object frame of set_objects
            if self.n_objects < 1:
                raise VfikError("obj_dist needs set_objects first")
            return (self.batch, self.n_objects, 2)
        d = _OUT_SHAPES[key]
        return (self.batch, self.n if d == "n" else d)

    def _host_inputs(self, io, keep, active=None, q_lo=None, q_hi=None):
        """The ABI-3 inputs of a host-array call: fresh-q gate and this cycle's per-arm joint limits."""
        if (q_lo is None) != (q_hi is None):
            raise ValueError("q_lo and q_hi come together")
        if active is not None:
            a = np.ascontiguousarray(np.asarray(active) != 0, dtype=np.int32)
            if a.shape != (self.batch,):
                raise ValueError("active must be (%d,), got %s" % (self.batch, a.shape))
            io.active = a.ctypes.data
            keep.append(a)
        for name, arr in (("q_lo", q_lo), ("q_hi", q_hi)):
            if arr is not None:
                a = np.ascontiguousarray(arr, dtype=self.io_dtype)
                if a.shape != (self.batch, self.n):
                    raise ValueError("%s must be (%d, %d), got %s" % (name, self.batch, self.n, a.shape))
                setattr(io, name, a.ctypes.data)
                keep.append(a)

    def step_host(self, q, null_control=None, want=("qdot_out",), q_ref=None, q_cmded=None, active=None, q_lo=None, q_hi=None,
                  into=None):
        """Host arrays in, host arrays out (copies + sync inside the library).  q_ref: /jpctrl/ref of the
        joint P controller (mixer channel 2); q_cmded: the LWR's commanded-position echo (bridge:199-203);
        active: fresh-q gate (B,) -- arms with 0 publish nothing and keep their state (vf:312-313); q_lo / q_hi:
        this cycle's joint limits per arm (nullspace:167).  `into`: a dict of arrays from an earlier call to write
        into (rows of gated arms then keep their previous content instead of zeros)."""
        q = np.ascontiguousarray(q, dtype=self.io_dtype)
        if q.shape != (self.batch, self.n):
            raise ValueError("q must be (%d, %d), got %s" % (self.batch, self.n, q.shape))
        io = IO()
        io.q = q.ctypes.data
        keep = [q]
        self._host_inputs(io, keep, active, q_lo, q_hi)
        for name, arr in (("q_ref", q_ref), ("q_cmded", q_cmded)):
            if arr is not None:
                a = np.ascontiguousarray(arr, dtype=self.io_dtype)
                if a.shape != (self.batch, self.n):
                    raise ValueError("%s must be (%d, %d), got %s" % (name, self.batch, self.n, a.shape))
                setattr(io, name, a.ctypes.data)
                keep.append(a)
        if null_control is not None:
            nc = np.ascontiguousarray(null_control, dtype=self.io_dtype)
            if nc.shape != (self.batch, _abi.NULL_CONTROLS):
                raise ValueError("null_control must be (batch, 4)")
            io.null_control = nc.ctypes.data
            keep.append(nc)
        out = {}
        for k in want:
            if into is not None and k in into:
                out[k] = into[k]
                self._check_host(k, out[k], (self.batch,) if k == "status" else self._shape(k), np.int32 if k == "status" else self.io_dtype)
            elif k == "status":
                out[k] = np.zeros(self.batch, dtype=np.int32)
            else:
                out[k] = np.zeros(self._shape(k), dtype=self.io_dtype)
            setattr(io, k, out[k].ctypes.data)
        self._chk(self.lib.vfik_step_host(self.h, C.byref(io)))
        return out

    # -- pipelined host path (vfik_submit_host / vfik_wait) ------------------------------------------
    def host_array(self, shape, dtype=None):
        """A pinned host array (hipHostMalloc) that submit_host can copy from / to asynchronously.
        Freed with the engine (close)."""
        dtype = np.dtype(self.io_dtype if dtype is None else dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = self.lib.vfik_host_alloc(self.h, max(nbytes, 1))
        if not p:
            raise VfikError("vfik_host_alloc: " + self.lib.vfik_last_error().decode())
        self._pinned.append(p)
        buf = (C.c_char * max(nbytes, 1)).from_address(p)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def submit_host(self, q, outs, null_control=None, q_ref=None, q_cmded=None, active=None, q_lo=None, q_hi=None):
        """Asynchronous vfik_step_host: ``q`` and the arrays of ``outs`` ({"qdot_out": array, ...}) must
        be C-contiguous arrays of the engine's dtype (``status`` and ``active``: int32) that stay untouched until
        :meth:`wait` -- use :meth:`host_array` (pinned memory) for overlap: copies from / to pageable memory make the
        call synchronous.  Returns the ticket."""
        io = IO()
        for name, arr, cols in (("q", q, self.n), ("null_control", null_control, _abi.NULL_CONTROLS), ("q_ref", q_ref, self.n),
                                ("q_cmded", q_cmded, self.n), ("q_lo", q_lo, self.n), ("q_hi", q_hi, self.n)):
            if arr is None:
                continue
            self._check_host(name, arr, (self.batch, cols), self.io_dtype)
            setattr(io, name, arr.ctypes.data)
        if (q_lo is None) != (q_hi is None):
            raise ValueError("q_lo and q_hi come together")
        if active is not None:
            self._check_host("active", active, (self.batch,), np.int32)
            io.active = active.ctypes.data
        for k, arr in outs.items():
            if k == "status":
                self._check_host(k, arr, (self.batch,), np.int32)
            else:
                self._check_host(k, arr, self._shape(k), self.io_dtype)
            setattr(io, k, arr.ctypes.data)
        t = C.c_long(-1)
        self._chk(self.lib.vfik_submit_host(self.h, C.byref(io), C.byref(t)))
        return int(t.value)

    @staticmethod
    def _check_host(name, arr, shape, dtype):
        if not isinstance(arr, np.ndarray) or arr.shape != tuple(shape) or arr.dtype != np.dtype(dtype) or not arr.flags.c_contiguous:
            raise ValueError("%s must be a C-contiguous %s array of shape %s" % (name, np.dtype(dtype).name, tuple(shape)))

    def wait(self, ticket):
        self._chk(self.lib.vfik_wait(self.h, int(ticket)))

    def rollout_host(self, q, n_cycles, dt, null_control=None, clamp=False, want=("qdot_out",), q_ref=None, active=None, q_lo=None,
                     q_hi=None, into=None):
        """n_cycles control cycles in one launch with q integrated on the device.
        Returns the outputs of the last cycle plus ``q`` = joint angles after it.  Arms gated off by ``active`` store
        nothing: their row of ``q`` comes back as it went in (a silent arm keeps its joint angles -- feeding the result into
        the next rollout is the normal closed-loop use), and their rows of the other outputs keep what ``into`` (a dict of
        arrays from an earlier call, as in :meth:`step_host`) held, zeros without it."""
        q = np.ascontiguousarray(q, dtype=self.io_dtype)
        if q.shape != (self.batch, self.n):
            raise ValueError("q must be (%d, %d), got %s" % (self.batch, self.n, q.shape))
        io = IO()
        io.q = q.ctypes.data
        keep = [q]
        self._host_inputs(io, keep, active, q_lo, q_hi)
        if q_ref is not None:
            a = np.ascontiguousarray(q_ref, dtype=self.io_dtype)
            if a.shape != (self.batch, self.n):
                raise ValueError("q_ref must be (%d, %d), got %s" % (self.batch, self.n, a.shape))
            io.q_ref = a.ctypes.data
            keep.append(a)
        if null_control is not None:
            nc = np.ascontiguousarray(null_control, dtype=self.io_dtype)
            io.null_control = nc.ctypes.data
            keep.append(nc)
        out = {}
        for k in want:
            if into is not None and k in into:
                out[k] = into[k]
                self._check_host(k, out[k], (self.batch,) if k == "status" else self._shape(k), np.int32 if k == "status" else self.io_dtype)
            else:
                out[k] = np.zeros(self.batch, dtype=np.int32) if k == "status" else np.zeros(self._shape(k), dtype=self.io_dtype)
            setattr(io, k, out[k].ctypes.data)
        out["q"] = q.copy()  # gated arms keep their angles (the kernel stores nothing for them)
        self._chk(self.lib.vfik_rollout_host(self.h, C.byref(io), int(n_cycles), float(dt), 1 if clamp else 0, out["q"].ctypes.data))
        return out

    def rollout(self, io, n_cycles, dt, q_out=None, clamp=False):
        """Asynchronous device-pointer form of :meth:`rollout_host`."""
        self._chk(self.lib.vfik_rollout(self.h, C.byref(io), int(n_cycles), float(dt), 1 if clamp else 0, C.c_void_p(_ptr(q_out))))

    def make_io(self, q, null_control=None, q_ref=None, q_cmded=None, active=None, q_lo=None, q_hi=None, **outs):
        """IO block from device pointers (torch tensors on this device, or raw addresses).  active: int32 [B]."""
        io = IO()
        io.q = _ptr(q)
        io.null_control = _ptr(null_control)
        io.q_ref = _ptr(q_ref)
        io.q_cmded = _ptr(q_cmded)
        io.active = _ptr(active)
        io.q_lo = _ptr(q_lo)
        io.q_hi = _ptr(q_hi)
        for k, v in outs.items():
            setattr(io, k, _ptr(v))
        return io

    def step(self, io):
        """Asynchronous launch on the handle's stream; ``io`` from :meth:`make_io`."""
        self._chk(self.lib.vfik_step(self.h, C.byref(io)))

    def stepper(self, io):
        """The hot enqueue as a zero-argument callable: ``byref(io)``, the handle and the prototype are bound ONCE, the
        return code is checked inline.  ``io`` (and the buffers it names) must outlive the callable; changing a member of
        ``io`` afterwards is seen by the next call (the library reads the struct at every launch).  What bench.py's timed loop and any
        closed-loop driver at rate should call: at a 5-us launch period the Python-side cost of ``step`` is a third of the budget."""
        fn, h, ref, err = self.lib.vfik_step, self.h, C.byref(io), self._chk

        def step():
            rc = fn(h, ref)
            if rc:
                err(rc)
        step.io = io  # keeps the struct alive
        return step

    def sync(self):
        self._chk(self.lib.vfik_sync(self.h))

    def time_steps(self, io, warmup, steps):
        ms = C.c_float(0.0)
        self._chk(self.lib.vfik_time_steps(self.h, C.byref(io), int(warmup), int(steps), C.byref(ms)))
        return float(ms.value)

    # -- device memory without torch -------------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = self.lib.vfik_dev_alloc(self.h, int(nbytes))
        if not p:
            raise VfikError("vfik_dev_alloc: " + self.lib.vfik_last_error().decode())
        return p

    def dev_free(self, p):
        self._chk(self.lib.vfik_dev_free(self.h, C.c_void_p(p)))

    def h2d(self, dst, arr):
        a = np.ascontiguousarray(arr)
        self._chk(self.lib.vfik_memcpy_h2d(self.h, C.c_void_p(dst), a.ctypes.data, a.nbytes))

    def d2h(self, arr, src):
        self._chk(self.lib.vfik_memcpy_d2h(self.h, arr.ctypes.data, C.c_void_p(src), arr.nbytes))

    def set_max_vel(self, values, first_arm=0):
        """Per-arm limiter speed (the /bridge/max_vel value each bridge keeps, bridge:612-623)."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        self._chk(self.lib.vfik_set_max_vel(self.h, int(first_arm), len(v), v.ctypes.data))

    def set_arm_weights(self, wy=None, wq=None, first_arm=0):
        """Per-arm IK weights (vf:295-309): wy (n_arms, 6) and/or wq (n_arms, n) starting at first_arm."""
        arrs = []
        for w, cols in ((wy, 6), (wq, self.n)):
            if w is None:
                arrs.append(None)
                continue
            a = np.ascontiguousarray(w, dtype=np.float64)
            if a.ndim != 2 or a.shape[1] != cols:
                raise ValueError("weights must be (n_arms, %d), got %s" % (cols, a.shape))
            arrs.append(a)
        counts = {a.shape[0] for a in arrs if a is not None}
        if len(counts) != 1:
            raise ValueError("wy and wq must cover the same arms")
        self._chk(self.lib.vfik_set_arm_weights(self.h, int(first_arm), counts.pop(),
                                                None if arrs[0] is None else arrs[0].ctypes.data,
                                                None if arrs[1] is None else arrs[1].ctypes.data))

    def set_small_batch_kernel(self, max_batch):
        """Lean launches of batches up to max_batch arms take the eight-lanes-per-arm kernel (0 = never)."""
        self._chk(self.lib.vfik_set_small_batch_kernel(self.h, int(max_batch)))

    @property
    def small_batch_launches(self):
        return int(self.lib.vfik_small_batch_launches(self.h))

    def probe_field(self, pose_dev, v6_dev):
        """The field of every arm at a given pose (vf:469-503): device pose[B][16] -> v6[B][6]."""
        self._chk(self.lib.vfik_probe_field(self.h, C.c_void_p(_ptr(pose_dev)), C.c_void_p(_ptr(v6_dev))))

    def object_distances(self, pose_dev, frames_dev, max_objects, out_dev):
        """Distance monitor (monitor_distance:148-167) on device arrays: out[B][max_objects][2]."""
        self._chk(self.lib.vfik_object_distances(self.h, C.c_void_p(_ptr(pose_dev)), C.c_void_p(_ptr(frames_dev)), int(max_objects),
                                                 C.c_void_p(_ptr(out_dev))))

    def track_error(self, pose_dev, v6_dev, out_dev, active_dev=None):
        """One step of the tracking-error estimator (vf:349-428) on device arrays; active_dev: int32 [B] gate
        (arms with 0 append no frame, vf:312-313), or None."""
        self._chk(self.lib.vfik_track_error(self.h, C.c_void_p(_ptr(pose_dev)), C.c_void_p(_ptr(v6_dev)), C.c_void_p(_ptr(out_dev)),
                                            C.c_void_p(_ptr(active_dev))))

    def track_reset(self):
        self._chk(self.lib.vfik_track_reset(self.h))

    def mix(self, cmds_dev, weights, out_dev):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self._chk(self.lib.vfik_mix(self.h, C.c_void_p(_ptr(cmds_dev)), w.ctypes.data, len(w), C.c_void_p(_ptr(out_dev))))
