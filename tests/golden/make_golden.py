#!/opt/conda/bin/python3.9
"""Generate golden vectors from the two fragments of the reference that are importable.

Run ONLY in the build container (the reference tree does not exist on the GPU box):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

Writes tests/golden/mixer_golden.npz and tests/golden/nullspace_golden.npz.  Only
arrays (inputs and the reference's outputs) are stored; no reference source travels.

What is executed from the reference:
  * src/command_mixer.py   -> CommandMixer.__init__/read   (command_mixer.py:32-82)
  * scripts/nullspace      -> restrict, nullspace, move_in_nullspace, check_limits
                              (nullspace:75-131), module globals sig/lastvec (:91-92)

Both files import modules that do not exist here (yarp, arcospyu.*).  None of the
functions above calls into them, so this harness registers empty placeholder
modules under those names for the duration of the import; the reference files are
not modified.  scripts/nullspace needs numpy<2 (`from numpy import mat`), hence
python3.9 + numpy 1.26.4.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# placeholder modules so that the module-level imports succeed
# ----------------------------------------------------------------------------
class _Cfg:
    nJoints = 7
    robotarm_portbasename = "/lwr/right"


class _Opts:
    namespace = "/0"


def _install_placeholders(n_joints):
    _Cfg.nJoints = n_joints
    yarp = types.ModuleType("yarp")

    class _Net:
        @staticmethod
        def init():
            pass

    yarp.Network = _Net
    arcospyu = types.ModuleType("arcospyu")
    cp = types.ModuleType("arcospyu.config_parser")

    class ConfigFileParser:
        def __init__(self, *a, **k):
            pass

        def get_all(self):
            return _Opts(), [], _Cfg()

    cp.ConfigFileParser = ConfigFileParser
    rt = types.ModuleType("arcospyu.robot_tools")

    class Lafik:
        def __init__(self, *a, **k):
            pass

    rt.Lafik = Lafik
    yt = types.ModuleType("arcospyu.yarp_tools")
    ych = types.ModuleType("arcospyu.yarp_tools.yarp_comm_helpers")

    class ArcosYarp:
        def __init__(self, *a, **k):
            pass

    ych.ArcosYarp = ArcosYarp
    for name, mod in [("yarp", yarp), ("arcospyu", arcospyu), ("arcospyu.config_parser", cp),
                      ("arcospyu.robot_tools", rt), ("arcospyu.yarp_tools", yt),
                      ("arcospyu.yarp_tools.yarp_comm_helpers", ych)]:
        sys.modules[name] = mod


def _load(path, name):
    loader = importlib.machinery.SourceFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------
# duck-typed port / bottle, exactly the surface CommandMixer.read touches
# ----------------------------------------------------------------------------
class _Val:
    def __init__(self, v):
        self.v = v

    def asDouble(self):
        return float(self.v)


class _Bottle:
    def __init__(self, vals):
        self.vals = list(vals)

    def size(self):
        return len(self.vals)

    def get(self, i):
        return _Val(self.vals[i])

    def __bool__(self):  # a yarp Bottle pointer is truthy when not NULL
        return True


class _Port:
    def __init__(self):
        self.pending = None

    def read(self, blocking=False):
        b, self.pending = self.pending, None
        return b


class _Clock:
    """Stands in for the `time` module inside command_mixer (only .time() is used)."""

    def __init__(self):
        self.now = 1000.0

    def time(self):
        return self.now


def make_mixer_golden():
    _install_placeholders(7)
    cm = _load(os.path.join(REF, "src", "command_mixer.py"), "ref_command_mixer")
    clock = _Clock()
    cm.time = clock  # module attribute used as time.time() (command_mixer.py:44,60,64)
    rng = np.random.default_rng(20261004)

    scenarios = {}
    for tag, (K, n) in {"k6n7": (6, 7), "k6n14": (6, 14), "k2n6": (2, 6)}.items():
        ports = [_Port() for _ in range(K)]
        wport = _Port()
        init_w = [1.0, 1.0] + [0.0] * (K - 2)
        guard = 2.0
        clock.now = 1000.0
        mixer = cm.CommandMixer(ports, wport, n, guard, list(init_w))
        T = 40
        nmax = n + 2
        clk = np.zeros(T)
        wb = np.full((T, K + 2), np.nan)
        wlen = np.full(T, -1, dtype=np.int64)
        cmd = np.full((T, K, nmax), np.nan)
        clen = np.full((T, K), -1, dtype=np.int64)
        expect = np.zeros((T, n))
        wafter = np.zeros((T, K))
        for t in range(T):
            # irregular clock: mostly 10 ms cycles, two long pauses that trip the watchdog
            clock.now += 0.01 if t not in (17, 29) else 2.5
            clk[t] = clock.now
            if t in (3, 9, 21, 33):
                ln = [K, 4, K + 2, 1][(3, 9, 21, 33).index(t)]
                vals = rng.uniform(-1.5, 1.5, size=ln)
                wb[t, :ln] = vals
                wlen[t] = ln
                wport.pending = _Bottle(vals)
            for k in range(K):
                u = rng.uniform()
                if t < 2 and k > 1:
                    continue  # nothing heard yet on the extra channels
                if u < 0.55:
                    ln = n
                elif u < 0.65:
                    ln = n - 1  # wrong size -> ignored (command_mixer.py:67-69)
                elif u < 0.70:
                    ln = n + 2
                else:
                    continue  # silent this cycle
                vals = rng.normal(size=ln)
                if t == 25 and k == 0 and ln == n:
                    vals[2] = np.nan  # NaN is reported but passed through (:71-75)
                cmd[t, k, :ln] = vals
                clen[t, k] = ln
                ports[k].pending = _Bottle(vals)
            expect[t] = mixer.read()
            wafter[t] = mixer.weights
        scenarios[tag] = dict(K=K, n=n, guard=guard, init_w=np.array(init_w), t0=1000.0, clock=clk,
                              wbottle=wb, wlen=wlen, cmd=cmd, cmdlen=clen, expect=expect,
                              weights_after=wafter)

    # constructor with wrong number of initial weights -> all zeros (command_mixer.py:37-39)
    ports = [_Port() for _ in range(3)]
    mixer = cm.CommandMixer(ports, None, 4, 2.0, [1.0, 1.0])
    for p in ports:
        p.pending = _Bottle([1.0, 2.0, 3.0, 4.0])
    bad_w = np.array(mixer.read())

    flat = {}
    for tag, sc in scenarios.items():
        for key, val in sc.items():
            flat[tag + "__" + key] = np.asarray(val)
    flat["badw__expect"] = bad_w
    np.savez(os.path.join(OUT, "mixer_golden.npz"), **flat)
    print("mixer_golden.npz written:", sorted(scenarios))


def _rand_jac(rng, n):
    """A plausible 6 x n geometric Jacobian: random unit axes, random lever arms."""
    z = rng.normal(size=(3, n))
    z /= np.linalg.norm(z, axis=0)
    r = rng.uniform(-0.8, 0.8, size=(3, n))
    return np.vstack([np.cross(z.T, r.T).T, z])


def make_nullspace_golden():
    out = {}
    for n in (7, 14, 6):
        _install_placeholders(n)
        ns = _load(os.path.join(REF, "scripts", "nullspace"), "ref_nullspace_%d" % n)
        from numpy import mat, eye
        P = mat(eye(6))
        rng = np.random.default_rng(7000 + n)

        # (1) single-shot restrict on independent Jacobians (state untouched)
        S = 12
        Js = np.stack([_rand_jac(rng, n) for _ in range(S)])
        if n == 7:
            # near-singular: two almost parallel columns
            Js[-1][:, 6] = Js[-1][:, 4] * (1.0 + 1e-7) + 1e-9 * rng.normal(size=6)
            # exactly rank deficient: duplicated column
            Js[-2][:, 5] = Js[-2][:, 3]
        Bs = np.stack([np.asarray(ns.restrict(P, mat(J))) for J in Js])
        out["n%d__restrict_J" % n] = Js
        out["n%d__restrict_B" % n] = Bs

        # (2) a smooth trajectory of Jacobians, stepping the stateful basis (sig / lastvec)
        T = 60
        J0, J1 = _rand_jac(rng, n), _rand_jac(rng, n)
        traj = np.stack([J0 * np.cos(0.05 * t) + J1 * np.sin(0.05 * t) for t in range(T)])
        controls = rng.uniform(-1, 1, size=(T, 4))
        rows = np.full((T, n, n), np.nan)
        rank = np.zeros(T, dtype=np.int64)
        qd = np.zeros((T, n))
        raw_u = np.zeros((T, n, n))
        for t in range(T):
            basis = np.asarray(ns.nullspace(P, mat(traj[t])))
            rank[t] = basis.shape[0]
            rows[t, :basis.shape[0]] = basis
            # same call again through move_in_nullspace (advances the state once more with the
            # same J, which is idempotent for the sign logic)
            qd[t] = ns.move_in_nullspace(P, mat(traj[t]), list(controls[t]))
            raw_u[t] = np.linalg.svd(np.asarray(ns.restrict(P, mat(traj[t]))).T)[0]
        out["n%d__traj_J" % n] = traj
        out["n%d__traj_control" % n] = controls
        out["n%d__traj_basis" % n] = rows
        out["n%d__traj_rank" % n] = rank
        out["n%d__traj_qdot" % n] = qd
        out["n%d__traj_raw_u" % n] = raw_u

        # (3) check_limits (nullspace:120-131): pass, trip low, trip high, exactly on the bound
        lim = np.stack([-np.linspace(1.0, 2.9, n), np.linspace(1.1, 3.0, n)], axis=1)
        C = 16
        q = rng.uniform(-0.9, 0.9, size=(C, n)) * np.abs(lim[:, 0])
        qdot = rng.normal(scale=1.5, size=(C, n))
        q[0], qdot[0] = 0.0, 0.0
        q[1, 2], qdot[1, 2] = lim[2, 1] - 0.3, 1.0 + 1e-12  # just past the upper bound
        q[2, 2], qdot[2, 2] = lim[2, 1] - 0.3, 1.0 - 1e-9   # just inside
        res = np.zeros((C, n))
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            for c in range(C):
                res[c] = ns.check_limits(list(q[c]), list(qdot[c]), [list(l) for l in lim])
        out["n%d__lim_limits" % n] = lim
        out["n%d__lim_q" % n] = q
        out["n%d__lim_qdot" % n] = qdot
        out["n%d__lim_out" % n] = res
    np.savez(os.path.join(OUT, "nullspace_golden.npz"), **out)
    print("nullspace_golden.npz written; numpy", np.__version__)


if __name__ == "__main__":
    make_mixer_golden()
    make_nullspace_golden()
