#!/opt/conda/bin/python3.9
"""Generate golden vectors from the two fragments of the reference that are importable.

Run ONLY in the build container (the reference tree does not exist on the GPU box):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

Writes tests/golden/mixer_golden.npz, tests/golden/nullspace_golden.npz and tests/golden/jpctrl_golden.npz.  Only
arrays (inputs and the reference's outputs) are stored; no reference source travels.

What is executed from the reference -- ONLY these definitions, nothing else of the files:
  * src/command_mixer.py   -> class CommandMixer (__init__/read, command_mixer.py:32-82)
  * scripts/nullspace      -> restrict, nullspace, move_in_nullspace, check_limits, matrixrank, sign
                              (nullspace:67-131) and the module globals sig / lastvec (:91-92)
  * scripts/joint_p_controller -> check_limits (joint_p_controller:79-89; round 3: the file parses as Python 3, its one pure
                              function is the clamp of the joint reference against `config.updateJntLimits(cur_pos)`)

The reference tree is untrusted content: the files are read as TEXT, parsed with `ast`, and only the
whitelisted top-level definitions (plus the files' own `import numpy` / `from numpy ... import` / `import time`
lines) are compiled and executed in a fresh namespace.  Their module-level side effects (signal handlers,
prctl process renaming, yarp initialisation, port creation, the control loops) never run, no module of the
reference is imported, and nothing is written next to it (`sys.dont_write_bytecode`).  scripts/nullspace needs
numpy<2 (`from numpy import mat`), hence python3.9 + numpy 1.26.4.
"""
import ast
import os
import sys

sys.dont_write_bytecode = True  # never leave __pycache__ inside the read-only reference tree

import numpy as np  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# extraction: whitelisted definitions of a reference file, executed in a fresh namespace
# ----------------------------------------------------------------------------
_ALLOWED_IMPORTS = {"numpy", "numpy.linalg", "numpy.linalg.linalg", "time", "math"}


def _extract(path, names, preset=None):
    """Namespace holding the top-level functions / classes / assignments called `names` of the file at `path`
    (read as text), plus the file's own imports of numpy / time / math.  Nothing else of the file is executed."""
    with open(path, "r") as f:
        tree = ast.parse(f.read(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.Import):
            if all(a.name in _ALLOWED_IMPORTS for a in node.names):
                keep.append(node)
        elif isinstance(node, ast.ImportFrom):
            if node.module in _ALLOWED_IMPORTS and node.level == 0:
                keep.append(node)
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            if node.name in names:
                keep.append(node)
        elif isinstance(node, ast.Assign):
            if all(isinstance(t, ast.Name) and t.id in names for t in node.targets):
                keep.append(node)
    found = {n.name for n in keep if isinstance(n, (ast.FunctionDef, ast.ClassDef))} | \
            {t.id for n in keep if isinstance(n, ast.Assign) for t in n.targets}
    missing = set(names) - found
    if missing:
        raise RuntimeError("%s: definitions not found: %s" % (path, sorted(missing)))
    ns = dict(preset or {})
    ns["__name__"] = "extracted_" + os.path.basename(path)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


class _NS:
    """attribute access to an extracted namespace (so that module globals like sig / lastvec stay shared)"""

    def __init__(self, ns):
        self.__dict__ = ns


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
    cm = _NS(_extract(os.path.join(REF, "src", "command_mixer.py"), ["CommandMixer"]))
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
        # nJoints is `config.nJoints` in the file (nullspace:61); the module globals sig / lastvec are built from it
        ns = _NS(_extract(os.path.join(REF, "scripts", "nullspace"),
                          ["restrict", "nullspace", "move_in_nullspace", "check_limits", "matrixrank", "sign", "sig", "lastvec"],
                          preset={"nJoints": n}))
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
    # (4) the raw sign LAPACK's SVD gives the unique null vector when its FIRST component is (nearly) zero: columns
    # 1..6 are made dependent up to eps * (a random direction), so u_0 = O(eps) and the rule "first NON-NEGLIGIBLE
    # component negative" is what decides.  Fresh state (sig = 1, lastvec = 0) for every case: nullspace() then
    # returns the raw vector.
    eps_list = [0.0, 1e-14, 1e-12, 1e-10, 1e-8, 1e-6, 1e-3]
    rng = np.random.default_rng(7777)
    Jz, Uz = [], []
    for eps in eps_list:
        for rep_ in range(4):
            ns = _NS(_extract(os.path.join(REF, "scripts", "nullspace"),
                              ["restrict", "nullspace", "move_in_nullspace", "check_limits", "matrixrank", "sign", "sig", "lastvec"],
                              preset={"nJoints": 7}))
            from numpy import mat, eye
            J = _rand_jac(rng, 7)
            coef = rng.normal(size=5)
            J[:, 6] = J[:, 1:6] @ coef + eps * J[:, 0]   # J u = 0 with u = (eps, coef, -1) / |.|
            basis = np.asarray(ns.nullspace(mat(eye(6)), mat(J)))
            assert basis.shape[0] == 1, (eps, basis.shape)
            Jz.append(J)
            Uz.append(basis[0])
    out["n7__zero_first_J"] = np.stack(Jz)
    out["n7__zero_first_u"] = np.stack(Uz)
    out["n7__zero_first_eps"] = np.repeat(eps_list, 4)
    np.savez(os.path.join(OUT, "nullspace_golden.npz"), **out)
    print("nullspace_golden.npz written; numpy", np.__version__)


def make_jpctrl_golden():
    """check_limits(ref, cur_pos) of scripts/joint_p_controller (:79-89): the reference clamped against the limits that
    `config.updateJntLimits(cur_pos)` returns for THIS position (robots whose limits move with the pose) -- here limits that
    shrink with |cur_pos| so that the position argument matters.  Stored: ref, cur_pos, the limits the stand-in config returned, and
    the reference's output (a list -> array)."""
    rng = np.random.default_rng(21)
    seen = []

    class _Config:
        def updateJntLimits(self, cur_pos):
            cur = np.asarray(cur_pos, dtype=float)
            lim = [[-2.0 + 0.1 * abs(c), 2.0 - 0.2 * abs(c)] for c in cur]
            seen.append(lim)
            return lim

    import builtins
    quiet = dict(vars(builtins))
    quiet["print"] = lambda *a, **k: None   # the function reports every clamp on stdout
    ns = _extract(os.path.join(REF, "scripts", "joint_p_controller"), ["check_limits"], preset={"config": _Config(), "__builtins__": quiet})
    refs, curs, outs = [], [], []
    for n in (6, 7, 14):
        for _ in range(40):
            ref = rng.uniform(-3.0, 3.0, n)
            ref[rng.integers(0, n)] = rng.choice([-2.0, 2.0, 1.8, -1.9])   # values on / near a limit: strict comparisons
            cur = rng.uniform(-1.5, 1.5, n)
            out = ns["check_limits"](ref.tolist(), cur.tolist())
            refs.append(np.pad(ref, (0, 14 - n), constant_values=np.nan))
            curs.append(np.pad(cur, (0, 14 - n), constant_values=np.nan))
            outs.append(np.pad(np.asarray(out, dtype=float), (0, 14 - n), constant_values=np.nan))
    lims = np.full((len(seen), 14, 2), np.nan)
    for k, lim in enumerate(seen):
        lims[k, :len(lim)] = lim
    np.savez(os.path.join(OUT, "jpctrl_golden.npz"), ref=np.stack(refs), cur_pos=np.stack(curs), limits=lims, ref_out=np.stack(outs))
    print("jpctrl_golden.npz written: %d cases" % len(refs))


if __name__ == "__main__":
    make_mixer_golden()
    make_nullspace_golden()
    make_jpctrl_golden()
