"""The oracle against the golden vectors captured from the reference (tests/golden/make_golden.py).

mixer_golden.npz      <- reference src/command_mixer.py CommandMixer.read   (bit-exact bar)
nullspace_golden.npz  <- reference scripts/nullspace restrict / nullspace / move_in_nullspace /
                         check_limits                                         (1e-9 bar, SVD-based)
"""
import os

import numpy as np
import pytest

from oracle import vfik_numpy as vn


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


class _Port:
    def __init__(self):
        self.pending = None

    def read(self, blocking=False):
        b, self.pending = self.pending, None
        return b


def _replay_mixer(g, tag, mixer_cls, clock_box):
    K, n = int(g[tag + "__K"]), int(g[tag + "__n"])
    ports = [_Port() for _ in range(K)]
    wport = _Port()
    clock_box[0] = float(g[tag + "__t0"])
    mixer = mixer_cls(ports, wport, n, float(g[tag + "__guard"]), list(g[tag + "__init_w"]))
    outs, ws = [], []
    for t in range(len(g[tag + "__clock"])):
        clock_box[0] = float(g[tag + "__clock"][t])
        wl = int(g[tag + "__wlen"][t])
        if wl >= 0:
            wport.pending = _Bottle(g[tag + "__wbottle"][t, :wl])
        for k in range(K):
            cl = int(g[tag + "__cmdlen"][t, k])
            if cl >= 0:
                ports[k].pending = _Bottle(g[tag + "__cmd"][t, k, :cl])
        outs.append(mixer.read())
        ws.append(list(mixer.weights))
    return np.array(outs), np.array(ws)


@pytest.mark.parametrize("tag", ["k6n7", "k6n14", "k2n6"])
def test_numpy_mixer_bit_exact(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "mixer_golden.npz"))
    box = [0.0]

    def mk(ports, wport, n, guard, w):
        return vn.CommandMixer(ports, wport, n, guard, w, clock=lambda: box[0])

    out, ws = _replay_mixer(g, tag, mk, box)
    exp = g[tag + "__expect"]
    assert np.array_equal(np.isnan(out), np.isnan(exp))
    assert np.array_equal(np.nan_to_num(out, nan=123.0).view(np.uint64), np.nan_to_num(exp, nan=123.0).view(np.uint64))
    assert np.array_equal(ws, g[tag + "__weights_after"])


def test_mixer_wrong_initial_weights(golden_dir):
    g = np.load(os.path.join(golden_dir, "mixer_golden.npz"))
    ports = [_Port() for _ in range(3)]
    m = vn.CommandMixer(ports, None, 4, 2.0, [1.0, 1.0])
    for p in ports:
        p.pending = _Bottle([1.0, 2.0, 3.0, 4.0])
    assert np.array_equal(np.array(m.read()), g["badw__expect"])


@pytest.mark.parametrize("tag", ["k6n7", "k6n14", "k2n6"])
def test_c_mix_bit_exact(golden_dir, oracle_c, tag):
    """vfo_mix on the per-cycle (last_command, weights) state reproduces every golden sum."""
    g = np.load(os.path.join(golden_dir, "mixer_golden.npz"))
    box = [0.0]
    captured = []

    class Spy(vn.CommandMixer):
        def read(self):
            r = super().read()
            captured.append((np.array(self.last_command), np.array(self.weights)))
            return r

    _replay_mixer(g, tag, lambda *a: Spy(*a, clock=lambda: box[0]), box)
    exp = g[tag + "__expect"]
    for t, (cmd, w) in enumerate(captured):
        got = oracle_c.mix(cmd, w)
        a, b = np.nan_to_num(got, nan=7.0), np.nan_to_num(exp[t], nan=7.0)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), t


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [6, 7, 14])
def test_restrict(golden_dir, oracle_c, n):
    g = np.load(os.path.join(golden_dir, "nullspace_golden.npz"))
    for J, Bref in zip(g["n%d__restrict_J" % n], g["n%d__restrict_B" % n]):
        assert np.abs(vn.restrict(np.eye(6), J) - Bref).max() < 1e-9
        # the last n=7 case has two columns parallel to 1e-7: pinv amplifies rounding there
        tol = 1e-9 if np.linalg.cond(J) < 1e6 else 1e-6
        assert np.abs(oracle_c.restrict(J) - Bref).max() < tol


@pytest.mark.parametrize("n", [6, 7, 14])
def test_nullspace_trajectory(golden_dir, oracle_c, n):
    """Stateful basis + move_in_nullspace along a smooth Jacobian path (sign continuity)."""
    g = np.load(os.path.join(golden_dir, "nullspace_golden.npz"))
    traj, ctrl = g["n%d__traj_J" % n], g["n%d__traj_control" % n]
    rank, qd_ref, basis_ref = g["n%d__traj_rank" % n], g["n%d__traj_qdot" % n], g["n%d__traj_basis" % n]
    ns_np = vn.Nullspace(n)
    ns_c = oracle_c.NullspaceC(n)
    for t in range(len(traj)):
        b_np = ns_np.nullspace(np.eye(6), traj[t])
        b_c = ns_c.basis(traj[t])
        assert b_np.shape[0] == rank[t] == b_c.shape[0]
        qd_np, _ = ns_np.move_in_nullspace(np.eye(6), traj[t], list(ctrl[t]))
        qd_c, _ = ns_c.move(traj[t], ctrl[t])
        if rank[t] == 1:
            # unique up to the tracked sign: must match the reference vector itself
            assert np.abs(b_np[0] - basis_ref[t, 0]).max() < 1e-9
            assert np.abs(b_c[0] - basis_ref[t, 0]).max() < 1e-9
            assert np.abs(np.array(qd_np) - qd_ref[t]).max() < 1e-9
            assert np.abs(qd_c - qd_ref[t]).max() < 1e-9
        elif rank[t] > 1:
            # nullity > 1: the basis inside the eigenspace is LAPACK's choice; the spanned subspace
            # is what can be compared (VFIK_ST_NULL_AMBIGUOUS)
            Pref = basis_ref[t, :rank[t]].T @ basis_ref[t, :rank[t]]
            assert np.abs(b_np.T @ b_np - Pref).max() < 1e-9
            assert np.abs(b_c.T @ b_c - Pref).max() < 1e-9
        else:
            assert np.all(np.array(qd_np) == 0) and np.all(qd_c == 0)


def test_raw_sign_rule(golden_dir):
    """The restated LAPACK sign convention (first component negative) holds on every golden step."""
    g = np.load(os.path.join(golden_dir, "nullspace_golden.npz"))
    assert np.all(g["n7__traj_raw_u"][:, 0, 0] < 0)


@pytest.mark.parametrize("n", [6, 7, 14])
def test_check_limits(golden_dir, oracle_c, n):
    g = np.load(os.path.join(golden_dir, "nullspace_golden.npz"))
    lim = g["n%d__lim_limits" % n]
    tripped = 0
    for q, qd, ref in zip(g["n%d__lim_q" % n], g["n%d__lim_qdot" % n], g["n%d__lim_out" % n]):
        got, t1 = vn.check_limits(list(q), list(qd), [list(l) for l in lim])
        assert np.array_equal(np.array(got), ref)
        got_c, t2 = oracle_c.check_limits(q, qd, lim[:, 0], lim[:, 1])
        assert np.array_equal(got_c, ref)
        assert t1 == t2
        tripped += t1
    assert 0 < tripped < len(g["n%d__lim_q" % n])
