"""Decay repellers whose INTEGER orders differ -- between obstacles (old/README.old:75 documents `ObstacleP ... 0.05 20` while the
feeder's own near-goal repeller has order 5, object_feeder:302) or between arms -- stay on the straight-line field path since ABI 5
(cycle_kernel's MIXO variants read one order byte per compact-image slot).  Every case against the oracle, which evaluates each
primitive with its own order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status")


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, engine, robots, synth
    return dict(oc=oracle_c, abi=_abi, engine=engine, robots=robots, synth=synth)


def _round(w, dt):
    w["q"] = w["q"].astype(dt).astype(np.float64)
    w["fields"]["p"] = w["fields"]["p"].astype(dt).astype(np.float64)
    w["fields"]["force"] = w["fields"]["force"].astype(dt).astype(np.float64)
    return w


def _check(env, chain, params, w, dt, want, max_slots, null_control=None, expect_path=1, active=None):
    eng = env["engine"].Engine(chain, w["q"].shape[0], io_dtype=dt, max_slots=max_slots, device=0, params=params)
    eng.set_small_batch_kernel(0)
    eng.set_fields(w["fields"], w["nfields"])
    assert eng.field_path == expect_path and eng.mixed_orders
    got = eng.step_host(w["q"], null_control=null_control, want=want, active=active)
    ref = env["oc"].cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], null_control=null_control, want=want)
    tol = 1e-9 if dt == np.float64 else 1e-6
    rows = slice(None) if active is None else active.astype(bool)
    for k in want:
        if k == "status":
            assert np.array_equal(got[k][rows], ref[k][rows])
        else:
            err = np.abs(got[k][rows].astype(np.float64) - ref[k][rows]).max()
            assert err < tol, (k, err)
    eng.close()
    return got, ref


@pytest.mark.parametrize("robot,dt,flags,nrep", [("lwr", np.float32, 0, 8), ("lwr", np.float64, 0, 8), ("lwr", np.float32, 3, 11),
                                                  ("lwr", np.float64, 7, 5), ("lwr_dual14", np.float32, 7, 16),
                                                  ("lwr_dual14", np.float64, 0, 19), ("powercube6", np.float32, 0, 3)])
def test_orders_that_differ_by_obstacle(env, robot, dt, flags, nrep):
    """Slot-wise the same for every arm (the README scene's shape): each wave agrees and takes the scalar-controlled powers.
    Orders 0 ... 20, ragged counts, more than one chunk of slots; lean launch (qdot_out + status) and every per-cycle row."""
    chain = env["robots"].by_name(robot)
    B = 64 * 9 + 17
    w = _round(env["synth"].make_workload(chain, B, nrep, seed=41, io_dtype=dt), dt)
    orders = np.array([5, 20, 2, 7, 1, 0, 13, 3, 20, 5, 4, 6, 2, 20, 9, 1, 5, 8, 3])[:nrep]
    w["fields"]["p"][:, 1:1 + nrep, 5] = orders
    rng = np.random.default_rng(3)
    w["nfields"][:] = 1 + rng.integers(0, nrep + 1, B)     # ragged: a shorter list is a prefix of the same order table
    w["nfields"][:64] = 1 + nrep
    params = env["abi"].default_params(flags=flags)
    ctrl = rng.uniform(-1, 1, (B, 4)).astype(dt).astype(np.float64) if (flags & 1) and chain.n <= 7 else None
    got, ref = _check(env, chain, params, w, dt, ("qdot_out", "status"), max(nrep, 1))
    assert np.abs(ref["qdot_out"]).max() > 0.05
    _check(env, chain, params, w, dt, ALL, max(nrep, 1), null_control=ctrl)


@pytest.mark.parametrize("robot,dt,flags", [("lwr", np.float32, 0), ("lwr", np.float64, 3), ("lwr_dual14", np.float32, 7), ("lwr_dual14", np.float64, 7)])
def test_orders_that_differ_by_arm(env, robot, dt, flags):
    """One odd arm in an otherwise order-5 batch (its wave takes the per-lane selects, the others the all-fives shortcut), then
    every arm with its own random orders (every wave on the per-lane path)."""
    chain = env["robots"].by_name(robot)
    nrep = 8 if chain.n <= 7 else 16
    B = 64 * 6 + 5
    w = _round(env["synth"].make_workload(chain, B, nrep, seed=43, io_dtype=dt), dt)
    w["fields"]["p"][130, 1 + 3, 5] = 2.0
    params = env["abi"].default_params(flags=flags)
    _check(env, chain, params, w, dt, ("qdot_out", "status"), nrep)
    rng = np.random.default_rng(5)
    w["fields"]["p"][:, 1:1 + nrep, 5] = rng.integers(0, 24, (B, nrep))
    w["fields"]["p"][7, 2, 5] = 127.0          # the largest order the byte holds
    _check(env, chain, params, w, dt, ("qdot_out", "status"), nrep)
    gate = (rng.uniform(size=B) < 0.7).astype(np.int32)    # the publishing variant honours the fresh-q gate
    _check(env, chain, params, w, dt, ALL, nrep, active=gate)


def test_readme_scene_goal_and_normal_with_order_20_obstacles(env):
    """old/README.old:71-75: a goal with an approach normal (object_feeder:248-303: attractor + funnel + order-5 near-goal repeller)
    and `ObstacleP ... 0.05 20` point obstacles, some arms over a table (ObstacleH): field path 2, mixed orders."""
    chain = env["robots"].lwr()
    B = 64 * 5 + 9
    for dt in (np.float32, np.float64):
        w = env["synth"].make_workload(chain, B, 5, seed=47, io_dtype=dt, max_fields=9)
        F = w["fields"]
        F["p"][:, 1:6, 5] = 20.0                                               # the README's obstacles
        F["p"][:, 1:6, 3] = 0.05
        F["id"][:, 6], F["type"][:, 6], F["force"][:, 6] = 2, 5, 30.0          # funnel at the goal along its z axis
        F["p"][:, 6, 0:3] = F["p"][:, 0, [3, 7, 11]]
        F["p"][:, 6, 3:6] = F["p"][:, 0, [2, 6, 10]]
        F["p"][:, 6, 6:10] = [0.15, 10.0, 0.15, 2.0]
        F["id"][:, 7], F["type"][:, 7], F["force"][:, 7] = 3, 2, -10.0         # near-goal repeller, order 5
        F["p"][:, 7, 0:3] = F["p"][:, 0, [3, 7, 11]] - 0.05 * F["p"][:, 0, [2, 6, 10]]
        F["p"][:, 7, 3:6] = [0.2, 0.001, 5.0]
        F["id"][:, 8], F["type"][:, 8], F["force"][:, 8] = 40, 4, -50.0        # a table for every third arm
        F["p"][:, 8] = 0.0
        F["p"][:, 8, 0:8] = [0.0, 0.0, -0.3, 0.02, -0.01, 1.0, 0.05, 5.0]
        w["nfields"][:] = 8
        w["nfields"][::3] = 9
        w = _round(w, dt)
        for flags in (0, 3):
            params = env["abi"].default_params(flags=flags)
            _check(env, chain, params, w, dt, ("qdot_out", "status"), 12, expect_path=2)
            _check(env, chain, params, w, dt, ALL, 12, expect_path=2)


def test_launches_the_mixed_variants_do_not_serve_take_the_general_path(env):
    """A rollout, a per-arm tool and per-arm weights over a batch with differing orders: the general path, same results as the oracle;
    VFIK_MIXED_ORDERS=0 restores ABI 4's classification."""
    import os
    chain = env["robots"].lwr()
    B = 200
    w = env["synth"].make_workload(chain, B, 6, seed=49, io_dtype=np.float64)
    w["fields"]["p"][:, 1:7, 5] = [5, 20, 5, 2, 20, 3]
    params = env["abi"].default_params(flags=3)
    eng = env["engine"].Engine(chain, B, io_dtype=np.float64, max_slots=8, device=0, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    assert eng.field_path == 1 and eng.mixed_orders
    rng = np.random.default_rng(2)
    wq = rng.uniform(0.5, 1.5, (B, 7))
    eng.set_arm_weights(wq=wq)
    got = eng.step_host(w["q"], want=("qdot_out",))
    for b in (0, 77, B - 1):
        p = env["abi"].default_params(flags=3, wq=list(wq[b]))
        ref = env["oc"].cycle_batch(chain, p, w["q"][b:b + 1], w["fields"][b:b + 1], w["nfields"][b:b + 1], want=("qdot_out",))
        assert np.abs(got["qdot_out"][b] - ref["qdot_out"][0]).max() < 1e-9
    eng.close()
    eng = env["engine"].Engine(chain, B, io_dtype=np.float64, max_slots=8, device=0, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    out = eng.rollout_host(w["q"], 5, 1e-3, want=("qdot_out",))
    st = env["oc"].new_states(B, 7)
    q = w["q"].copy()
    for _ in range(5):
        ref = env["oc"].cycle_batch(chain, params, q, w["fields"], w["nfields"], states=st, want=("qdot_out",))
        q = q + 1e-3 * ref["qdot_out"]
    assert np.abs(out["q"] - q).max() < 1e-9
    eng.close()
    os.environ["VFIK_MIXED_ORDERS"] = "0"
    try:
        eng = env["engine"].Engine(chain, B, io_dtype=np.float64, max_slots=8, device=0, params=params)
        eng.set_fields(w["fields"], w["nfields"])
        assert eng.field_path == 0 and not eng.mixed_orders
        eng.close()
    finally:
        del os.environ["VFIK_MIXED_ORDERS"]
