"""Closed-loop rollout (vfik_rollout): K control cycles in one launch with q integrated on
the device, against (a) the oracle stepped K times on the host with the same Euler update and (b) K
single-cycle launches of the HIP path itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, engine, robots, synth

    class E:
        pass

    e = E()
    e.oc, e.abi, e.engine, e.robots, e.synth = oracle_c, _abi, engine, robots, synth
    return e


def _oracle_rollout(env, chain, params, w, K, dt, ctrl=None, clamp=False):
    q = w["q"].copy()
    states = env.oc.new_states(q.shape[0], chain.n) if params.flags & env.abi.F_NULLSPACE else None
    status = np.zeros(q.shape[0], dtype=np.int32)
    for _ in range(K):
        ref = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], null_control=ctrl, states=states)
        status |= ref["status"]
        qn = q + dt * ref["qdot_out"]
        if clamp:
            qn = np.clip(qn, chain.q_lo, chain.q_hi)
        last_q, q = q, qn
    return q, ref, status


@pytest.mark.parametrize("robot,nobs,flags", [("lwr", 8, 0), ("lwr", 3, 1 | 4), ("lwr_dual14", 12, 1 | 2 | 4), ("powercube6", 2, 4 | 8)])
def test_rollout_matches_stepped_oracle(env, robot, nobs, flags):
    chain = env.robots.by_name(robot)
    B, K, dt = 1024, 40, 0.01
    w = env.synth.make_workload(chain, B, nobs, seed=31, io_dtype=np.float64)
    params = env.abi.default_params(flags=flags, max_vel=0.7)
    ctrl = np.random.default_rng(5).uniform(-1, 1, (B, 4)) if (flags & 1) and chain.n == 7 else None
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=16, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    got = eng.rollout_host(w["q"], K, dt, null_control=ctrl, clamp=True, want=("qdot_out", "qdot_vf", "pose", "qdist", "status"))
    q_ref, ref, st_ref = _oracle_rollout(env, chain, params, w, K, dt, ctrl, clamp=True)
    assert np.abs(got["q"] - q_ref).max() < 1e-8, np.abs(got["q"] - q_ref).max()
    for k in ("qdot_out", "qdot_vf", "pose", "qdist"):
        assert np.abs(got[k] - ref[k]).max() < 1e-7, (k, np.abs(got[k] - ref[k]).max())
    assert np.array_equal(got["status"], st_ref)  # status bits accumulate over the cycles
    assert np.abs(got["q"] - w["q"]).max() > 0.05   # the arms did move
    eng.close()


@pytest.mark.parametrize("robot,dt_io,flags", [("lwr", np.float32, 0), ("lwr", np.float64, 0), ("lwr", np.float64, 1 | 4), ("powercube6", np.float32, 1 | 4 | 8),
                                               ("lwr_dual14", np.float64, 1 | 2 | 4)])
def test_lean_rollout_only_q_and_qdot_out(env, robot, dt_io, flags):
    """A rollout that asks for nothing but q and qdot_out runs the LEAN variant (7 joints and fewer; longer chains are
    stepped launches either way): same trajectory as the oracle stepped on the host."""
    chain = env.robots.by_name(robot)
    B, K, dt = 777, 30, 0.01
    w = env.synth.make_workload(chain, B, 5, seed=37, io_dtype=dt_io)
    params = env.abi.default_params(flags=flags, max_vel=0.7)
    eng = env.engine.Engine(chain, B, io_dtype=dt_io, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    got = eng.rollout_host(w["q"], K, dt, clamp=True, want=("qdot_out",))
    q_ref, ref, st_ref = _oracle_rollout(env, chain, params, w, K, dt, None, clamp=True)
    tol_q, tol_v = (1e-8, 1e-7) if dt_io == np.float64 else (2e-6, 2e-5)
    assert np.abs(got["q"] - q_ref).max() < tol_q, np.abs(got["q"] - q_ref).max()
    assert np.abs(got["qdot_out"] - ref["qdot_out"]).max() < tol_v
    # the same with the status bits, which accumulate over the cycles (also over the launches of a stepped rollout)
    eng.reset_state()
    got = eng.rollout_host(w["q"], K, dt, clamp=True, want=("qdot_out", "status"))
    assert np.abs(got["q"] - q_ref).max() < tol_q
    if dt_io == np.float64:
        assert np.array_equal(got["status"], st_ref)
    eng.close()


@pytest.mark.parametrize("robot", ["lwr", "lwr_dual14"])
def test_rollout_on_the_general_field_path(env, robot):
    """Mixed field sets (a funnel, a hemisphere, fractional and differing decay orders, 11 slots) send the batch down
    the general field path: the in-kernel loop (7 joints) and the stepped launches (14) against the stepped oracle."""
    chain = env.robots.by_name(robot)
    B, K, dt = 600, 20, 0.01
    w = env.synth.make_workload(chain, B, 7, seed=41, io_dtype=np.float64, max_fields=10)
    F = w["fields"]
    F["p"][::3, 2, 5] = 2.0          # another integer order on every third arm
    F["p"][1::3, 3, 5] = 2.5         # a fractional order
    F["id"][:, 8], F["type"][:, 8], F["force"][:, 8] = 2, 5, 30.0   # funnel
    F["p"][:, 8, :10] = [0.3, 0.2, 0.5, 0.1, 0.2, -0.9, 0.15, 10.0, 0.15, 2.0]
    F["id"][:, 9], F["type"][:, 9], F["force"][:, 9] = 40, 4, -50.0  # hemisphere
    F["p"][:, 9, :8] = [0.2, -0.1, -0.5, 0.05, -0.02, 1.0, 0.05, 5.0]
    w["nfields"][:] = 10
    params = env.abi.default_params(flags=env.abi.F_MIXER | env.abi.F_LIMITER, max_vel=0.7)
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=12, params=params)
    eng.set_fields(F, w["nfields"])
    assert eng.lib.vfik_slots_in_use(eng.h) == 11
    got = eng.rollout_host(w["q"], K, dt, clamp=True, want=("qdot_out", "status"))
    q_ref, ref, st_ref = _oracle_rollout(env, chain, params, w, K, dt, None, clamp=True)
    assert np.abs(got["q"] - q_ref).max() < 1e-8, np.abs(got["q"] - q_ref).max()
    assert np.abs(got["qdot_out"] - ref["qdot_out"]).max() < 1e-7
    assert np.array_equal(got["status"], st_ref)
    eng.close()


@pytest.mark.parametrize("flags", [0, 1 | 4])
def test_rollout_with_tool_and_weights(env, flags):
    """The general (non-PLAIN) kernel variants in the rollout loop: a tool offset and non-unit IK weights."""
    chain = env.robots.lwr()
    B, K, dt = 512, 25, 0.01
    w = env.synth.make_workload(chain, B, 4, seed=43, io_dtype=np.float64)
    params = env.abi.default_params(flags=flags, wy=[1, 1, 1, 0.3, 0.3, 0.1], wq=[1, 0.5, 1, 0.7, 1, 0.4, 1] + [1.0] * 9)
    tool = np.eye(4)
    tool[:3, 3] = [0.02, -0.01, 0.2]
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    eng.set_tool(tool.reshape(16))
    got = eng.rollout_host(w["q"], K, dt, want=("qdot_out", "pose"))
    q = w["q"].copy()
    states = env.oc.new_states(B, 7) if flags & 1 else None
    for _ in range(K):
        ref = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], tool=tool.reshape(16), states=states)
        q = q + dt * ref["qdot_out"]
    assert np.abs(got["q"] - q).max() < 1e-8
    assert np.abs(got["qdot_out"] - ref["qdot_out"]).max() < 1e-7 and np.abs(got["pose"] - ref["pose"]).max() < 1e-8
    eng.close()


def test_rollout_equals_repeated_single_launches(env):
    chain = env.robots.lwr()
    B, K, dt = 4096, 25, 0.004
    w = env.synth.make_workload(chain, B, 8, seed=32, io_dtype=np.float32)
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    ctrl = np.random.default_rng(6).uniform(-1, 1, (B, 4)).astype(np.float32)
    e1 = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=params)
    e2 = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=params)
    for e in (e1, e2):
        e.set_fields(w["fields"], w["nfields"])
    # float64 integration on the host between launches vs float64 integration in registers: the only
    # difference is that the stepped path rounds q to float32 at every launch boundary
    q = w["q"].astype(np.float64)
    for _ in range(K):
        out = e1.step_host(q.astype(np.float32), null_control=ctrl, want=("qdot_out",))
        q = q.astype(np.float32).astype(np.float64) + dt * out["qdot_out"].astype(np.float64)
    got = e2.rollout_host(w["q"], K, dt, null_control=ctrl, want=("qdot_out",))
    assert np.abs(got["q"].astype(np.float64) - q).max() < 2e-5
    assert np.abs(got["qdot_out"] - out["qdot_out"]).max() < 5e-4
    e1.close()
    e2.close()


def test_rollout_argument_errors(env):
    chain = env.robots.lwr()
    eng = env.engine.Engine(chain, 64, io_dtype=np.float64, max_slots=2)
    q = np.zeros((64, 7))
    with pytest.raises(env.engine.VfikError, match="n_cycles"):
        eng.rollout_host(q, 0, 0.01)
    with pytest.raises(env.engine.VfikError):
        eng.rollout_host(q, 10, float("nan"))
    out = eng.rollout_host(q, 3, 0.01)  # no fields: nothing moves
    assert np.all(out["q"] == 0.0) and np.all(out["qdot_out"] == 0.0)
    eng.close()
