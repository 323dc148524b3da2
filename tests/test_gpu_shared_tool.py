"""The batch's shared tool on the PLAIN kernel variants (a run-time, wave-uniform branch of those kernels): vfclik's normal state is an
arm with a hand on it -- `set tool` of old/README.old:84, scripts/vf:321-332 -- and a tool must not send a launch to the general
variants.  Parity with the CPU oracle for every kernel family a launch with a tool can reach: lean (qdot_out only), publishing lean
(every per-cycle row, /pose_no_tool recomposed from the tool pose), with the aux block and with differing decay orders, the
eight-lanes size (served by one lane per arm: that kernel has no tool), the stepped rollout, both chains, both I/O types."""
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

    class E:
        pass

    e = E()
    e.oc, e.abi, e.engine, e.robots, e.synth = oracle_c, _abi, engine, robots, synth
    return e


def _tool(rot=True):
    t = np.eye(4)
    t[:3, 3] = [0.02, -0.01, 0.2]   # (old/README.old:84 sets 0 0 0.2)
    if rot:
        c, s = np.cos(0.3), np.sin(0.3)
        t[:3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ np.array([[1, 0, 0], [0, np.cos(0.2), -np.sin(0.2)], [0, np.sin(0.2), np.cos(0.2)]])
    return t.reshape(16)


def _check(got, ref, tol, keys):
    for k in keys:
        if k == "status":
            assert np.array_equal(got[k], ref[k])
            continue
        err = np.abs(got[k].astype(np.float64) - ref[k]).max()
        assert np.all(np.isfinite(got[k])) and err < tol, "%s: %.3e" % (k, err)


@pytest.mark.parametrize("robot,B,nobs,dt,tol,flags,want", [
    ("lwr", 65536, 8, np.float32, 1e-6, 0, ("qdot_out", "status")),
    ("lwr", 65536, 8, np.float32, 1e-6, 5, ("qdot_out", "status")),
    ("lwr", 8192 + 37, 8, np.float32, 1e-6, 7, ALL),
    ("lwr", 8192 + 37, 4, np.float64, 1e-9, 5, ALL),
    ("lwr", 1, 4, np.float64, 1e-9, 5, ALL),
    ("lwr", 777, 8, np.float32, 1e-6, 5, ("qdot_out", "pose", "pose_nt", "qdist", "status")),
    ("lwr_dual14", 65536, 16, np.float32, 1e-6, 7, ("qdot_out", "status")),
    ("lwr_dual14", 4096 + 64 + 3, 16, np.float32, 1e-6, 7, ALL),
    ("lwr_dual14", 4096 + 64 + 3, 8, np.float64, 1e-9, 7, ALL),
    ("powercube6", 5000, 6, np.float32, 1e-6, 0, ALL),
])
def test_shared_tool_on_the_plain_variants(env, robot, B, nobs, dt, tol, flags, want):
    chain = getattr(env.robots, robot)()
    w = env.synth.make_workload(chain, B, nobs, seed=51, io_dtype=dt)
    params = env.abi.default_params(flags=flags)
    tool = _tool()
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=max(8, nobs), params=params)
    eng.set_fields(w["fields"], w["nfields"])
    eng.set_tool(tool)
    assert eng.field_path == 1
    assert eng.dh_pattern == 1, "a tool must not take the chain off the DH-pattern kernels"
    got = eng.step_host(w["q"], want=want)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], tool=tool)
    _check(got, ref, tol, want)
    assert np.linalg.norm(ref["pose"][:, [3, 7, 11]] - ref["pose_nt"][:, [3, 7, 11]], axis=1).min() > 0.15   # (the tool is really on)
    # back to no tool: the same handle, the same kernels
    eng.set_tool(np.eye(4).reshape(16))
    got = eng.step_host(w["q"], want=want)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"])
    _check(got, ref, tol, want)
    eng.close()


def test_shared_tool_with_the_aux_block_and_differing_orders(env):
    """The README scene (old/README.old:71-84): goalAndNormal (attractor + funnel + order-5 near-goal repeller), obstacles of order 20, a
    tool -- straight-line field path, order planes, aux block and the tool in one launch."""
    chain = env.robots.lwr()
    B = 4096 + 64 + 9
    w = env.synth.make_workload(chain, B, 5, seed=52, io_dtype=np.float32, max_fields=8)
    F = w["fields"]
    F["id"][:, 6], F["type"][:, 6], F["force"][:, 6] = 2, 5, 30.0
    F["p"][:, 6, 0:3] = F["p"][:, 0, [3, 7, 11]]
    F["p"][:, 6, 3:6] = F["p"][:, 0, [2, 6, 10]]
    F["p"][:, 6, 6:10] = [0.15, 10.0, 0.15, 2.0]
    F["id"][:, 7], F["type"][:, 7], F["force"][:, 7] = 3, 2, -10.0
    F["p"][:, 7, 0:3] = F["p"][:, 0, [3, 7, 11]] - 0.05 * F["p"][:, 0, [2, 6, 10]]
    F["p"][:, 7, 3:6] = [0.05, 0.001, 5.0]
    F["p"][:, 1:6, 5] = 20.0
    F["p"][:, 1:6, 3] = 0.05
    w["nfields"][:] = 8
    params = env.abi.default_params(flags=5)
    tool = _tool(rot=False)
    eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=10, params=params)
    eng.set_fields(F, w["nfields"])
    eng.set_tool(tool)
    assert eng.field_path == 2 and eng.mixed_orders == 1   # (straight-line path with the aux block, order planes)
    for want in (("qdot_out", "status"), ALL):
        got = eng.step_host(w["q"], want=want)
        ref = env.oc.cycle_batch(chain, params, w["q"], F, w["nfields"], tool=tool)
        _check(got, ref, 1e-6, want)
    eng.close()


@pytest.mark.parametrize("robot,dt", [("lwr", np.float32), ("lwr_dual14", np.float32), ("lwr", np.float64)])
def test_rollout_with_the_shared_tool_is_stepped(env, robot, dt):
    """vfik_rollout with a tool: single-cycle launches of the PLAIN kernels integrating q (the in-kernel loop has no register for the tool)."""
    chain = getattr(env.robots, robot)()
    n = chain.n
    B, K, h = 300, 12, 0.01
    w = env.synth.make_workload(chain, B, 4, seed=53, io_dtype=dt)
    params = env.abi.default_params(flags=5)
    tool = _tool()
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    eng.set_tool(tool)
    got = eng.rollout_host(w["q"], K, h, want=("qdot_out",))
    q = w["q"].astype(np.float64).copy()
    states = env.oc.new_states(B, n)
    for _ in range(K):
        ref = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], tool=tool, states=states)
        q = q + h * ref["qdot_out"]
        if dt == np.float32:
            q = q.astype(np.float32).astype(np.float64)
    tol = 1e-8 if dt == np.float64 else 2e-5
    assert np.abs(got["q"] - q).max() < tol, np.abs(got["q"] - q).max()
    eng.close()


def test_equal_per_arm_tools_are_the_shared_tool(env):
    """What a port-level caller does: every arm's /tool bottle forwarded as a per-arm array (vf:321-326).  All rows equal = the batch's
    shared tool: the launch stays on the DH-pattern kernels, and the results are those of the per-arm image (values rounded to float32)."""
    chain = env.robots.lwr()
    B = 5000
    w = env.synth.make_workload(chain, B, 8, seed=54, io_dtype=np.float32)
    params = env.abi.default_params(flags=5)
    tool = _tool()
    eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    eng.set_tool(np.tile(tool, (B, 1)), per_arm=True)
    assert eng.dh_pattern == 1
    got = eng.step_host(w["q"], want=ALL)
    t32 = tool.astype(np.float32).astype(np.float64)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], tool=t32)
    _check(got, ref, 1e-6, ALL)
    tools = np.tile(tool, (B, 1))
    tools[B // 2, 3] += 0.01          # one arm with another hand: per-arm tools, the general variants
    eng.set_tool(tools, per_arm=True)
    assert eng.dh_pattern == 0
    got = eng.step_host(w["q"], want=ALL)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], tool=tools.astype(np.float32).astype(np.float64))
    _check(got, ref, 1e-6, ALL)
    eng.close()


@pytest.mark.parametrize("robot,B,nobs,dt,tol,flags,want,with_tool", [
    ("lwr", 65536, 8, np.float32, 1e-6, 5, ("qdot_out", "status"), False),
    ("lwr", 65536, 8, np.float32, 1e-6, 0, ("qdot_out", "status"), True),
    ("lwr", 8192 + 37, 8, np.float32, 1e-6, 7, ALL, False),
    ("lwr", 8192 + 37, 8, np.float32, 1e-6, 7, ALL, True),
    ("lwr", 1, 4, np.float64, 1e-9, 5, ALL, False),
    ("lwr", 2, 4, np.float64, 1e-9, 5, ALL, True),
    ("lwr", 777, 8, np.float32, 1e-6, 5, ("qdot_out", "pose", "pose_nt", "qdist", "status"), True),
    ("lwr", 5000, 4, np.float64, 1e-9, 5, ALL, False),            # float64 beyond the eight-lanes sizes: the general variants
    ("powercube6", 5000, 6, np.float32, 1e-6, 0, ALL, False),
    ("powercube6", 300, 6, np.float64, 1e-9, 0, ALL, True),
    ("lwr_dual14", 4096 + 64 + 3, 16, np.float32, 1e-6, 7, ALL, False),   # 14 joints: weights take the general variants
])
def test_shared_ik_weights_on_the_plain_variants(env, robot, B, nobs, dt, tol, flags, want, with_tool):
    """IK weights other than one for the whole batch (/weight, vf:295-309: 't' and 'j' weights; handlers.set_wik_*) on the kernels built
    for plain chains (WTSC), alone and together with the shared tool; then back to unit weights on the same handle."""
    chain = getattr(env.robots, robot)()
    n = chain.n
    rng = np.random.default_rng(57)
    w = env.synth.make_workload(chain, B, nobs, seed=55, io_dtype=dt)
    wy = [1.0, 1.0, 1.0, 0.3, 0.3, 0.1]
    wq = list(rng.uniform(0.2, 1.0, n)) + [1.0] * (16 - n)
    params = env.abi.default_params(flags=flags, wy=wy, wq=wq)
    tool = _tool() if with_tool else None
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=max(8, nobs), params=params)
    eng.set_fields(w["fields"], w["nfields"])
    if with_tool:
        eng.set_tool(tool)
    assert eng.field_path == 1 and eng.dh_pattern == 1
    got = eng.step_host(w["q"], want=want)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], tool=tool)
    _check(got, ref, tol, want)
    plain = env.abi.default_params(flags=flags)
    ref1 = env.oc.cycle_batch(chain, plain, w["q"], w["fields"], w["nfields"], tool=tool)
    assert np.abs(ref["qdot_out"] - ref1["qdot_out"]).max() > 1e-3    # (the weights really act)
    eng.set_params(wy=[1.0] * 6, wq=[1.0] * 16)
    got = eng.step_host(w["q"], want=want)
    _check(got, ref1, tol, want)
    eng.close()


def test_shared_weights_with_the_aux_block_and_differing_orders(env):
    """The README scene with a tool AND joint weights: order planes, aux block, TOOLC and WTSC in one launch."""
    chain = env.robots.lwr()
    B = 4096 + 64 + 9
    w = env.synth.make_workload(chain, B, 5, seed=58, io_dtype=np.float32, max_fields=8)
    F = w["fields"]
    F["id"][:, 6], F["type"][:, 6], F["force"][:, 6] = 2, 5, 30.0
    F["p"][:, 6, 0:3] = F["p"][:, 0, [3, 7, 11]]
    F["p"][:, 6, 3:6] = F["p"][:, 0, [2, 6, 10]]
    F["p"][:, 6, 6:10] = [0.15, 10.0, 0.15, 2.0]
    F["id"][:, 7], F["type"][:, 7], F["force"][:, 7] = 3, 2, -10.0
    F["p"][:, 7, 0:3] = F["p"][:, 0, [3, 7, 11]] - 0.05 * F["p"][:, 0, [2, 6, 10]]
    F["p"][:, 7, 3:6] = [0.05, 0.001, 5.0]
    F["p"][:, 1:6, 5] = 20.0
    F["p"][:, 1:6, 3] = 0.05
    w["nfields"][:] = 8
    params = env.abi.default_params(flags=5, wq=[1, 0.5, 1, 0.7, 1, 0.4, 1] + [1.0] * 9)
    tool = _tool(rot=False)
    eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=10, params=params)
    eng.set_fields(F, w["nfields"])
    eng.set_tool(tool)
    assert eng.field_path == 2 and eng.mixed_orders == 1
    for want in (("qdot_out", "status"), ALL):
        got = eng.step_host(w["q"], want=want)
        ref = env.oc.cycle_batch(chain, params, w["q"], F, w["nfields"], tool=tool)
        _check(got, ref, 1e-6, want)
    eng.close()


def test_equal_per_arm_weights_and_bridge_state_are_batch_wide(env):
    """What the port layer produces when every arm's handlers send the same /weight and /bridge/weight (handlers.py:189-230,481-497): per-arm
    arrays with equal rows.  The library stores them as the batch's (the DH-pattern kernels keep serving the launch; results are those of a
    handle configured batch-wide, to the bit); one arm that differs brings the per-arm path back, and clearing it the batch-wide one."""
    chain = env.robots.lwr()
    B = 6000
    f = env.abi
    w = env.synth.make_workload(chain, B, 8, seed=59, io_dtype=np.float32)
    wy = [1.0, 1.0, 1.0, 0.5, 0.5, 0.25]
    wq = [1.0, 0.5, 1.0, 0.75, 1.0, 0.5, 1.0]
    mw = [0.75, 0.5, 0.0, 0.0, 0.0, 0.0]
    shared = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER | f.F_LIMITER, wy=wy, wq=wq + [1.0] * 9, mix_w=mw, max_vel=0.25)   # (values exact in float32: the per-arm image holds them in the I/O type)
    e0 = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=shared)
    e0.set_fields(w["fields"], w["nfields"])
    want = ("qdot_out", "qdot_null", "pose", "status")
    base = e0.step_host(w["q"], want=want)
    e0.close()
    ref = env.oc.cycle_batch(chain, shared, w["q"], w["fields"], w["nfields"])
    _check(base, ref, 1e-6, want)
    eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=f.default_params(flags=f.F_NULLSPACE | f.F_MIXER | f.F_LIMITER))
    eng.set_fields(w["fields"], w["nfields"])
    eng.set_arm_weights(wy=np.tile(wy, (B, 1)), wq=np.tile(wq, (B, 1)))
    eng.set_mixer_weights(np.tile(mw, (B, 1)))
    eng.set_max_vel(np.full(B, 0.25))
    assert eng.dh_pattern == 1
    got = eng.step_host(w["q"], want=want)
    for k in want:
        assert np.array_equal(got[k], base[k]), k
    # one arm with other joint weights and another mixer row: per arm again (the general variants), same numbers for every other arm
    wq2 = np.tile(wq, (B, 1))
    wq2[17] = [0.5] * 7
    eng.set_arm_weights(wy=np.tile(wy, (B, 1)), wq=wq2)
    mw2 = np.tile(mw, (B, 1))
    mw2[99] = [0.02, 0.0, 0.0, 0.0, 0.0, 0.0]    # (small enough to stay under the limiter: otherwise only the direction would show)
    eng.set_mixer_weights(mw2)
    assert eng.dh_pattern == 0
    got2 = eng.step_host(w["q"], want=want)
    others = np.ones(B, bool)
    others[[17, 99]] = False
    assert np.abs(got2["qdot_out"][others] - base["qdot_out"][others]).max() < 1e-6
    assert np.abs(got2["qdot_out"][17] - base["qdot_out"][17]).max() > 1e-4 and np.abs(got2["qdot_out"][99] - base["qdot_out"][99]).max() > 1e-4
    # ... and back
    eng.set_arm_weights(wy=np.tile(wy, (B, 1)), wq=np.tile(wq, (B, 1)))
    eng.set_mixer_weights(np.tile(mw, (B, 1)))
    assert eng.dh_pattern == 1
    got3 = eng.step_host(w["q"], want=want)
    for k in ("qdot_out", "pose", "status"):
        assert np.array_equal(got3[k], base[k]), k
    eng.close()
