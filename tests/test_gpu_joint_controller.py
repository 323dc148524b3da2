"""Joint P controller (scripts/joint_p_controller:89-146) as mixer channel 2 and the LWR command form of
the bridge (bridge:198-203), fused into the control cycle; checked against the oracle restatements."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c, vfik_numpy
    from vfclik_amd import _abi, engine, robots, synth
    return dict(oc=oracle_c, vn=vfik_numpy, abi=_abi, engine=engine, robots=robots, synth=synth)


def _expected(env, chain, params, w, ref, ctrl, mixw, ext=None):
    oc, abi = env["oc"], env["abi"]
    jc, at_goal = oc.joint_p(ref, w["q"], chain.q_lo, chain.q_hi, params.jp_kp, params.jp_delta)
    B, n = w["q"].shape
    ext_all = np.zeros((4, B, n)) if ext is None else ext.copy()
    ext_all[0] = jc
    r = oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], null_control=ctrl, ext_cmd=ext_all)
    return r, jc, at_goal


@pytest.mark.parametrize("robot,dt", [("lwr", np.float64), ("lwr", np.float32), ("powercube6", np.float64), ("lwr_dual14", np.float64)])
def test_joint_p_controller_is_mixer_channel_2(env, robot, dt):
    abi = env["abi"]
    chain = env["robots"].by_name(robot)
    B = 300
    w = env["synth"].make_workload(chain, B, 3, seed=11, io_dtype=dt)
    rng = np.random.default_rng(5)
    # references: some beyond the limits (clamped), some within delta of q (at_goal), some far
    ref = rng.uniform(1.3 * chain.q_lo, 1.3 * chain.q_hi, (B, chain.n)).astype(dt).astype(np.float64)
    ref[:40] = (w["q"][:40] + rng.uniform(-0.05, 0.05, (40, chain.n))).astype(dt)
    ref[40:60] = (w["q"][40:60] - 1.0).astype(dt)  # signed comparison: large negative errors still count as reached
    ctrl = rng.uniform(-1, 1, (B, 4)).astype(dt).astype(np.float64)
    if chain.n - 6 > 1:
        ctrl[:] = 0.0  # nullity > 1: the reference's SVD basis is not unique, /control is not honoured
    params = abi.default_params(flags=abi.F_NULLSPACE | abi.F_MIXER, mix_w=[1.0, 0.5, 0.7, 0.0, 0.25, 0.0], jp_kp=0.8)
    ext = rng.uniform(-1, 1, (4, B, chain.n)).astype(dt).astype(np.float64)
    eng = env["engine"].Engine(chain, B, io_dtype=dt, max_slots=8, device=0, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    for ch in range(4):
        eng.set_ext_cmd(2 + ch, ext[ch])  # channel 2's external command must be ignored once q_ref is given
    got = eng.step_host(w["q"], null_control=ctrl, q_ref=ref, want=("qdot_vf", "qdot_null", "qdot_out", "status"))
    exp, jc, at_goal = _expected(env, chain, params, w, ref, ctrl, None, ext)
    tol = 1e-6 if dt == np.float64 else 2e-5
    assert np.abs(got["qdot_out"] - exp["qdot_out"]).max() < tol
    assert np.array_equal((got["status"] & abi.ST_JOINT_AT_GOAL) != 0, at_goal != 0)
    assert at_goal[:60].all() and not at_goal.all()
    if dt == np.float64:
        # the mixer sum itself is bit-exact given the same channel values
        cmd = np.stack([got["qdot_vf"], got["qdot_null"], jc, ext[1], ext[2], ext[3]])
        for b in (0, 17, B - 1):
            assert np.array_equal(env["oc"].mix(cmd[:, b], np.array(list(params.mix_w))), got["qdot_out"][b])
    eng.close()


def test_numpy_and_c_restatements_agree(env):
    chain = env["robots"].lwr()
    rng = np.random.default_rng(2)
    q = rng.uniform(chain.q_lo, chain.q_hi, (50, 7))
    ref = rng.uniform(1.5 * chain.q_lo, 1.5 * chain.q_hi, (50, 7))
    out, flags = env["oc"].joint_p(ref, q, chain.q_lo, chain.q_hi, 1.5, 0.087)
    limits = list(zip(chain.q_lo, chain.q_hi))
    for b in range(50):
        o, reached = env["vn"].joint_p_controller(ref[b].tolist(), q[b].tolist(), limits, 1.5, 0.087)
        assert np.array_equal(np.array(o), out[b]) and bool(flags[b]) == reached


def test_lwr_command_form(env):
    abi = env["abi"]
    chain = env["robots"].lwr()
    B = 256
    w = env["synth"].make_workload(chain, B, 2, seed=4, io_dtype=np.float64)
    rng = np.random.default_rng(9)
    q_cmded = w["q"] + rng.normal(0, 0.01, w["q"].shape)
    params = abi.default_params(flags=abi.F_MIXER | abi.F_LIMITER, max_vel=0.2)
    eng = env["engine"].Engine(chain, B, io_dtype=np.float64, max_slots=8, device=0, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    mixw = np.tile(np.array([1.0, 1.0, 0, 0, 0, 0]), (B, 1))
    mixw[::5] = 0.0  # these arms have no controller: direct_control (bridge:604)
    eng.set_mixer_weights(mixw)
    got = eng.step_host(w["q"], q_cmded=q_cmded, want=("qdot_out", "status"))
    # oracle: per-arm weights -> run it twice (weights are batch-wide there) and pick
    e1 = env["oc"].cycle_batch(chain, params, w["q"], w["fields"], w["nfields"])
    p0 = abi.default_params(flags=abi.F_MIXER | abi.F_LIMITER, max_vel=0.2, mix_w=[0] * 6)
    e0 = env["oc"].cycle_batch(chain, p0, w["q"], w["fields"], w["nfields"])
    direct = np.zeros(B, dtype=bool)
    direct[::5] = True
    lim = np.where(direct[:, None], e0["qdot_out"], e1["qdot_out"])
    exp = env["oc"].lwr_cmd(lim, w["q"], q_cmded, direct)
    assert np.abs(got["qdot_out"] - exp).max() < 1e-6
    assert np.array_equal(got["qdot_out"][::5], np.zeros_like(got["qdot_out"][::5]))
    # numpy twin of the command form
    for b in (1, 2, 5):
        assert np.allclose(env["vn"].lwr_command(lim[b].tolist(), w["q"][b].tolist(), q_cmded[b].tolist(), bool(direct[b])), exp[b], atol=0, rtol=0)
    # the position form is not something to integrate: the rollout refuses it
    with pytest.raises(env["engine"].VfikError):
        io = eng.make_io(0x1000, q_cmded=0x1000)
        eng.rollout(io, 3, 1e-3)
    eng.close()


def test_rollout_with_joint_controller_converges_to_the_reference(env):
    """Joint control as handlers.set_joint_control sets it (weights [0,0,1,...], handlers.py:189-204):
    q' = kp (ref - q) integrated on the device converges to the clamped reference."""
    abi = env["abi"]
    chain = env["robots"].lwr()
    B = 128
    w = env["synth"].make_workload(chain, B, 0, seed=3, io_dtype=np.float64)
    rng = np.random.default_rng(1)
    ref = rng.uniform(1.2 * chain.q_lo, 1.2 * chain.q_hi, (B, 7))
    params = abi.default_params(flags=abi.F_MIXER, mix_w=[0, 0, 1, 0, 0, 0], jp_kp=1.5)
    eng = env["engine"].Engine(chain, B, io_dtype=np.float64, max_slots=4, device=0, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    out = eng.rollout_host(w["q"], 4000, 2e-3, q_ref=ref, want=("qdot_out", "status"))
    target = np.clip(ref, chain.q_lo, chain.q_hi)
    # e(t) = e0 exp(-kp t): after 8 s at kp 1.5 the error is e0 * 6e-6
    assert np.abs(out["q"] - target).max() < 1e-4
    assert ((out["status"] & abi.ST_JOINT_AT_GOAL) != 0).all()
    eng.close()


def test_kept_reference_equals_the_reference_clamp(env, golden_dir):
    """io.q_ref_out (the reference the controller KEEPS, joint_p_controller:121) against the reference's own check_limits outputs
    (tests/golden/jpctrl_golden.npz): limits that move with the position go in as this cycle's io.q_lo / io.q_hi."""
    import os
    abi = env["abi"]
    g = np.load(os.path.join(golden_dir, "jpctrl_golden.npz"))
    n_of = np.sum(~np.isnan(g["ref"]), axis=1)
    for robot, n in (("powercube6", 6), ("lwr", 7), ("lwr_dual14", 14)):
        sel = np.nonzero(n_of == n)[0]
        chain = env["robots"].by_name(robot)
        B = len(sel)
        ref, lim, want = g["ref"][sel, :n], g["limits"][sel, :n], g["ref_out"][sel, :n]
        w = env["synth"].make_workload(chain, B, 1, seed=3, io_dtype=np.float64)
        params = abi.default_params(flags=abi.F_MIXER, mix_w=[0, 0, 1, 0, 0, 0], jp_kp=1.0)
        eng = env["engine"].Engine(chain, B, io_dtype=np.float64, max_slots=2, device=0, params=params)
        eng.set_fields(w["fields"], w["nfields"])
        q = np.clip(g["cur_pos"][sel, :n], 0.9 * chain.q_lo, 0.9 * chain.q_hi)
        got = eng.step_host(q, q_ref=ref, q_lo=lim[:, :, 0].copy(), q_hi=lim[:, :, 1].copy(), want=("qdot_out", "q_ref_out"))
        assert np.array_equal(got["q_ref_out"], want), robot
        assert np.abs(got["qdot_out"] - (want - q)).max() < 1e-12   # channel 2 alone: kp (clamp(ref) - q), kp = 1
        eng.close()


@pytest.mark.parametrize("robot,dt", [("lwr", np.float64), ("lwr", np.float32), ("lwr_dual14", np.float64), ("powercube6", np.float64)])
def test_limiter_and_lwr_command_form_equal_the_reference_record(env, golden_dir, robot, dt):
    """The fused limiter + LWR command form against what `LWR_Bridge.set_vel` of the REFERENCE returned for the same numbers
    (tests/golden/bridge_golden.npz, made by running bridge:182-210; make_golden_blocks.py).  The record's joint velocities go in
    as the /bridge/mechanismcmd channel with mixer weights [0,0,0,1,0,0], each arm with its record's max_vel, last_q as q and
    last_qcmded as the robot's echo; `direct_control` arms have every mixer weight 0 (bridge:604), which makes their mixed command
    0 -- so of the record's direct cases only those with a zero command are reachable through the bridge loop, and only those are
    compared here (the others pin the CPU oracles, tests/test_reference_blocks.py)."""
    import os
    abi = env["abi"]
    chain = env["robots"].by_name(robot)
    g = np.load(os.path.join(golden_dir, "bridge_golden.npz"))
    sel = np.nonzero((g["n"] == chain.n) & (~g["direct"] | (np.nan_to_num(np.abs(g["qdot"])).max(axis=1) == 0.0)))[0]
    B = len(sel)
    assert B >= 40 and g["direct"][sel].sum() >= 4
    n = chain.n
    qdot, q, qc, exp = (g[k][sel, :n].astype(dt).astype(np.float64) for k in ("qdot", "last_q", "last_qcmded", "cmd"))
    mv = g["max_vel"][sel].astype(dt).astype(np.float64)
    if dt == np.float32:    # the kernel sees float32-rounded inputs: restate bridge:188-203 on those (the oracle is pinned bit-exact)
        exp = np.array([env["vn"].lwr_command(env["vn"].limiter(qdot[b].tolist(), float(mv[b]))[0],
                                              q[b].tolist(), qc[b].tolist(), bool(g["direct"][sel][b])) for b in range(B)])
    w = env["synth"].make_workload(chain, B, 1, seed=2, io_dtype=dt)
    params = abi.default_params(flags=abi.F_MIXER | abi.F_LIMITER, max_vel=1.0)
    eng = env["engine"].Engine(chain, B, io_dtype=dt, max_slots=4, device=0, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    mixw = np.tile(np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0]), (B, 1))
    mixw[g["direct"][sel]] = 0.0
    eng.set_mixer_weights(mixw)
    eng.set_max_vel(g["max_vel"][sel])
    eng.set_ext_cmd(3, qdot)
    got = eng.step_host(q, q_cmded=qc, want=("qdot_out", "status"))
    err = np.abs(got["qdot_out"] - exp).max()
    assert err < (1e-12 if dt == np.float64 else 5e-7), err
    scaled = np.abs(qdot).max(axis=1) > mv
    assert 5 < scaled.sum() < B - 5
    assert np.array_equal((got["status"] & abi.ST_LIMITED) != 0, scaled)
    eng.close()
