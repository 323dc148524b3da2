"""ABI 3: the fresh-q gate and per-arm, per-cycle joint limits.

The reference advances an arm only when that arm's own joint angles arrived (scripts/vf:312-313,
scripts/nullspace:162-163, scripts/debug_jointlimits:61) and re-reads the joint limits every cycle
(scripts/nullspace:167 `rob.get_limits()`, scripts/joint_p_controller:80 `config.updateJntLimits(cur_pos)`).
HIP path vs the oracle driven the same way (tolerances as in test_gpu_parity.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL64 = 1e-9
TOL32 = 1e-6
OUTS = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status")


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c, vfik_numpy
    from vfclik_amd import _abi, engine, robots, synth

    class E:
        pass

    e = E()
    e.oc, e.vn, e.abi, e.engine, e.robots, e.synth = oracle_c, vfik_numpy, _abi, engine, robots, synth
    return e


def _close(got, ref, tol, keys, rows=None):
    for k in keys:
        a, b = got[k], ref[k]
        if rows is not None:
            a, b = a[rows], b[rows]
        if k == "status":
            assert np.array_equal(a, b), (k, np.nonzero(a != b)[0][:8])
        else:
            err = np.abs(a.astype(np.float64) - b)
            assert err.max() < tol, "%s: %.3e" % (k, err.max())


@pytest.mark.parametrize("dt,tol", [(np.float64, TOL64), (np.float32, TOL32)])
def test_silent_arms_keep_their_state_and_publish_nothing(env, dt, tol):
    """Half of the arms get no joint angles for 3 of 8 cycles: their output rows stay what they were, their
    nullspace sign memory does not advance, and every later output equals the oracle's that skipped those cycles."""
    chain = env.robots.lwr()
    B, K = 192, 8
    w = env.synth.make_workload(chain, B, 3, seed=5, io_dtype=dt)
    params = env.abi.default_params(flags=env.abi.F_NULLSPACE | env.abi.F_MIXER)
    rng = np.random.default_rng(11)
    ctrl = rng.uniform(-1, 1, (B, 4)).astype(dt).astype(np.float64)
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=4, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    states = env.oc.new_states(B, chain.n)
    silent = np.arange(B) % 2 == 1
    q = w["q"].copy()
    got = ref = None
    for t in range(K):
        active = np.ones(B, dtype=bool) if t not in (2, 3, 4) else ~silent
        prev = None if got is None else {k: v.copy() for k, v in got.items()}
        got = eng.step_host(q, null_control=ctrl, want=OUTS, active=active, into=got)
        ref = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], null_control=ctrl, states=states, active=active, into=ref)
        ref.pop("states")
        _close(got, ref, tol, OUTS)              # gated rows included: both sides left them alone
        if prev is not None and not active.all():
            for k in OUTS:                        # bit-for-bit what the previous cycle left
                assert np.array_equal(got[k][~active], prev[k][~active]), k
        # every arm moves (a big step: the nullspace vector turns, so a stale sign memory would show)
        q = (q + 0.05 * got["qdot_out"].astype(np.float64) + rng.normal(0, 0.05, q.shape)).astype(dt).astype(np.float64)
        q = np.clip(q, chain.q_lo * 0.95, chain.q_hi * 0.95).astype(dt).astype(np.float64)
    assert np.abs(ref["qdot_null"]).max() > 1e-3
    eng.close()


def test_gate_on_device_pointers_and_status_rows(env):
    """vfik_step with device pointers: a gated arm's status and qdot rows are not written."""
    import torch
    chain = env.robots.lwr()
    B = 130  # ragged: not a multiple of the wave size
    w = env.synth.make_workload(chain, B, 2, seed=9, io_dtype=np.float32)
    params = env.abi.default_params()
    eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=2, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(w["q"].astype(np.float32)).to(dev)
    qd = torch.full((B, 7), 7.5, dtype=torch.float32, device=dev)
    st = torch.full((B,), -3, dtype=torch.int32, device=dev)
    act_h = (np.arange(B) % 3 != 0).astype(np.int32)
    act = torch.from_numpy(act_h).to(dev)
    eng.step(eng.make_io(q, active=act, qdot_out=qd, status=st))
    eng.sync()
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out", "status"))
    on = act_h != 0
    assert np.abs(qd.cpu().numpy()[on] - ref["qdot_out"][on]).max() < TOL32
    assert np.array_equal(st.cpu().numpy()[on], ref["status"][on])
    assert np.all(qd.cpu().numpy()[~on] == 7.5) and np.all(st.cpu().numpy()[~on] == -3)
    eng.close()


def test_tracking_error_history_of_a_silent_arm(env):
    """The estimator of vf:349-428 sits inside `if qInBottle`: one of two arm groups stays silent for 10 cycles;
    its history, and so every later /track_error, equals a per-arm estimator that never saw those cycles."""
    vn = env.vn
    chain = env.robots.lwr()
    B, K, dt = 64, 30, 1.0 / 150.0
    w = env.synth.make_workload(chain, B, 2, seed=41, io_dtype=np.float64)
    params = env.abi.default_params()
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=4, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    d_pose, d_v6, d_out, d_act = eng.dev_alloc(B * 16 * 8), eng.dev_alloc(B * 6 * 8), eng.dev_alloc(B * 8 * 8), eng.dev_alloc(B * 4)
    eng.h2d(d_out, np.zeros((B, 8)))
    est = [vn.TrackingError() for _ in range(B)]
    q = w["q"].copy()
    silent = np.arange(B) >= B // 2
    last = np.zeros((B, 8))
    compared = 0
    for t in range(K):
        active = np.ones(B, dtype=bool) if not (8 <= t < 18) else ~silent
        out = eng.step_host(q, want=("qdot_out", "pose", "v6"))
        eng.h2d(d_pose, out["pose"])
        eng.h2d(d_v6, out["v6"])
        eng.h2d(d_act, active.astype(np.int32))
        eng.track_error(d_pose, d_v6, d_out, d_act)
        got = np.zeros((B, 8))
        eng.d2h(got, d_out)
        for b in range(B):
            if not active[b]:
                assert np.array_equal(got[b], last[b])  # nothing published
                continue
            r = est[b].update(vn.listToKdlFrame(out["pose"][b]), out["v6"][b, :3], out["v6"][b, 3:])
            if r is None:
                assert np.all(got[b] == 0.0)
            else:
                compared += 1
                assert np.abs(got[b] - r).max() < 1e-9, (t, b)
        last = got
        q = q + dt * 0.8 * out["qdot_out"]
    assert compared > B * (K - 16)
    for p in (d_pose, d_v6, d_out, d_act):
        eng.dev_free(p)
    eng.close()


def _moving_limits(chain, q, t, rng, dt):
    """Configuration-dependent limits in the manner of joint_p_controller:121-125: a window that depends on the
    arm's own pose and on the cycle, different for every arm, always lo < hi."""
    B, n = q.shape
    mid = 0.5 * (chain.q_lo + chain.q_hi) + 0.2 * np.sin(q[:, ::-1] + 0.3 * t)
    half = 0.5 * (chain.q_hi - chain.q_lo) * rng.uniform(0.35, 1.0, (B, n))
    lo, hi = (mid - half).astype(dt).astype(np.float64), (mid + half).astype(dt).astype(np.float64)
    assert np.all(hi - lo > 0.1)
    return lo, hi


@pytest.mark.parametrize("robot,dt,tol", [("lwr", np.float64, TOL64), ("lwr", np.float32, TOL32), ("lwr_dual14", np.float64, TOL64),
                                          ("lwr_dual14", np.float32, TOL32)])
def test_limits_that_differ_per_arm_and_change_between_cycles(env, robot, dt, tol):
    """check_limits (nullspace:120-131), distToCenter (debug_jointlimits:66-67), the joint-limit task and the joint
    controller's clamp (joint_p_controller:79-89, the clamped reference kept, :121) all read THIS cycle's limits."""
    abi = env.abi
    chain = env.robots.by_name(robot)
    n = chain.n
    B = 256
    w = env.synth.make_workload(chain, B, 3, seed=21, io_dtype=dt)
    params = abi.default_params(flags=abi.F_NULLSPACE | abi.F_JOINT_LIMIT_TASK | abi.F_MIXER, mix_w=[1, 1, 0.5, 0, 0, 0], jl_gain=0.8)
    rng = np.random.default_rng(3)
    # /control needs the reference's SVD basis, which is unique only for nullity 1 (VFIK_ST_NULL_AMBIGUOUS otherwise):
    # the 14-joint chain runs the joint-limit task alone, as BASELINE's C5 does
    ctrl = rng.uniform(-2, 2, (B, 4)).astype(dt).astype(np.float64) if n <= 7 else None
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=4, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    states = env.oc.new_states(B, n)
    q = w["q"].copy()
    ref_j = rng.uniform(chain.q_lo * 1.2, chain.q_hi * 1.2, (B, n)).astype(dt).astype(np.float64)  # some beyond the limits
    stops = clamps = 0
    for t in range(3):
        lo, hi = _moving_limits(chain, q, t, rng, dt)
        got = eng.step_host(q, null_control=ctrl, q_ref=ref_j, q_lo=lo, q_hi=hi, want=OUTS + ("q_ref_out",))
        # the oracle: the controller's command is channel 2 (joint_p_controller:78); clamp against this cycle's limits
        jc = np.zeros((B, n))
        at_goal = np.zeros(B, dtype=np.int32)
        kept = np.clip(ref_j, lo, hi)
        for b in range(B):
            o, f = env.oc.joint_p(ref_j[b:b + 1], q[b:b + 1], lo[b], hi[b], params.jp_kp, params.jp_delta)
            jc[b], at_goal[b] = o[0], f[0]
        ext = np.zeros((4, B, n))
        ext[0] = jc
        ref = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], null_control=ctrl, ext_cmd=ext, states=states, q_lo=lo, q_hi=hi)
        ref["status"] = ref["status"] | np.where(at_goal != 0, abi.ST_JOINT_AT_GOAL, 0).astype(np.int32)
        _close(got, ref, tol, OUTS)
        assert np.abs(got["q_ref_out"] - kept).max() < (1e-12 if dt == np.float64 else 1e-6)
        stops += int((ref["status"] & abi.ST_LIMIT_STOP != 0).sum())
        clamps += int((kept != ref_j).sum())
        # static limits give a different answer: the per-cycle arrays are really used
        stat = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], null_control=ctrl, want=("qdist",))
        assert np.abs(stat["qdist"] - ref["qdist"]).max() > 1e-2
        ref_j = got["q_ref_out"].astype(np.float64)  # the controller keeps the clamped reference
        q = np.clip(q + 0.02 * ref["qdot_out"], chain.q_lo, chain.q_hi).astype(dt).astype(np.float64)
    assert 0 < stops < 3 * B and clamps > 0
    with pytest.raises(env.engine.VfikError):
        eng._chk(_only_lo(eng, q, lo))
    eng.close()


def _only_lo(eng, q, lo):
    """q_lo without q_hi is refused by the library (VFIK_E_ARG) and nothing runs."""
    import ctypes as C
    from vfclik_amd.engine import IO
    qa = np.ascontiguousarray(q, dtype=eng.io_dtype)
    la = np.ascontiguousarray(lo, dtype=eng.io_dtype)
    io = IO()
    io.q, io.q_lo = qa.ctypes.data, la.ctypes.data
    return eng.lib.vfik_step_host(eng.h, C.byref(io))


def test_rollout_clamps_to_the_arm_limits(env):
    """vfik_rollout with clamp: q stays inside the per-arm window handed in with the launch."""
    chain = env.robots.lwr()
    B = 128
    w = env.synth.make_workload(chain, B, 2, seed=8, io_dtype=np.float64)
    params = env.abi.default_params()
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=2, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    lo = w["q"] - 0.01
    hi = w["q"] + 0.02
    out = eng.rollout_host(w["q"], 200, 5e-3, clamp=True, q_lo=lo, q_hi=hi)
    assert np.all(out["q"] >= lo - 1e-15) and np.all(out["q"] <= hi + 1e-15)
    assert (np.abs(out["q"] - lo) < 1e-15).any() or (np.abs(out["q"] - hi) < 1e-15).any()  # some joints ran into their window
    free = eng.rollout_host(w["q"], 200, 5e-3, clamp=True)
    assert np.abs(free["q"] - w["q"]).max() > 0.05
    eng.close()


def test_nan_reference_row_leaves_channel_2_to_the_external_command(env):
    """Arms without a joint controller (NaN row in q_ref) mix the external /bridge/jointcmd on channel 2; the others
    mix kp * (clamp(ref) - q) and ignore it (joint_p_controller:78)."""
    abi = env.abi
    chain = env.robots.lwr()
    B, n = 96, 7
    w = env.synth.make_workload(chain, B, 1, seed=2, io_dtype=np.float64)
    params = abi.default_params(flags=abi.F_MIXER, mix_w=[1, 0, 0.7, 0, 0, 0])
    rng = np.random.default_rng(4)
    extc = rng.normal(0, 1, (B, n))
    ref_j = rng.uniform(chain.q_lo, chain.q_hi, (B, n))
    no_ctl = np.arange(B) % 4 == 0
    ref_in = ref_j.copy()
    ref_in[no_ctl] = np.nan
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=2, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    eng.set_ext_cmd(2, extc)
    got = eng.step_host(w["q"], q_ref=ref_in, want=("qdot_out", "status"))
    jc, flag = env.oc.joint_p(ref_j, w["q"], chain.q_lo, chain.q_hi, params.jp_kp, params.jp_delta)
    ext = np.zeros((4, B, n))
    ext[0] = np.where(no_ctl[:, None], extc, jc)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], ext_cmd=ext, want=("qdot_out", "status"))
    assert np.abs(got["qdot_out"] - ref["qdot_out"]).max() < TOL64
    at = (got["status"] & abi.ST_JOINT_AT_GOAL) != 0
    assert not at[no_ctl].any() and np.array_equal(at[~no_ctl], flag[~no_ctl] != 0)
    eng.close()


def test_changed_batch_mixer_weights_reach_arms_with_their_own_bridge_state(env):
    """vfik_set_params with a CHANGED mix_w after vfik_set_max_vel created per-arm bridge state: the new weights
    are written to every arm (they used to be ignored silently); vfik_set_mixer_weights(NULL) keeps per-arm max_vel."""
    abi = env.abi
    chain = env.robots.lwr()
    B = 64
    w = env.synth.make_workload(chain, B, 1, seed=6, io_dtype=np.float64)
    params = abi.default_params(flags=abi.F_MIXER | abi.F_LIMITER, max_vel=10.0)
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=2, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    mv = np.where(np.arange(B) % 2 == 0, 0.05, 10.0)
    eng.set_max_vel(mv)
    eng.set_params(mix_w=[0.25, 0, 0, 0, 0, 0])
    got = eng.step_host(w["q"], want=("qdot_out", "qdot_vf"))
    exp = 0.25 * got["qdot_vf"]
    lead = np.abs(exp).max(axis=1)
    exp = exp * np.minimum(1.0, mv / np.maximum(lead, 1e-300))[:, None]
    assert np.abs(got["qdot_out"] - exp).max() < 1e-12
    eng.set_mixer_weights(np.tile([1.0, 0, 0, 0, 0, 0], (B, 1)))
    eng.set_mixer_weights(None)  # back to the batch's 0.25; the per-arm limiter speeds stay
    again = eng.step_host(w["q"], want=("qdot_out",))
    assert np.abs(again["qdot_out"] - exp).max() < 1e-12
    eng.close()


@pytest.mark.parametrize("robot", ["lwr", "lwr_dual14"])
def test_rollout_leaves_gated_arms_alone(env, robot):
    """vfik_rollout with the fresh-q gate: gated arms are not integrated -- their q_out / qdot_out rows keep what the
    caller put there -- while the others equal a rollout without the gate (in-kernel loop for 7 joints, stepped
    launches for 14)."""
    chain = env.robots.by_name(robot)
    B, K = 96, 25
    w = env.synth.make_workload(chain, B, 2, seed=17, io_dtype=np.float64)
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=2, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    full = eng.rollout_host(w["q"], K, 2e-3, want=("qdot_out", "status"))
    eng.reset_state()
    active = np.arange(B) % 3 != 0
    got = eng.rollout_host(w["q"], K, 2e-3, want=("qdot_out", "status"), active=active)
    # (the gated launch is another kernel variant than the lean one: same arithmetic, another schedule -- not bit-equal)
    assert np.abs(got["q"][active] - full["q"][active]).max() < 1e-12
    assert np.abs(got["qdot_out"][active] - full["qdot_out"][active]).max() < 1e-10
    # a silent arm keeps its joint angles (the reference's arm keeps its state): q comes back as it went in, so that the result
    # can be fed into the next rollout; the other outputs keep the caller's rows (zeros here, `into`'s rows below)
    assert np.array_equal(got["q"][~active], w["q"][~active]) and np.all(got["qdot_out"][~active] == 0.0)
    assert np.abs(full["q"] - w["q"]).max() > 1e-3
    # closed loop over two gated rollouts: the gated arms never move, the others advance; `into` keeps their last command
    nxt = eng.rollout_host(got["q"], K, 2e-3, want=("qdot_out", "status"), active=active, into={"qdot_out": full["qdot_out"].copy()})
    assert np.array_equal(nxt["q"][~active], w["q"][~active])
    assert np.array_equal(nxt["qdot_out"][~active], full["qdot_out"][~active])
    assert np.abs(nxt["q"][active] - got["q"][active]).max() > 1e-5
    eng.close()
