"""Observers: goal distance (monitor_distance) from the control-cycle kernel and the
tracking-error estimator of scripts/vf (vf:349-428) as its own kernel, against the NumPy restatements."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_goal_distance_and_tracking_error():
    import __graft_entry__ as g
    g.build()
    from oracle import vfik_numpy as vn
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    B, K, dt = 256, 14, 1.0 / 150.0
    w = synth.make_workload(chain, B, 2, seed=41, io_dtype=np.float64)
    params = _abi.default_params(flags=_abi.F_MIXER, mix_w=[1, 0, 0, 0, 0, 0])
    eng = engine.Engine(chain, B, io_dtype=np.float64, max_slots=4, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    d_pose, d_v6, d_out = eng.dev_alloc(B * 16 * 8), eng.dev_alloc(B * 6 * 8), eng.dev_alloc(B * 8 * 8)
    est = [vn.TrackingError() for _ in range(B)]
    q = w["q"].copy()
    rng = np.random.default_rng(2)
    seen_values = 0
    for t in range(K):
        out = eng.step_host(q, want=("qdot_out", "pose", "v6", "goal_dist"))
        # goal distance: every arm, every cycle
        for b in range(0, B, 17):
            D, ang = vn.goal_distance(out["pose"][b], w["fields"]["p"][b, 0, :16])
            assert abs(out["goal_dist"][b, 0] - D) < 1e-12 and abs(out["goal_dist"][b, 1] - ang) < 1e-9
        eng.h2d(d_pose, out["pose"])
        eng.h2d(d_v6, out["v6"])
        eng.track_error(d_pose, d_v6, d_out)
        got = np.zeros((B, 8))
        eng.d2h(got, d_out)
        for b in range(B):
            ref = est[b].update(vn.listToKdlFrame(out["pose"][b]), out["v6"][b, :3], out["v6"][b, 3:])
            if ref is None:
                assert np.all(got[b] == 0.0)
            else:
                seen_values += 1
                assert np.abs(got[b] - ref).max() < 1e-9, (t, b, got[b], ref)
        # an imperfect robot: it follows the command only partly, some arms not at all
        follow = np.where(np.arange(B) % 5 == 0, 0.0, rng.uniform(0.5, 1.0, B))[:, None]
        q = q + dt * follow * out["qdot_out"]
    assert seen_values == B * (K - 5)
    assert 0 < got[:, 7].sum() < B  # some arms track, the frozen ones do not
    eng.track_reset()
    eng.track_error(d_pose, d_v6, d_out)
    eng.d2h(got, d_out)
    assert np.all(got == 0.0)
    for p in (d_pose, d_v6, d_out):
        eng.dev_free(p)
    eng.close()


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-9), (np.float32, 2e-2)])
def test_object_distances_match_the_monitor(dt, tol):
    """vfik_object_distances vs the restatement of monitor_distance:148-167 (xyz distance, rotation angle in
    degrees) for random object frames, plus the corner cases: same orientation (0), half turn (180)."""
    import __graft_entry__ as g
    g.build()
    from oracle import vfik_numpy as vn
    from vfclik_amd import engine, robots
    chain = robots.lwr()
    B, O = 130, 5
    rng = np.random.default_rng(12)
    pose = chain.fk(rng.uniform(chain.q_lo, chain.q_hi, (B, 7))).reshape(B, 16).astype(dt)
    frames = chain.fk(rng.uniform(chain.q_lo, chain.q_hi, (B * O, 7))).reshape(B, O, 16).astype(dt)
    frames[:, 0] = pose                                  # object on the tool: distance 0, angle 0
    frames[:, 1] = pose
    frames[:, 1, [0, 1, 4, 5, 8, 9]] *= -1               # R_cur * Rz(pi): a half turn, columns x and y negated
    frames[:, 1, 3] += 0.25                              # and 25 cm away in x
    eng = engine.Engine(chain, B, io_dtype=dt, max_slots=2)
    esz = np.dtype(dt).itemsize
    d_pose, d_fr, d_out = eng.dev_alloc(B * 16 * esz), eng.dev_alloc(B * O * 16 * esz), eng.dev_alloc(B * O * 2 * esz)
    eng.h2d(d_pose, pose)
    eng.h2d(d_fr, frames)
    eng.object_distances(d_pose, d_fr, O, d_out)
    got = np.zeros((B, O, 2), dtype=dt)
    eng.d2h(got, d_out)
    for b in range(B):
        ref = vn.object_distances(pose[b].astype(np.float64), {k: frames[b, k].astype(np.float64) for k in range(O)})
        for k, (oid, dxyz, dang) in enumerate(ref):
            assert oid == k
            assert abs(got[b, k, 0] - dxyz) < (1e-12 if dt == np.float64 else 1e-6)
            assert abs(got[b, k, 1] - dang) < tol, (b, k, got[b, k, 1], dang)
    assert np.abs(got[:, 0]).max() < (1e-6 if dt == np.float64 else 0.1)
    assert np.abs(got[:, 1, 0] - 0.25).max() < 1e-6 and np.abs(got[:, 1, 1] - 180.0).max() < (1e-5 if dt == np.float64 else 0.1)
    with pytest.raises(engine.VfikError):
        eng.object_distances(d_pose, d_fr, 0, d_out)
    for p in (d_pose, d_fr, d_out):
        eng.dev_free(p)
    eng.close()


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-9), (np.float32, 2e-5)])
def test_field_probe_at_arbitrary_poses(dt, tol):
    """vfik_probe_field (vf:469-503): each arm's field set -- every primitive type, ragged -- at poses that are
    NOT the arm's forward kinematics, against the oracle's field evaluation; per-arm speed scales apply."""
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    B = 200
    rng = np.random.default_rng(31)
    w = synth.make_workload(chain, B, 3, seed=31, io_dtype=dt, max_fields=6)
    F = w["fields"]
    for b in range(0, B, 3):  # add a hemisphere and a funnel to every third arm
        F[b, 4]["id"], F[b, 4]["type"], F[b, 4]["force"] = 30, 4, -50.0
        F[b, 4]["p"][:8] = [0.2, -0.1, -0.5, 0.05, -0.02, 1.0, 0.05, 5.0]
        F[b, 5]["id"], F[b, 5]["type"], F[b, 5]["force"] = 2, 5, 30.0
        F[b, 5]["p"][:10] = [0.3, 0.2, 0.5, 0.1, 0.2, -0.9, 0.15, 10.0, 0.15, 2.0]
        w["nfields"][b] = 6
    F["p"] = F["p"].astype(dt).astype(np.float64)
    params = _abi.default_params(rot_slowdown=0.2)
    eng = engine.Engine(chain, B, io_dtype=dt, max_slots=12, params=params)
    eng.set_fields(F, w["nfields"])
    speed = rng.uniform(0.05, 0.41, B)
    eng.set_speed_scale(speed)
    poses = chain.fk(rng.uniform(chain.q_lo, chain.q_hi, (B, 7))).reshape(B, 16).astype(dt)
    esz = np.dtype(dt).itemsize
    d_pose, d_v6 = eng.dev_alloc(B * 16 * esz), eng.dev_alloc(B * 6 * esz)
    eng.h2d(d_pose, poses)
    eng.probe_field(d_pose, d_v6)
    got = np.zeros((B, 6), dtype=dt)
    eng.d2h(got, d_v6)
    ref = oracle_c.probe_field(params, F, w["nfields"], poses.astype(np.float64), speed=speed)
    assert np.abs(got - ref).max() < tol
    assert np.abs(got).max() > 0.01
    eng.dev_free(d_pose); eng.dev_free(d_v6)
    eng.close()


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-9), (np.float32, 1e-3)])
def test_observers_fused_into_the_cycle_call(dt, tol):
    """ABI 4: io.track_error / io.obj_dist -- the tracking-error estimator and the distance monitor run in the SAME call
    on the cycle's own device pose / twist (no host round trip, one synchronisation), with and without pose / v6 asked
    for, under the fresh-q gate; against the NumPy restatements fed the published pose / twist."""
    import __graft_entry__ as g
    g.build()
    from oracle import vfik_numpy as vn
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    B, K, O, step = 192, 12, 3, 1.0 / 150.0
    w = synth.make_workload(chain, B, 2, seed=43, io_dtype=dt)
    params = _abi.default_params(flags=_abi.F_NULLSPACE | _abi.F_MIXER)
    eng = engine.Engine(chain, B, io_dtype=dt, max_slots=4, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    with pytest.raises(engine.VfikError):   # the monitor needs its objects first
        eng.step_host(w["q"], want=("qdot_out", "obj_dist"))
    rng = np.random.default_rng(7)
    frames = chain.fk(rng.uniform(chain.q_lo, chain.q_hi, (B * O, 7))).reshape(B, O, 16)
    frames[:, 0] = w["fields"]["p"][:, 0, :16]          # object 0 = the goal (object_feeder:215-227)
    eng.set_objects(frames)
    est = [vn.TrackingError() for _ in range(B)]
    q = w["q"].copy()
    outs = prev = None
    seen = 0
    for t in range(K):
        active = rng.uniform(size=B) > 0.2 if t % 3 == 1 else None
        ask_pose = t % 2 == 0                          # with and without the caller asking for pose / v6 itself
        want = ("qdot_out", "track_error", "obj_dist", "goal_dist") + (("pose", "v6") if ask_pose else ())
        if outs is not None:
            outs = {k: v for k, v in outs.items() if k in want}
        out = eng.step_host(q, want=want, active=active, into=outs)
        ref_full = eng.step_host(q, want=("pose", "v6"))  # the same cycle's pose / twist (stateless outputs) for the restatement
        act = np.ones(B, dtype=bool) if active is None else active
        if outs is not None and not act.all():
            # a gated arm got no new pose: its distances and its tracking error keep what the caller's arrays held (ABI 5; until then
            # obj_dist was recomputed from the arm's previous pose)
            assert np.array_equal(out["obj_dist"][~act], prev["obj_dist"][~act]) and np.array_equal(out["track_error"][~act], prev["track_error"][~act])
        outs = dict(out)
        prev = {k: v.copy() for k, v in out.items()}
        for b in range(B):
            if not act[b]:
                continue
            ref = est[b].update(vn.listToKdlFrame(ref_full["pose"][b].astype(np.float64)), ref_full["v6"][b, :3].astype(np.float64),
                                ref_full["v6"][b, 3:].astype(np.float64))
            if ref is None:
                assert np.all(out["track_error"][b] == 0.0)
            else:
                seen += 1
                assert np.abs(out["track_error"][b, :7] - ref[:7]).max() < (1e-7 if dt == np.float64 else 5e-6), (t, b, out["track_error"][b], ref)
            if b % 13 == 0:
                mon = vn.object_distances(ref_full["pose"][b].astype(np.float64), {k: frames[b, k].astype(dt).astype(np.float64) for k in range(O)})
                for k, (oid, dxyz, dang) in enumerate(mon):
                    assert abs(out["obj_dist"][b, k, 0] - dxyz) < (1e-9 if dt == np.float64 else 1e-5)
                    assert abs(out["obj_dist"][b, k, 1] - dang) < (1e-6 if dt == np.float64 else 5e-2)
                # object 0 is the goal: the monitor's entry equals the cycle kernel's own goal distance
                assert abs(out["obj_dist"][b, 0, 0] - out["goal_dist"][b, 0]) < (1e-9 if dt == np.float64 else 1e-5)
        q = q + step * np.where(act[:, None], out["qdot_out"].astype(np.float64), 0.0)
    assert seen > B * (K - 6) * 0.7
    # observers belong to single cycles
    with pytest.raises(engine.VfikError):
        eng.rollout_host(q, 3, 1e-3, want=("qdot_out", "track_error"))
    # a partial update of the objects must keep n_objects
    with pytest.raises(engine.VfikError):
        eng.set_objects(frames[:5, :2], first_arm=0)
    eng.set_objects(frames[5:9], first_arm=5)
    eng.close()
