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
