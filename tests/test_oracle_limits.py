"""The oracle's per-cycle joint limits and fresh-q gate (CPU): the C batch driver against the reference-shaped
NumPy loop built with the same limits (scripts/nullspace:167, scripts/debug_jointlimits:66-67), and gated
arms left alone (scripts/vf:312-313)."""
import numpy as np


def test_per_arm_limits_and_gate(oracle_c):
    from oracle import vfik_numpy as vn
    from vfclik_amd import _abi, robots, synth
    chain = robots.lwr()
    B, n = 12, 7
    w = synth.make_workload(chain, B, 2, seed=13, io_dtype=np.float64)
    params = _abi.default_params(flags=_abi.F_NULLSPACE | _abi.F_JOINT_LIMIT_TASK | _abi.F_MIXER)
    rng = np.random.default_rng(0)
    ctrl = rng.uniform(-2, 2, (B, 4))
    half = 0.5 * (chain.q_hi - chain.q_lo) * rng.uniform(0.4, 1.0, (B, n))
    mid = 0.5 * (chain.q_hi + chain.q_lo) + rng.uniform(-0.2, 0.2, (B, n))
    lo, hi = mid - half, mid + half
    active = np.arange(B) % 3 != 1
    into = {k: np.full((B, n), 9.0) for k in ("qdot_vf", "qdot_null", "qdot_out", "qdist")}
    into["status"] = np.full(B, -1, dtype=np.int32)
    states = oracle_c.new_states(B, n)
    got = oracle_c.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], null_control=ctrl, states=states, q_lo=lo, q_hi=hi,
                               active=active, into=into, want=("qdot_vf", "qdot_null", "qdot_out", "qdist", "status"))
    pd = _abi.params_to_dict(params)
    stopped = 0
    for b in range(B):
        if not active[b]:
            assert np.all(got["qdot_out"][b] == 9.0) and got["status"][b] == -1
            assert np.all(np.array(states[b].lastvec) == 0.0)  # state untouched
            continue
        arm = vn.ArmCycle(chain.B, chain.jtype, lo[b], hi[b], pd)
        arm.set_fields({int(f["id"]): [float(f["force"]), int(f["type"]), f["p"][:_abi.FIELD_NPARAMS[int(f["type"])]].tolist()]
                        for f in w["fields"][b][: w["nfields"][b]]})
        r = arm.cycle(w["q"][b].tolist(), null_control=ctrl[b])
        for k in ("qdot_vf", "qdot_null", "qdot_out", "qdist"):
            assert np.abs(got[k][b] - r[k]).max() < 1e-9, (b, k)
        assert got["status"][b] == r["status"]
        stopped += bool(r["status"] & _abi.ST_LIMIT_STOP)
    assert 0 < stopped < B
