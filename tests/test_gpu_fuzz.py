"""Randomised differential test: 36 seeded configurations (chain, batch size, dtype, feature flags, field
mix, tools, weights, speed scales, external channels) through the C-ABI against the CPU oracle.  Batch
sizes straddle the wave size (1, 63, 64, 65, ...), field lists are ragged and include the general path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BATCHES = [1, 2, 63, 64, 65, 127, 129, 500, 1000]


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, engine, robots, synth
    return dict(oc=oracle_c, abi=_abi, engine=engine, robots=robots, synth=synth)


def _random_fields(abi, chain, B, rng, dt, general):
    M = int(rng.integers(1, 9))
    # half of the configurations: one (safe distance, force) for every decay repeller, as the object feeder sends them -- the
    # uniform repeller image; the others mix two of each -- the compact image
    mixed = bool(rng.integers(0, 2))
    F = np.zeros((B, M), dtype=abi.FIELD_DTYPE)
    n = np.zeros(B, dtype=np.int32)
    order = float(rng.choice([2.0, 5.0, 20.0]))
    # a third of the repeller-only configurations: integer orders that differ by field (round 4: the order planes / MIXO variants);
    # a few of those with orders that differ by arm as well
    by_field = (not general) and rng.random() < 0.33
    by_arm = by_field and rng.random() < 0.4
    table = rng.choice([0.0, 1.0, 2.0, 3.0, 5.0, 20.0, 31.0], size=M)
    for b in range(B):
        cnt = int(rng.integers(0, M + 1))
        for k in range(cnt):
            f = F[b, k]
            f["id"] = int(rng.integers(1, 60))
            t = 1 if k == 0 and rng.random() < 0.85 else (int(rng.choice([1, 2, 4, 5, 0])) if general else 2)
            f["type"] = t
            if t == 1:
                f["force"] = float(rng.uniform(0.5, 2.0))
                f["p"][:16] = chain.fk(rng.uniform(0.8 * chain.q_lo, 0.8 * chain.q_hi))[0].reshape(16)
                f["p"][16] = rng.uniform(0.02, 0.2)
            elif t == 2:
                f["force"] = float(rng.choice([-10.0, -4.0])) if mixed else -10.0
                f["p"][:6] = [*rng.uniform(-0.8, 0.8, 2), rng.uniform(0, 1.2), rng.uniform(0.03, 0.1),
                              float(rng.choice([0.001, 0.004])) if mixed else 0.001,
                              (float(rng.integers(0, 12)) if by_arm and rng.random() < 0.2 else float(table[k]) if by_field else order)
                              if not general else float(rng.choice([2.0, 5.0, 3.5]))]
            elif t == 4:
                f["force"] = -50.0
                f["p"][:8] = [*rng.uniform(-0.5, 0.5, 2), -0.5, *(rng.normal(size=2) * 0.1), 1.0, 0.05, 5.0]
            elif t == 5:
                f["force"] = 30.0
                f["p"][:10] = [*rng.uniform(-0.6, 0.6, 3), *rng.normal(size=3), 0.15, 10.0, 0.15, 2.0]
        n[b] = cnt
    F["p"] = F["p"].astype(dt).astype(np.float64)  # inputs are rounded to the I/O type before both sides see them
    F["force"] = F["force"].astype(dt).astype(np.float64)
    return F, n


@pytest.mark.parametrize("seed", range(int(os.environ.get("VFIK_FUZZ_SEEDS", "36"))))  # more seeds: VFIK_FUZZ_SEEDS=400
def test_random_configuration(env, seed):
    abi = env["abi"]
    rng = np.random.default_rng(1000 + seed)
    robot = ["lwr", "powercube6", "lwr_dual14", "lwr"][seed % 4]
    chain = env["robots"].by_name(robot)
    n = chain.n
    B = BATCHES[seed % len(BATCHES)]
    dt = np.float32 if seed % 3 == 2 else np.float64
    flags = int(rng.choice([0, abi.F_MIXER, abi.F_NULLSPACE | abi.F_MIXER, abi.F_NULLSPACE | abi.F_MIXER | abi.F_LIMITER,
                            abi.F_NULLSPACE | abi.F_JOINT_LIMIT_TASK | abi.F_MIXER, abi.F_LIMITER]))
    general = bool(seed % 2)
    F, nf = _random_fields(abi, chain, B, rng, dt, general)
    q = rng.uniform(0.9 * chain.q_lo, 0.9 * chain.q_hi, (B, n)).astype(dt).astype(np.float64)
    kw = dict(flags=flags, speed_scale=float(rng.uniform(0.05, 1.0)), max_vel=float(rng.uniform(0.2, 2.0)),
              mix_w=list(rng.uniform(0, 1, 6).round(3)))
    unit_w = rng.random() < 0.5
    if not unit_w:
        kw["wy"] = list(rng.uniform(0.1, 1.0, 6))
        kw["wq"] = list(rng.uniform(0.1, 1.0, n)) + [1.0] * (16 - n)
    params = abi.default_params(**kw)
    tool = None
    if rng.random() < 0.5:
        tool = np.eye(4)
        c, s = np.cos(0.4), np.sin(0.4)
        tool[:3, :3] = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
        tool[:3, 3] = rng.uniform(-0.2, 0.2, 3)
        tool = tool.reshape(16)
    ctrl = None
    if flags & abi.F_NULLSPACE and n - 6 == 1:
        ctrl = rng.uniform(-1, 1, (B, 4)).astype(dt).astype(np.float64)
    ext = None
    if flags & abi.F_MIXER and rng.random() < 0.5:
        ext = rng.uniform(-1, 1, (4, B, n)).astype(dt).astype(np.float64)

    eng = env["engine"].Engine(chain, B, io_dtype=dt, max_slots=3 * F.shape[1], params=params)
    eng.set_fields(F, nf)
    if tool is not None:
        eng.set_tool(tool)
    if ext is not None:
        for ch in range(4):
            eng.set_ext_cmd(2 + ch, ext[ch])
    want = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status")
    got = eng.step_host(q, null_control=ctrl, want=want)
    ref = env["oc"].cycle_batch(chain, params, q, F, nf, tool=tool, null_control=ctrl, ext_cmd=ext)
    tol = 1e-9 if dt == np.float64 else 2e-5
    for k in want:
        if k == "status":
            assert np.array_equal(got[k], ref[k]), (seed, np.nonzero(got[k] != ref[k])[0][:5])
        else:
            assert np.isfinite(got[k]).all(), (seed, k)
            err = np.abs(got[k].astype(np.float64) - ref[k]).max()
            assert err < tol, (seed, robot, B, k, err)
    # the same launch asking for qdot_out alone (the LEAN kernel variant where the configuration allows it)
    eng.reset_state()
    alone = eng.step_host(q, null_control=ctrl, want=("qdot_out",))
    assert np.abs(alone["qdot_out"].astype(np.float64) - ref["qdot_out"]).max() < tol, (seed, "qdot_out alone")
    if seed % 3 == 0:
        # ABI 3 on this configuration: a fresh-q gate and this cycle's own joint limits per arm (vf:312-313, nullspace:167)
        eng.reset_state()
        active = rng.random(B) < 0.7
        half = 0.5 * (chain.q_hi - chain.q_lo) * rng.uniform(0.5, 1.0, (B, n))
        mid = 0.5 * (chain.q_hi + chain.q_lo) + rng.uniform(-0.1, 0.1, (B, n))
        lo, hi = (mid - half).astype(dt).astype(np.float64), (mid + half).astype(dt).astype(np.float64)
        into_g = {k: (np.full(B, -7, dtype=np.int32) if k == "status" else np.full(got[k].shape, 3.25, dtype=dt)) for k in want}
        into_r = {k: (np.full(B, -7, dtype=np.int32) if k == "status" else np.full(got[k].shape, 3.25)) for k in want}
        g2 = eng.step_host(q, null_control=ctrl, want=want, active=active, q_lo=lo, q_hi=hi, into=into_g)
        r2 = env["oc"].cycle_batch(chain, params, q, F, nf, tool=tool, null_control=ctrl, ext_cmd=ext, active=active, q_lo=lo, q_hi=hi,
                                   into=into_r)
        for k in want:
            if k == "status":
                assert np.array_equal(g2[k], r2[k]), (seed, "gated status")
            else:
                assert np.abs(g2[k].astype(np.float64) - r2[k]).max() < tol, (seed, "gated", k)
        assert np.all(g2["qdot_out"][~active] == 3.25)
    eng.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_session(env, seed):
    """A control session instead of a single cycle: 12 cycles on one handle with a random fresh-q gate every cycle, /control
    messages, field sets of random arm ranges replaced between cycles (switching the batch between the straight-line and the
    general field path and back), a speed-scale change -- against the oracle carrying its own per-arm state."""
    abi = env["abi"]
    rng = np.random.default_rng(5000 + seed)
    robot = ["lwr", "lwr_dual14", "powercube6"][seed % 3]
    chain = env["robots"].by_name(robot)
    n = chain.n
    B = [65, 200, 700][seed % 3]
    dt = np.float32 if seed % 2 else np.float64
    tol = 1e-9 if dt == np.float64 else 2e-5
    flags = [abi.F_NULLSPACE | abi.F_MIXER, abi.F_NULLSPACE | abi.F_JOINT_LIMIT_TASK | abi.F_MIXER, 0][seed % 3]
    params = abi.default_params(flags=flags)
    F, nf = _random_fields(abi, chain, B, rng, dt, general=False)
    M = F.shape[1]
    eng = env["engine"].Engine(chain, B, io_dtype=dt, max_slots=3 * M, params=params)
    eng.set_fields(F, nf)
    states = env["oc"].new_states(B, n)
    q = rng.uniform(0.8 * chain.q_lo, 0.8 * chain.q_hi, (B, n)).astype(dt).astype(np.float64)
    want = ("qdot_vf", "qdot_null", "qdot_out", "status")
    got = ref = None
    for t in range(12):
        if t in (3, 7, 9):  # new field sets for a random range of arms: general types at t = 3, repellers again at 7, 9
            lo = int(rng.integers(0, B - 1))
            hi = int(rng.integers(lo + 1, B + 1))
            Fn, nn = _random_fields(abi, chain, hi - lo, rng, dt, general=(t == 3))
            Fw = np.zeros((hi - lo, M), dtype=abi.FIELD_DTYPE)
            k = min(M, Fn.shape[1])
            Fw[:, :k] = Fn[:, :k]
            nw = np.minimum(nn, k).astype(np.int32)
            if t == 9:  # everything back to one decay order: the straight-line path again
                lo, hi = 0, B
                Fw, nw = np.zeros((B, M), dtype=abi.FIELD_DTYPE), np.zeros(B, dtype=np.int32)
                Fa, na = _random_fields(abi, chain, B, rng, dt, general=False)
                k = min(M, Fa.shape[1])
                Fw[:, :k] = Fa[:, :k]
                nw = np.minimum(na, k).astype(np.int32)
            eng.set_fields(Fw, nw, first_arm=lo)
            F[lo:hi], nf[lo:hi] = Fw, nw
        if t == 5:
            params.speed_scale = 0.3
            eng.set_params(speed_scale=0.3)
        active = rng.random(B) < 0.75 if t else np.ones(B, dtype=bool)
        ctrl = rng.uniform(-1, 1, (B, 4)).astype(dt).astype(np.float64) if (flags & abi.F_NULLSPACE and n == 7) else None
        got = eng.step_host(q, null_control=ctrl, want=want, active=active, into=got)
        ref = env["oc"].cycle_batch(chain, params, q, F, nf, null_control=ctrl, states=states, active=active, into=ref, want=want)
        ref.pop("states")
        for k in want:
            if k == "status":
                assert np.array_equal(got[k], ref[k]), (seed, t, "status")
            else:
                assert np.abs(got[k].astype(np.float64) - ref[k]).max() < tol, (seed, t, k)
        q = np.clip(q + 0.01 * ref["qdot_out"] + rng.normal(0, 0.01, q.shape), 0.95 * chain.q_lo, 0.95 * chain.q_hi).astype(dt).astype(np.float64)
    eng.close()
