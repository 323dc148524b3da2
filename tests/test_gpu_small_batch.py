"""The eight-lanes-per-arm kernel for small lean batches (cycle_sub8_kernel; BASELINE north_star's "wavefront per arm"
mapping) against the oracle, at BASELINE's small configurations: C1 (one arm, goal only) and C2
(4 096 arms, goal + 4 obstacles, float64), plus ragged batch sizes, more than 8 repellers, float32 I/O, 6 joints."""
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


@pytest.mark.parametrize("robot,B,nobs,dt,tol", [
    ("lwr", 1, 0, np.float64, 1e-9),        # C1
    ("lwr", 4096, 4, np.float64, 1e-9),     # C2
    ("lwr", 13, 3, np.float64, 1e-9),       # ragged: two waves, the second with 5 of 8 groups
    ("lwr", 1000, 8, np.float32, 1e-6),
    ("lwr", 333, 19, np.float32, 1e-6),     # three rounds of repellers per lane
    ("powercube6", 777, 5, np.float64, 1e-9),
])
def test_eight_lanes_per_arm_matches_the_oracle(env, robot, B, nobs, dt, tol):
    chain = env.robots.by_name(robot)
    w = env.synth.make_workload(chain, B, nobs, seed=B + nobs, io_dtype=dt)
    if nobs > 1:
        w["nfields"] = (1 + (np.arange(B) * 5) % (nobs + 1)).astype(np.int32)  # ragged field counts
        w["nfields"][0] = nobs + 1
    params = env.abi.default_params()
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=max(1, nobs), params=params)
    eng.set_fields(w["fields"], w["nfields"])
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out", "status"))
    eng.set_small_batch_kernel(0)
    lane = eng.step_host(w["q"], want=("qdot_out", "status"))
    assert eng.small_batch_launches == 0
    eng.set_small_batch_kernel(16384)
    sub8 = eng.step_host(w["q"], want=("qdot_out", "status"))
    assert eng.small_batch_launches == 1  # the launch really took the other kernel
    for got in (lane, sub8):
        assert np.abs(got["qdot_out"] - ref["qdot_out"]).max() < tol
        assert np.array_equal(got["status"], ref["status"])
    # what the kernel does not serve (the field twist /vector_out, a gate, per-arm limits ...) stays with one lane per arm
    eng.step_host(w["q"], want=("qdot_out", "v6"))
    assert eng.small_batch_launches == 1
    eng.step_host(w["q"], want=("qdot_out",), active=np.ones(B, dtype=np.int32))
    assert eng.small_batch_launches == 1
    eng.set_small_batch_kernel(0)
    eng.step_host(w["q"], want=("qdot_out",))
    assert eng.small_batch_launches == 1
    eng.close()


FULL = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "qdist", "status")


@pytest.mark.parametrize("robot,B,nobs,dt,tol,flags", [
    ("lwr", 1, 0, np.float64, 1e-9, 5),          # C1 with vfclik's default process set: nullspace + mixer (vfclik:95-97)
    ("lwr", 2, 2, np.float64, 1e-9, 5),          # a dual-arm robot
    ("lwr", 4096, 4, np.float64, 1e-9, 5),       # C2's batch
    ("lwr", 77, 3, np.float64, 1e-9, 7),         # + joint-limit task
    ("lwr", 1000, 8, np.float32, 1e-6, 5),
    ("lwr", 333, 11, np.float32, 1e-6, 15),      # + limiter
    ("powercube6", 90, 2, np.float64, 1e-9, 5),  # 6 joints: no nullspace direction (status NULL_AMBIGUOUS never, nullity 0)
    ("lwr", 40, 2, np.float64, 1e-9, 0),         # no module, all outputs
])
def test_default_process_set_on_eight_lanes_per_arm(env, robot, B, nobs, dt, tol, flags):
    """What vfclik runs per arm -- vf + nullspace + debug_jointlimits + the bridge's mixer, everything they publish every
    cycle -- through cycle_sub8_kernel<NS>: every output against the oracle over three cycles (the sign memory advances),
    /control active, and against the one-lane-per-arm kernel."""
    chain = env.robots.by_name(robot)
    w = env.synth.make_workload(chain, B, nobs, seed=3 * B + nobs, io_dtype=dt)
    params = env.abi.default_params(flags=flags, max_vel=0.3)
    rng = np.random.default_rng(B)
    dq = rng.normal(size=w["q"].shape) * 0.04
    engs = {}
    for name, mb in (("sub8", 1 << 20), ("lane", 0)):
        e = env.engine.Engine(chain, B, io_dtype=dt, max_slots=max(1, nobs), params=params)
        e.set_fields(w["fields"], w["nfields"])
        e.set_small_batch_kernel(mb)
        engs[name] = e
    states = env.oc.new_states(B, chain.n) if flags & 1 else None
    q = w["q"].copy()
    for t in range(3):
        ctrl = rng.uniform(-1, 1, (B, 4)) if flags & 1 else None
        qr = q.astype(dt).astype(np.float64)
        ref = env.oc.cycle_batch(chain, params, qr, w["fields"], w["nfields"], null_control=ctrl, states=states)
        got = {k: e.step_host(qr, null_control=ctrl, want=FULL) for k, e in engs.items()}
        for name in ("sub8", "lane"):
            for k in FULL:
                if k == "status":
                    assert np.array_equal(got[name][k], ref[k]), (name, t)
                else:
                    err = np.abs(got[name][k].astype(np.float64) - ref[k]).max()
                    assert err < tol, (name, k, t, err)
        q = np.clip(q + dq, 0.9 * chain.q_lo, 0.9 * chain.q_hi)
    assert engs["sub8"].small_batch_launches == 3 and engs["lane"].small_batch_launches == 0
    for e in engs.values():
        e.close()


def test_an_arm_gives_the_same_result_wherever_it_sits_in_the_batch(env):
    """The nullspace warm start decides per arm: permuting the arms of a batch -- other wave mates, some of them with a
    fresh state, some mid-trajectory -- permutes the results and changes nothing else, bit for bit, in both kernels."""
    chain = env.robots.lwr()
    B = 512
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    w = env.synth.make_workload(chain, B, 3, seed=8, io_dtype=np.float64)
    rng = np.random.default_rng(8)
    perm = rng.permutation(B)
    ctrl = rng.uniform(-1, 1, (B, 4))
    move = 0.3 * rng.normal(size=(1, 7))
    for mb in (0, 1 << 20):
        res = []
        for order in (np.arange(B), perm):
            e = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=4, params=params)
            e.set_small_batch_kernel(mb)
            e.set_fields(w["fields"][order], w["nfields"][order])
            q = w["q"][order].copy()
            # two cycles for HALF of the arms only (gate): afterwards the batch mixes arms with and without a stored vector
            half = (order % 2 == 0)
            e.step_host(q, null_control=ctrl[order], want=("qdot_out",), active=half)
            q2 = q + move                               # a real move: some arms need the second projection
            out = e.step_host(q2, null_control=ctrl[order], want=("qdot_null", "qdot_out", "status"))
            back = np.empty_like(order)
            back[order] = np.arange(B)
            res.append({k: v[back] for k, v in out.items()})
            e.close()
        for k in ("qdot_null", "qdot_out", "status"):
            assert np.array_equal(res[0][k], res[1][k]), (mb, k)
