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
    # anything the lean launch does not cover stays with the lane-per-arm kernel
    eng.step_host(w["q"], want=("qdot_out", "pose"))
    assert eng.small_batch_launches == 1
    eng.set_small_batch_kernel(0)
    eng.step_host(w["q"], want=("qdot_out",))
    assert eng.small_batch_launches == 1
    eng.close()
