"""Batches beyond one wave per SIMD: the two-waves-per-SIMD build of the lean launch (cycle_kernel, WAVES 2: the default there) and
the persistent launch (cycle_kernel, PERS, VFIK_PERSISTENT=1: one wave per SIMD striding over the 64-arm chunks, the next chunk's
inputs in flight into a second LDS area).  Parity against the oracle and against the launch in rounds of one wave per SIMD
(VFIK_TWO_WAVES=0) on ragged batch sizes: waves with one, two and three chunks, a partial last chunk."""
import os

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
    e.n_simd = 4 * 256
    return e


MODES = ("rounds", "two", "pers")


def _engine(env, chain, B, nobs, params, mode):
    """A handle whose big lean launches go in rounds of one wave per SIMD, as two waves per SIMD, or persistent."""
    want = {"VFIK_PERSISTENT": "1" if mode == "pers" else "0", "VFIK_TWO_WAVES": "1" if mode == "two" else "0"}
    old = {k: os.environ.get(k) for k in want}
    os.environ.update(want)
    try:
        eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=max(1, nobs), params=params)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return eng


@pytest.mark.parametrize("B,nobs", [(65536 + 64 * 3 + 17, 8), (131072 + 5, 8), (65536 * 2 + 64 * 700, 3), (65537, 11), (200000, 0)])
def test_persistent_launch_matches_oracle_and_rounds(env, B, nobs):
    chain = env.robots.lwr()
    params = env.abi.default_params()
    w = env.synth.make_workload(chain, B, nobs, seed=21, io_dtype=np.float32)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out", "status"))
    outs = {}
    for mode in MODES:
        eng = _engine(env, chain, B, nobs, params, mode)
        eng.set_fields(w["fields"], w["nfields"])
        outs[mode] = eng.step_host(w["q"], want=("qdot_out", "status"))
        eng.close()
    for mode in MODES:
        err = np.abs(outs[mode]["qdot_out"].astype(np.float64) - ref["qdot_out"])
        assert err.max() < 1e-6, (mode, float(err.max()), int(np.argmax(err.max(axis=1))))
        assert np.array_equal(outs[mode]["status"], ref["status"])
    # same arithmetic in every launch (another schedule): equal to rounding of the float32 store
    assert np.abs(outs["pers"]["qdot_out"] - outs["rounds"]["qdot_out"]).max() < 1e-6
    assert np.abs(outs["two"]["qdot_out"] - outs["rounds"]["qdot_out"]).max() < 1e-6


@pytest.mark.parametrize("robot,flags", [("powercube6", 0), ("powercube6", 7), ("lwr", 7)])
def test_two_waves_build_of_the_other_lean_variants(env, robot, flags):
    """The 6-joint chain and the joint-limit-task flag set through the two-waves-per-SIMD build, 70 000 arms."""
    chain = env.robots.by_name(robot)
    params = env.abi.default_params(flags=flags)
    B = 70000 + 33
    w = env.synth.make_workload(chain, B, 5, seed=23, io_dtype=np.float32)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out", "status"))
    eng = _engine(env, chain, B, 5, params, "two")
    eng.set_fields(w["fields"], w["nfields"])
    got = eng.step_host(w["q"], want=("qdot_out", "status"))
    eng.close()
    assert np.abs(got["qdot_out"].astype(np.float64) - ref["qdot_out"]).max() < 2e-6
    assert np.array_equal(got["status"], ref["status"])


@pytest.mark.parametrize("mode", ["two", "pers"])
def test_big_launch_with_the_nullspace_module_keeps_its_state(env, mode):
    """The default process set (nullspace + mixer) at 150 000 arms, three cycles: the sign memory of every arm advances
    through the two-waves / the persistent launch exactly as through the oracle's state."""
    chain = env.robots.lwr()
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    B = 150000 + 13
    w = env.synth.make_workload(chain, B, 4, seed=22, io_dtype=np.float32)
    rng = np.random.default_rng(5)
    eng = _engine(env, chain, B, 4, params, mode)
    eng.set_fields(w["fields"], w["nfields"])
    import torch
    q = w["q"].copy()
    dq = rng.normal(size=q.shape) * 0.05
    states = env.oc.new_states(B, chain.n)
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    for t in range(3):
        q32 = q.astype(np.float32)
        # lean device-pointer launch (qdot_out + status only: what takes these kernels)
        qd = torch.from_numpy(q32).cuda()
        out = torch.zeros(B, chain.n, dtype=torch.float32, device="cuda")
        st = torch.zeros(B, dtype=torch.int32, device="cuda")
        eng.step(eng.make_io(qd, qdot_out=out, status=st))
        torch.cuda.synchronize()
        ref = env.oc.cycle_batch(chain, params, q32.astype(np.float64), w["fields"], w["nfields"], states=states, want=("qdot_out", "status"))
        err = np.abs(out.cpu().numpy().astype(np.float64) - ref["qdot_out"])
        assert err.max() < 1e-6, (t, float(err.max()))
        assert np.array_equal(st.cpu().numpy(), ref["status"])
        q = np.clip(q + dq, 0.9 * chain.q_lo, 0.9 * chain.q_hi)
    eng.close()
