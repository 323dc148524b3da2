"""Batches beyond one wave per SIMD take the persistent launch (cycle_kernel, PERS: one wave per SIMD striding over the
64-arm chunks, the next chunk's inputs in flight into a second LDS area).  Parity against the oracle and against the
launch in rounds (VFIK_PERSISTENT=0) on ragged batch sizes: waves with one, two and three chunks, a partial last chunk."""
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


def _engine(env, chain, B, nobs, params, persistent):
    old = os.environ.get("VFIK_PERSISTENT")
    os.environ["VFIK_PERSISTENT"] = "1" if persistent else "0"
    try:
        eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=max(1, nobs), params=params)
    finally:
        if old is None:
            del os.environ["VFIK_PERSISTENT"]
        else:
            os.environ["VFIK_PERSISTENT"] = old
    return eng


@pytest.mark.parametrize("B,nobs", [(65536 + 64 * 3 + 17, 8), (131072 + 5, 8), (65536 * 2 + 64 * 700, 3), (65537, 11), (200000, 0)])
def test_persistent_launch_matches_oracle_and_rounds(env, B, nobs):
    chain = env.robots.lwr()
    params = env.abi.default_params()
    w = env.synth.make_workload(chain, B, nobs, seed=21, io_dtype=np.float32)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out", "status"))
    outs = {}
    for pers in (True, False):
        eng = _engine(env, chain, B, nobs, params, pers)
        eng.set_fields(w["fields"], w["nfields"])
        outs[pers] = eng.step_host(w["q"], want=("qdot_out", "status"))
        eng.close()
    for pers in (True, False):
        err = np.abs(outs[pers]["qdot_out"].astype(np.float64) - ref["qdot_out"])
        assert err.max() < 1e-6, (pers, float(err.max()), int(np.argmax(err.max(axis=1))))
        assert np.array_equal(outs[pers]["status"], ref["status"])
    # same arithmetic in both launches (another schedule): equal to rounding of the float32 store
    assert np.abs(outs[True]["qdot_out"] - outs[False]["qdot_out"]).max() < 1e-6


def test_persistent_launch_with_the_nullspace_module_keeps_its_state(env):
    """The default process set (nullspace + mixer) at 150 000 arms, three cycles: the sign memory of every arm advances
    through the persistent launch exactly as through the oracle's state."""
    chain = env.robots.lwr()
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    B = 150000 + 13
    w = env.synth.make_workload(chain, B, 4, seed=22, io_dtype=np.float32)
    rng = np.random.default_rng(5)
    eng = _engine(env, chain, B, 4, params, True)
    eng.set_fields(w["fields"], w["nfields"])
    import torch
    q = w["q"].copy()
    dq = rng.normal(size=q.shape) * 0.05
    states = env.oc.new_states(B, chain.n)
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    for t in range(3):
        q32 = q.astype(np.float32)
        # lean device-pointer launch (qdot_out + status only: what takes the persistent kernel)
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
