"""The DH-pattern specialisation of the lean float32-I/O kernels (vfik_dh_pattern; vfclik_amd/csrc/vfik_kernel.h: DhPattern): chains
that match the built pattern of their joint count take variants in which links with a = 0, alpha = +-pi/2 / 0 and d = 0 cost no
arithmetic; every other chain the general DH form.  Both against the oracle, and against each other."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status")


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, chain, engine, robots, synth
    return dict(oc=oracle_c, abi=_abi, engine=engine, robots=robots, synth=synth, chain=chain)


def _step(env, chain, params, w, want, nrep, off=False):
    if off:
        os.environ["VFIK_DH_PATTERN"] = "0"
    try:
        eng = env["engine"].Engine(chain, w["q"].shape[0], io_dtype=np.float32, max_slots=max(nrep, 1), device=0, params=params)
    finally:
        os.environ.pop("VFIK_DH_PATTERN", None)
    eng.set_small_batch_kernel(0)
    eng.set_fields(w["fields"], w["nfields"])
    pat = eng.dh_pattern
    out = eng.step_host(w["q"], want=want)
    eng.close()
    return out, pat


@pytest.mark.parametrize("robot,flags,nrep", [("lwr", 0, 8), ("lwr", 5, 8), ("lwr", 7, 3), ("lwr_dual14", 7, 16), ("lwr_dual14", 0, 5), ("powercube6", 5, 4)])
def test_pattern_variants_equal_the_general_form_and_the_oracle(env, robot, flags, nrep):
    chain = env["robots"].by_name(robot)
    B = 64 * 5 + 7
    w = env["synth"].make_workload(chain, B, nrep, seed=61, io_dtype=np.float32)
    params = env["abi"].default_params(flags=flags)
    ref = env["oc"].cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=ALL)
    for want in (("qdot_out", "status"), ALL):
        got, pat = _step(env, chain, params, w, want, nrep)
        gen, pat0 = _step(env, chain, params, w, want, nrep, off=True)
        assert pat == 1 and pat0 == 0
        for k in want:
            if k == "status":
                assert np.array_equal(got[k], ref[k]) and np.array_equal(gen[k], ref[k])
            else:
                assert np.abs(got[k].astype(np.float64) - ref[k]).max() < 1e-6, k
                assert np.abs(got[k].astype(np.float64) - gen[k]).max() < 3e-7, k    # (the float32 store's last bit at most)


def test_other_scenes_on_a_pattern_chain(env):
    """Differing orders (MIXO variants), the goalAndNormal scene (FUN) and a batch beyond one wave per SIMD (two-waves build)."""
    chain = env["robots"].lwr()
    params = env["abi"].default_params(flags=5)
    B = 64 * 3 + 5
    w = env["synth"].make_workload(chain, B, 8, seed=63, io_dtype=np.float32)
    w["fields"]["p"][:, 1:9, 5] = [5, 20, 20, 5, 5, 20, 20, 20]
    ref = env["oc"].cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out",))
    got, pat = _step(env, chain, params, w, ("qdot_out",), 8)
    assert pat == 1 and np.abs(got["qdot_out"] - ref["qdot_out"]).max() < 1e-6
    w = env["synth"].make_workload(chain, B, 5, seed=64, io_dtype=np.float32, max_fields=7)
    F = w["fields"]
    F["id"][:, 6], F["type"][:, 6], F["force"][:, 6] = 2, 5, 30.0
    F["p"][:, 6, 0:3] = F["p"][:, 0, [3, 7, 11]]
    F["p"][:, 6, 3:6] = F["p"][:, 0, [2, 6, 10]]
    F["p"][:, 6, 6:10] = [0.15, 10.0, 0.15, 2.0]
    w["nfields"][:] = 7
    ref = env["oc"].cycle_batch(chain, params, w["q"], F, w["nfields"], want=ALL)
    got, _ = _step(env, chain, params, w, ALL, 8)
    for k in ALL[:-1]:
        assert np.abs(got[k].astype(np.float64) - ref[k]).max() < 1e-6, k
    Bbig = 65536 + 64 * 40 + 3
    w = env["synth"].make_workload(chain, Bbig, 4, seed=65, io_dtype=np.float32)
    for flags in (0, 5):
        p = env["abi"].default_params(flags=flags)
        ref = env["oc"].cycle_batch(chain, p, w["q"], w["fields"], w["nfields"], want=("qdot_out",))
        got, _ = _step(env, chain, p, w, ("qdot_out",), 4)
        assert np.abs(got["qdot_out"] - ref["qdot_out"]).max() < 1e-6


def test_chains_that_do_not_match_run_the_general_form(env):
    """One link off the pattern (an a-offset, another twist, a z-offset where the pattern has none): vfik_dh_pattern 0, the oracle's
    results; a chain with MORE zeros than the pattern asks for still qualifies."""
    from vfclik_amd import robots as R
    lim = np.array([170, 120, 170, 120, 170, 120, 170], dtype=float) * math.pi / 180
    base = list(R._LWR_DH)
    cases = []
    dh = list(base); dh[2] = (0.05, -math.pi / 2, 0.4, 0.0); cases.append((dh, 0))        # a != 0
    dh = list(base); dh[3] = (0.0, 1.0, 0.0, 0.0); cases.append((dh, 0))                    # alpha = 1 rad
    dh = list(base); dh[1] = (0.0, -math.pi / 2, 0.02, 0.0); cases.append((dh, 0))          # d != 0 where the pattern has d = 0
    dh = list(base); dh[0] = (0.0, math.pi / 2, 0.0, 0.0); cases.append((dh, 1))            # one more zero: still the pattern
    dh = list(base); dh[6] = (0.0, math.pi / 2, 0.078, 0.0); cases.append((dh, 0))          # the last link twisted
    params = env["abi"].default_params(flags=5)
    for dh, expect in cases:
        chain = env["chain"].Chain.from_dh(dh, -lim, lim, name="lwr_variant")
        w = env["synth"].make_workload(chain, 200, 4, seed=67, io_dtype=np.float32)
        ref = env["oc"].cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=ALL)
        got, pat = _step(env, chain, params, w, ALL, 4)
        assert pat == expect, (dh, pat)
        for k in ALL[:-1]:
            assert np.abs(got[k].astype(np.float64) - ref[k]).max() < 1e-6, (k, expect)
        assert np.array_equal(got["status"], ref["status"])
