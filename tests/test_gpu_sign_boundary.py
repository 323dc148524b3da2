"""The nullspace sign memory at its decision boundary (scripts/nullspace:101-105): the continuity test compares the new
basis vector v with the one stored at the previous cycle, `norm(sig v - lastvec) > norm(-sig v - lastvec)`, i.e. it looks
at the sign of v . lastvec.  The kernel keeps lastvec as FLOAT32 (vfik_kernel.hip, nullspace state), the oracle as double:
this test jumps the arm between two cycles to a pose whose nullspace direction is at ~90 degrees to the previous one, so
that v . lastvec is +-1e-2 ... +-1e-9, at float64 and float32 I/O.

Expected: wherever |v . lastvec| is above what a float32 lastvec can resolve (rounding each component to 6e-8 relative
moves the dot product by < 1e-7) the kernel takes the oracle's decision and publishes the oracle's vector; below that the
sign of the dot product is rounding noise of the STATE's precision and the kernel may publish either +-v -- always a unit
nullspace vector, never anything else.  (The reference itself decides such a case on the rounding noise of its doubles.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit_null(oc, chain, q):
    J, _ = oc.jacobian(chain, q)
    _, s, vh = np.linalg.svd(J)
    return vh[-1]


def _find_quarter_turns(oc, chain, rng, want, span=2.5):
    """(q1, direction, t*) with u(q1) . u(q1 + t* d) = 0 along a straight joint-space line inside the limits."""
    found = []
    lo, hi = 0.9 * chain.q_lo, 0.9 * chain.q_hi
    while len(found) < want:
        q1 = rng.uniform(0.5 * lo, 0.5 * hi)
        d = rng.normal(size=chain.n)
        d /= np.linalg.norm(d)
        u1 = _unit_null(oc, chain, q1)
        prev, tprev, fprev = u1, 0.0, 1.0
        for t in np.linspace(0.02, span, 126):
            q = q1 + t * d
            if np.any(q < lo) or np.any(q > hi):
                break
            u = _unit_null(oc, chain, q)
            if u @ prev < 0:
                u = -u          # continuous branch along the line
            f = float(u @ u1)
            if f * fprev < 0:   # bracketed: bisect, keeping the branch continuous from the bracket's lower end
                a, b, ua = tprev, t, prev
                for _ in range(60):
                    m = 0.5 * (a + b)
                    um = _unit_null(oc, chain, q1 + m * d)
                    if um @ ua < 0:
                        um = -um
                    if (um @ u1) * fprev > 0:
                        a, ua = m, um
                    else:
                        b = m
                found.append((q1, d, 0.5 * (a + b)))
                break
            prev, tprev, fprev = u, t, f
    return found


@pytest.mark.parametrize("io_dtype,tol", [(np.float64, 1e-9), (np.float32, 1e-6)])
def test_sign_continuity_when_the_direction_jumps_a_quarter_turn(io_dtype, tol):
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c as oc
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    rng = np.random.default_rng(2026)
    cases = _find_quarter_turns(oc, chain, rng, 24)
    offsets = [s * e for e in (1e-2, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9, 0.0) for s in (1.0, -1.0)][:-1]  # dt around the root
    q1 = np.array([c[0] for c in cases for _ in offsets]).astype(io_dtype).astype(np.float64)
    q2 = np.array([c[0] + (c[2] + o) * c[1] for c in cases for o in offsets]).astype(io_dtype).astype(np.float64)
    B = q1.shape[0]
    params = _abi.default_params(flags=_abi.F_NULLSPACE | _abi.F_MIXER)
    w = synth.make_workload(chain, B, 1, seed=4, io_dtype=io_dtype)
    ctrl = np.zeros((B, 4))
    ctrl[:, 0] = 1.0
    eng = engine.Engine(chain, B, io_dtype=io_dtype, max_slots=2, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    states = oc.new_states(B, chain.n)
    for q in (q1, q2):
        got = eng.step_host(q, null_control=ctrl, want=("qdot_null", "status"))
        ref = oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], null_control=ctrl, states=states, want=("qdot_null", "status"))
    assert np.array_equal(got["status"], ref["status"])
    # what the decision looked at: the dot product of the two cycles' vectors, on the inputs as both sides saw them
    u1 = np.array([_unit_null(oc, chain, q) for q in q1])
    u2 = np.array([_unit_null(oc, chain, q) for q in q2])
    dots = np.abs((u1 * u2).sum(1))
    gq, rq = got["qdot_null"].astype(np.float64), ref["qdot_null"]
    live = ~(ref["status"] & _abi.ST_LIMIT_STOP).astype(bool)   # a limit stop zeroes the command on both sides
    assert live.sum() > B // 2
    same = np.abs(gq - rq).max(axis=1) < tol
    flipped = np.abs(gq + rq).max(axis=1) < tol
    # always a unit nullspace vector times the gain, with one sign or the other
    assert np.all((same | flipped)[live])
    assert np.allclose(np.linalg.norm(gq[live], axis=1), params.null_gain, atol=10 * tol)
    resolvable = dots > 1e-5
    assert resolvable.sum() >= 4 * len(cases)                   # the 1e-2 and 1e-4 offsets, most of the 1e-5 ones
    assert np.all(same[live & resolvable]), "sign decision differs from the double-state oracle at |v . lastvec| = %s" % dots[live & resolvable & ~same][:5]
    below = live & (dots < 1e-8)
    print("io %s: %d cases, %d resolvable all equal; |dot| < 1e-8: %d cases, %d with the other sign"
          % (np.dtype(io_dtype).name, B, int((live & resolvable).sum()), int(below.sum()), int((below & flipped & ~same).sum())))
    eng.close()
