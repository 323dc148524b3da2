"""Parity of the HIP path (through the C-ABI) with the CPU oracle, on a real MI355X.

Tolerances (floating point path; north_star bar = 1e-6 rad/s on joint velocities):
  * io float64 : 1e-9  on every output (the arithmetic is float64 on both sides; differences come
                 from libm vs ocml sin/cos/atan2/pow and from operation order)
  * io float32 : 1e-6  on joint velocities and other outputs -- inputs are rounded to float32
                 BEFORE both the oracle and the kernel see them; the kernel's outputs are rounded to
                 float32 once on store (|qdot| <= ~7 rad/s, half-ulp <= 4.8e-7)
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL64 = 1e-9
TOL32 = 1e-6
ALL = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status")


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, engine, robots, synth
    lib = engine.load_library()
    assert lib.vfik_device_count() >= 1, "no HIP device: the gpu tests need an MI355X"

    class E:
        pass

    e = E()
    e.oc, e.abi, e.engine, e.robots, e.synth = oracle_c, _abi, engine, robots, synth
    return e


def _compare(got, ref, tol, keys):
    worst = {}
    for k in keys:
        if k == "status":
            assert np.array_equal(got[k], ref[k]), "status differs at arms %s" % np.nonzero(got[k] != ref[k])[0][:8]
            continue
        err = np.abs(got[k].astype(np.float64) - ref[k])
        assert np.all(np.isfinite(got[k])), k
        worst[k] = float(err.max())
        assert worst[k] < tol, "%s: max err %.3e (arm %d) exceeds %.1e" % (k, worst[k], int(np.argmax(err.max(axis=1))), tol)
    return worst


def _run_both(env, chain, params, w, io_dtype, want=ALL, null_control=None, max_slots=None, tool=None, ext=None):
    B = w["q"].shape[0]
    eng = env.engine.Engine(chain, B, io_dtype=io_dtype, max_slots=max_slots or max(1, 3 * w["fields"].shape[1]),
                            params=params)
    eng.set_fields(w["fields"], w["nfields"])
    per_arm = tool is not None and np.ndim(tool) == 2
    if tool is not None:
        eng.set_tool(tool, per_arm=per_arm)
    if ext is not None:
        for ch in range(4):
            eng.set_ext_cmd(2 + ch, ext[ch])
    got = eng.step_host(w["q"], null_control=null_control, want=want)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], tool=tool, null_control=null_control,
                             ext_cmd=ext)
    eng.close()
    return got, ref


# ---- BASELINE.json configs -------------------------------------------------------------------------
def test_c1_single_arm_against_both_oracles(env):
    """C1: one 7-DOF arm, one goal, no obstacles -- also against the reference-style NumPy loop."""
    from oracle import vfik_numpy as vn
    chain = env.robots.lwr()
    w = env.synth.make_workload(chain, 1, 0, seed=3, io_dtype=np.float64)
    params = env.abi.default_params(flags=env.abi.F_NULLSPACE | env.abi.F_MIXER)
    ctrl = np.array([[0.7, 0.0, 0.0, 0.0]])
    got, ref = _run_both(env, chain, params, w, np.float64, null_control=ctrl)
    _compare(got, ref, TOL64, ALL)
    arm = vn.ArmCycle(chain.B, chain.jtype, chain.q_lo, chain.q_hi, env.abi.params_to_dict(params))
    f = w["fields"][0][0]
    arm.set_fields({int(f["id"]): [float(f["force"]), int(f["type"]), f["p"][:17].tolist()]})
    r = arm.cycle(w["q"][0].tolist(), null_control=ctrl[0])
    for k in ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist"):
        assert np.abs(got[k][0] - r[k]).max() < TOL64, k


def test_c2_b4096_fp64(env):
    chain = env.robots.lwr()
    w = env.synth.make_workload(chain, 4096, 4, seed=0, io_dtype=np.float64)
    params = env.abi.default_params()
    got, ref = _run_both(env, chain, params, w, np.float64)
    worst = _compare(got, ref, TOL64, ALL)
    print("C2 worst errors", worst)


def test_c3_full_size_fp32(env):
    """C3 at BASELINE's full size: 65 536 arms x 8 obstacles, float32 I/O, every arm checked."""
    chain = env.robots.lwr()
    w = env.synth.make_workload(chain, 65536, 8, seed=0, io_dtype=np.float32)
    params = env.abi.default_params()
    got, ref = _run_both(env, chain, params, w, np.float32, want=("qdot_out", "qdot_vf", "status"))
    worst = _compare(got, ref, TOL32, ("qdot_out", "qdot_vf", "status"))
    print("C3 worst errors", worst)
    assert np.abs(ref["qdot_out"]).max() > 1.0  # the workload is not trivially zero


def test_c5_dual_arm_nullspace_joint_limit_task(env):
    """C5: 14-DOF chain, 16 obstacles, nullspace joint-limit task + mixer, float32 I/O."""
    chain = env.robots.lwr_dual14()
    w = env.synth.make_workload(chain, 8192, 16, seed=0, io_dtype=np.float32)
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_JOINT_LIMIT_TASK | f.F_MIXER)
    got, ref = _run_both(env, chain, params, w, np.float32)
    worst = _compare(got, ref, TOL32, ALL)
    print("C5 worst errors", worst)
    assert np.all(ref["status"] & f.ST_NULL_AMBIGUOUS)  # nullity 8: the SVD basis is not unique
    assert np.abs(ref["qdot_null"]).max() > 1e-3


def test_c5_full_size_lean_launch(env):
    """C5 at BASELINE's full size -- 65 536 arms x 14 joints x 16 obstacles, float32 I/O -- through the launch the
    bench times (qdot_out and status only: the LEAN kernel variant), every arm against the oracle."""
    chain = env.robots.lwr_dual14()
    w = env.synth.make_workload(chain, 65536, 16, seed=0, io_dtype=np.float32)
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_JOINT_LIMIT_TASK | f.F_MIXER)
    got, ref = _run_both(env, chain, params, w, np.float32, want=("qdot_out", "status"))
    worst = _compare(got, ref, TOL32, ("qdot_out", "status"))
    print("C5 full size worst errors", worst)
    assert np.abs(ref["qdot_out"]).max() > 1.0


def test_c3n_full_size_two_cycles(env):
    """C3 with the nullspace module and the mixer on (the default process set, vfclik:95-97) at 65 536 arms, float32
    I/O, LEAN launch: the first cycle takes the cold path of the null vector, the second (arms moved a little) the
    warm-started one; both against the oracle that carries its own sign memory."""
    chain = env.robots.lwr()
    B = 65536
    w = env.synth.make_workload(chain, B, 8, seed=0, io_dtype=np.float32)
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    ctrl = np.random.default_rng(5).uniform(-1, 1, (B, 4)).astype(np.float32).astype(np.float64)
    eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    states = env.oc.new_states(B, chain.n)
    q = w["q"]
    for cyc in range(3):
        got = eng.step_host(q, null_control=ctrl, want=("qdot_out", "status"))
        ref = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], null_control=ctrl, states=states, want=("qdot_out", "qdot_null", "status"))
        _compare(got, ref, TOL32, ("qdot_out", "status"))
        q = np.clip(q + 2e-3 * ref["qdot_out"], chain.q_lo, chain.q_hi).astype(np.float32).astype(np.float64)
    assert np.abs(ref["qdot_null"]).max() > 0.1
    eng.close()


def test_c3d_full_size_fp64_io(env):
    """C3's batch with float64 I/O (65 536 arms, 8 obstacles) at the float64 tolerance."""
    chain = env.robots.lwr()
    w = env.synth.make_workload(chain, 65536, 8, seed=0, io_dtype=np.float64)
    params = env.abi.default_params()
    got, ref = _run_both(env, chain, params, w, np.float64, want=("qdot_out", "status"))
    _compare(got, ref, TOL64, ("qdot_out", "status"))


@pytest.mark.parametrize("rank", [0, 5])
def test_c4_one_shard_of_the_524288_arm_batch(env, rank):
    """C4 = 524 288 arms over 8 GPUs with no collective on the data path: every rank computes a contiguous
    slice of the batch with its own handle.  One GPU is here: it takes the slice of `rank` out of the global batch
    (what bench.py's rank does with its own seed) and must reproduce the oracle's rows of exactly those arms."""
    from vfclik_amd import sharding
    chain = env.robots.lwr()
    total, world = 524288, 8
    lo, hi = sharding.shard_range(total, rank, world)
    assert hi - lo == 65536
    # the global batch is generated shard by shard (seed = global shard index), so no rank ever holds all of it
    w = env.synth.make_workload(chain, hi - lo, 8, seed=100 + rank, io_dtype=np.float32)
    params = env.abi.default_params()
    got, ref = _run_both(env, chain, params, w, np.float32, want=("qdot_out", "status"))
    _compare(got, ref, TOL32, ("qdot_out", "status"))


def test_c4_whole_batch_on_one_gpu(env):
    """All 524 288 arms of C4 in ONE handle on one GPU (eight times the one-wave-per-SIMD batch: the launch runs in
    rounds; index arithmetic beyond 2^19 arms x 24 quad planes), every arm against the oracle.  The eight shards are
    what the eight ranks of `bench.py --gpus 8` hold (their seeds), concatenated."""
    chain = env.robots.lwr()
    parts = [env.synth.make_workload(chain, 65536, 8, seed=1 + r, io_dtype=np.float32) for r in range(8)]
    w = {k: np.concatenate([p[k] for p in parts]) for k in ("q", "fields", "nfields")}
    params = env.abi.default_params()
    got, ref = _run_both(env, chain, params, w, np.float32, want=("qdot_out", "status"), max_slots=8)
    _compare(got, ref, TOL32, ("qdot_out", "status"))
    # a shard computed on its own gives the same rows (no arm sees another: SURVEY 8e)
    g5, _ = _run_both(env, chain, params, parts[5], np.float32, want=("qdot_out",), max_slots=8)
    assert np.array_equal(g5["qdot_out"], got["qdot_out"][5 * 65536:6 * 65536])


@pytest.mark.parametrize("name,nobs", [("powercube6", 3), ("lwr", 2)])
def test_other_joint_counts(env, name, nobs):
    chain = env.robots.by_name(name)
    w = env.synth.make_workload(chain, 512, nobs, seed=5, io_dtype=np.float64)
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    got, ref = _run_both(env, chain, params, w, np.float64)
    _compare(got, ref, TOL64, ALL)


def test_ten_joint_chain_with_prismatic(env):
    """n = 10 (the reference also drives a 10-joint iCub arm+torso) with two prismatic joints."""
    from vfclik_amd.chain import Chain
    rng = np.random.default_rng(11)
    segs = []
    for i in range(10):
        axis = rng.normal(size=3)
        tip = np.eye(4)
        tip[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        if np.linalg.det(tip[:3, :3]) < 0:
            tip[:3, 0] *= -1
        tip[:3, 3] = rng.uniform(-0.2, 0.2, 3)
        segs.append((1 if i in (2, 7) else 0, axis, tip))
    lo = np.where([s[0] == 1 for s in segs], -0.3, -2.5)
    chain = Chain.from_segments(segs, lo, -lo, name="rand10")
    w = env.synth.make_workload(chain, 256, 3, seed=2, io_dtype=np.float64)
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_JOINT_LIMIT_TASK | f.F_MIXER)
    got, ref = _run_both(env, chain, params, w, np.float64)
    _compare(got, ref, TOL64, ALL)


# ---- field primitives, ragged and empty field lists ---------------------------------------------------
def _mixed_fields(env, chain, B, seed):
    """Every primitive type, ragged counts (0 .. 7 fields), ids out of order, a second attractor."""
    rng = np.random.default_rng(seed)
    w = env.synth.make_workload(chain, B, 0, seed=seed, io_dtype=np.float64, max_fields=8)
    F = w["fields"]
    n = np.zeros(B, dtype=np.int32)
    for b in range(B):
        kinds = rng.permutation([1, 2, 2, 4, 5, 1, 0, 2])[: rng.integers(0, 8)]
        for k, t in enumerate(kinds):
            f = F[b, k]
            f["id"] = int(rng.integers(1, 40))
            f["type"] = int(t)
            f["p"][:] = 0.0
            f["force"] = 1.0
            if t == 1:
                f["force"] = float(rng.uniform(0.5, 2.0))
                f["p"][:16] = chain.fk(rng.uniform(0.8 * chain.q_lo, 0.8 * chain.q_hi))[0].reshape(16)
                f["p"][16] = rng.uniform(0.02, 0.2)
            elif t == 2:
                f["force"] = -10.0
                f["p"][:6] = [*rng.uniform(-0.8, 0.8, 3), rng.uniform(0.03, 0.1), 0.001, rng.choice([2.0, 5.0, 20.0, 2.5])]
            elif t == 4:
                f["force"] = -50.0
                f["p"][:8] = [*rng.uniform(-0.5, 0.5, 2), -0.5, *(rng.normal(size=2) * 0.1), 1.0, 0.05, 5.0]
            elif t == 5:
                f["force"] = 30.0
                f["p"][:10] = [*rng.uniform(-0.6, 0.6, 3), *rng.normal(size=3), 0.15, 10.0, 0.15, 2.0]
        n[b] = len(kinds)
    w["nfields"] = n
    return w


def test_all_primitive_types_ragged(env):
    chain = env.robots.lwr()
    w = _mixed_fields(env, chain, 2048, seed=21)
    assert (w["nfields"] == 0).any() and (w["nfields"] == 7).any()
    params = env.abi.default_params()
    got, ref = _run_both(env, chain, params, w, np.float64, max_slots=24)
    _compare(got, ref, TOL64, ALL)
    empty = w["nfields"] == 0
    assert np.all(got["qdot_out"][empty] == 0.0)  # only the seed field (type 0): no motion (vf:148-151)


@pytest.mark.parametrize("robot,flags,dt,tol", [("lwr", 0, np.float64, TOL64), ("lwr", 5, np.float32, TOL32),
                                                 ("lwr_dual14", 7, np.float64, TOL64), ("lwr_dual14", 0, np.float32, TOL32)])
def test_general_field_path_lean_launch(env, robot, flags, dt, tol):
    """Mixed primitive types (attractors, funnels, hemispheres, repellers of several orders; ragged) with nothing but
    qdot_out / status asked for: the LEAN variant of the general field path, 7 and 14 joints."""
    chain = env.robots.by_name(robot)
    w = _mixed_fields(env, chain, 1024, seed=33)
    for k in ("q",):
        w[k] = w[k].astype(dt).astype(np.float64)
    w["fields"]["p"] = w["fields"]["p"].astype(dt).astype(np.float64)
    w["fields"]["force"] = w["fields"]["force"].astype(dt).astype(np.float64)
    params = env.abi.default_params(flags=flags)
    got, ref = _run_both(env, chain, params, w, dt, want=("qdot_out", "status"), max_slots=24)
    _compare(got, ref, tol, ("qdot_out", "status"))
    assert np.abs(ref["qdot_out"]).max() > 0.1


def test_fractional_decay_order_takes_the_pow_path(env):
    chain = env.robots.lwr()
    w = env.synth.make_workload(chain, 256, 4, seed=8, io_dtype=np.float64)
    w["fields"]["p"][:, 1:, 5] = 3.7
    got, ref = _run_both(env, chain, env.abi.default_params(), w, np.float64)
    _compare(got, ref, TOL64, ("qdot_out", "v6"))


def test_goal_reached_and_half_turn(env):
    """Forced rare branches: tool exactly on the goal (D = 0, theta = 0), and orientation errors of
    pi - delta, delta in {1e-3, 1e-5, 1e-7, 1e-9, 0}, about assorted axes (the near-pi branch of the
    rotation logarithm; at exactly pi the axis sign is arbitrary, so only |omega| is compared)."""
    chain = env.robots.lwr()
    B = 88
    w = env.synth.make_workload(chain, B, 1, seed=4, io_dtype=np.float64)
    T = chain.fk(w["q"])
    rng = np.random.default_rng(0)
    deltas = [1e-3, 1e-5, 1e-7, 1e-9, 0.0]
    exact_pi = np.zeros(B, dtype=bool)
    for b in range(B):
        G = T[b].copy()
        if b >= 8:
            ax = rng.normal(size=3) if b % 3 else np.eye(3)[(b // 3) % 3]
            ax = ax / np.linalg.norm(ax)
            d = deltas[b % 5]
            exact_pi[b] = d == 0.0
            ang = np.pi - d
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            G[:3, :3] = (np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K) @ G[:3, :3]
            G[:3, 3] += rng.normal(size=3) * 0.1
        w["fields"]["p"][b, 0, :16] = G.reshape(16)
    got, ref = _run_both(env, chain, env.abi.default_params(), w, np.float64)
    _compare(got, ref, 1e-7, ("pose", "qdist"))
    # FK of the goal (host) and of the kernel agree to rounding: the residual command is ~1e-15
    assert np.abs(got["qdot_out"][:8]).max() < 1e-12 and np.abs(ref["qdot_out"][:8]).max() < 1e-12
    for k in ("qdot_out", "v6"):
        err = np.abs(got[k] - ref[k]).max(1)
        assert err[~exact_pi].max() < 1e-7, (k, err[~exact_pi].max())
    wn = np.linalg.norm(got["v6"][:, 3:], axis=1)
    assert np.abs(wn - np.linalg.norm(ref["v6"][:, 3:], axis=1)).max() < 1e-7
    assert np.abs(wn[8:] - 1.0).max() < 1e-9  # speedScale 1, theta > rot_slowdown: unit angular speed


# ---- tool, weights, parameters ---------------------------------------------------------------------------
def test_tool_shared_and_per_arm_weights_and_params(env):
    chain = env.robots.lwr()
    B = 1024
    w = env.synth.make_workload(chain, B, 4, seed=6, io_dtype=np.float64)
    rng = np.random.default_rng(6)
    params = env.abi.default_params(speed_scale=0.41, wy=[1, 1, 1, 0.3, 0.3, 0.1],
                                    wq=list(rng.uniform(0.2, 1.0, 7)) + [1.0] * 9, rot_slowdown=0.09)
    params.lambda_ = 0.03
    tool = np.eye(4)
    tool[:3, 3] = [0.0, 0.0, 0.2]  # old/README.old:84
    got, ref = _run_both(env, chain, params, w, np.float64, tool=tool.reshape(16))
    _compare(got, ref, TOL64, ALL)
    tools = np.tile(np.eye(4), (B, 1, 1))
    tools[:, :3, 3] = rng.uniform(-0.2, 0.2, (B, 3))
    c, s = np.cos(0.3), np.sin(0.3)
    tools[:, :3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    got, ref = _run_both(env, chain, params, w, np.float64, tool=tools.reshape(B, 16))
    _compare(got, ref, TOL64, ALL)


@pytest.mark.parametrize("robot,dt,tol,with_tool", [("lwr", np.float32, TOL32, False), ("lwr", np.float64, TOL64, False),
                                                    ("lwr", np.float32, TOL32, True), ("lwr", np.float64, TOL64, True),
                                                    ("lwr_dual14", np.float32, TOL32, False), ("lwr_dual14", np.float64, TOL64, True)])
def test_every_row_published_at_lane_per_arm_batch_sizes(env, robot, dt, tol, with_tool):
    """Every per-cycle row of a batch above the eight-lanes limit, so that full waves assemble their output tiles in LDS (the frames'
    16-column rows as swizzled 16-byte quads, both element sizes) and the batch's last, partial wave stores lane by lane: the
    publishing lean variant (no tool) and the general one (tool).  vf:341-342,462-466; nullspace:180-184; debug_jointlimits:69-73."""
    chain = getattr(env.robots, robot)()
    B = 4096 + 3 * 64 + 5
    w = env.synth.make_workload(chain, B, 4, seed=21, io_dtype=dt)
    params = env.abi.default_params(flags=env.abi.F_NULLSPACE | env.abi.F_MIXER | env.abi.F_JOINT_LIMIT_TASK)
    tool = None
    if with_tool:
        tool = np.eye(4)
        tool[:3, 3] = [0.02, -0.01, 0.2]
        tool = tool.reshape(16)
    got, ref = _run_both(env, chain, params, w, dt, tool=tool)
    _compare(got, ref, tol, ALL)
    assert np.abs(got["pose"][:, 15] - 1.0).max() == 0.0 and np.abs(got["pose_nt"][:, 12:15]).max() == 0.0


@pytest.mark.parametrize("robot,dt,tol,B", [("lwr_dual14", np.float64, TOL64, 4096 + 70), ("lwr_dual14", np.float32, TOL32, 300),
                                            ("lwr", np.float64, TOL64, 4096 + 70)])
def test_per_arm_tools_and_mixer_weights_use_the_tail_of_the_lds_region(env, robot, dt, tol, B):
    """Per-arm tools and per-arm mixer weights are the last rows of a wave's LDS region; launches whose full region does not fit a CU four
    times (float64 I/O, 14 joints: 44 KB) go as one wave per block and ask for those rows only when the options are set (launch_t).
    Both options set, then cleared again on the same handle; the straight-line and the general field path.  vf:321-332, command_mixer.py:48-53."""
    chain = getattr(env.robots, robot)()
    rng = np.random.default_rng(23)
    f = env.abi
    mw = [0.8, 0.5, 0.0, 0.0, 0.0, 0.0]
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER | f.F_JOINT_LIMIT_TASK, mix_w=mw)
    tools = np.tile(np.eye(4), (B, 1, 1))
    tools[:, :3, 3] = rng.uniform(-0.2, 0.2, (B, 3))
    c, s_ = np.cos(0.3), np.sin(0.3)
    tools[:, :3, :3] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
    tools = tools.reshape(B, 16).astype(dt).astype(np.float64)
    for general in (False, True):
        w = env.synth.make_workload(chain, B, 6, seed=24, io_dtype=dt)
        if general:
            w["fields"]["p"][B // 3, 2, 5] = 2.5   # a fractional decay order: the whole batch on the general path
        eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=8, params=params)
        eng.set_fields(w["fields"], w["nfields"])
        assert eng.field_path == (0 if general else 1)
        eng.set_tool(tools, per_arm=True)
        eng.set_mixer_weights(np.tile(mw, (B, 1)))
        got = eng.step_host(w["q"], want=ALL)
        ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], tool=tools)
        _compare(got, ref, tol, ALL)
        eng.set_mixer_weights(None)
        eng.set_tool(np.eye(4).reshape(16))
        got = eng.step_host(w["q"], want=ALL)
        ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"])
        _compare(got, ref, tol, ALL)
        eng.close()


@pytest.mark.parametrize("robot,dt,tol", [("lwr", np.float64, 1e-9), ("lwr", np.float32, 2e-5), ("lwr_dual14", np.float64, 1e-9)])
def test_ik_weights_of_each_arm(env, robot, dt, tol):
    """Every arm's vf process keeps its own 't' / 'j' weights (vf:164-179,295-309): three groups of arms with
    different weights in one batch, each compared with the oracle run on that group's weights."""
    chain = env.robots.by_name(robot)
    n, B = chain.n, 192
    f = env.abi
    flags = f.F_NULLSPACE | f.F_MIXER
    w = env.synth.make_workload(chain, B, 3, seed=16, io_dtype=dt)
    rng = np.random.default_rng(16)
    groups = [(slice(0, 64), [1.0] * 6, [1.0] * n),                                   # untouched arms: the batch weights
              (slice(64, 128), [1, 1, 1, 0.2, 0.2, 0.05], [1.0] * n),                 # 't' only
              (slice(128, 192), [0.5, 1, 1, 1, 0.3, 1], list(rng.uniform(0.1, 1.0, n)))]  # 't' and 'j'
    params = f.default_params(flags=flags)
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    eng.set_arm_weights(wy=np.tile(groups[1][1], (64, 1)), first_arm=64)
    eng.set_arm_weights(wy=np.tile(groups[2][1], (64, 1)), wq=np.tile(groups[2][2], (64, 1)), first_arm=128)
    want = ("qdot_vf", "qdot_out", "status")
    got = eng.step_host(w["q"], want=want)
    for sl, wy, wq in groups:
        p = f.default_params(flags=flags, wy=wy, wq=wq + [1.0] * (16 - n))
        ref = env.oc.cycle_batch(chain, p, w["q"][sl], w["fields"][sl], w["nfields"][sl])
        for k in ("qdot_vf", "qdot_out"):
            assert np.abs(got[k][sl] - ref[k]).max() < tol, (k, sl)
    # the groups really differ, and batch-wide weights set afterwards take every arm back
    assert np.abs(got["qdot_vf"][64:128] - env.oc.cycle_batch(chain, params, w["q"][64:128], w["fields"][64:128], w["nfields"][64:128])["qdot_vf"]).max() > 1e-3
    eng.set_params(wy=[1, 1, 1, 0.5, 0.5, 0.5])
    got = eng.step_host(w["q"], want=want)
    p = f.default_params(flags=flags, wy=[1, 1, 1, 0.5, 0.5, 0.5])
    ref = env.oc.cycle_batch(chain, p, w["q"], w["fields"], w["nfields"])
    assert np.abs(got["qdot_vf"] - ref["qdot_vf"]).max() < tol
    with pytest.raises(env.engine.VfikError):
        eng.set_arm_weights(wy=np.full((1, 6), np.nan))
    with pytest.raises(env.engine.VfikError):
        eng.set_arm_weights(wy=np.ones((2, 6)), first_arm=B - 1)
    eng.close()


# ---- nullspace module ---------------------------------------------------------------------------------------
def test_nullspace_sign_memory_over_a_trajectory(env):
    """Stateful parity: 40 cycles along a joint trajectory, /control active (nullspace:95-117)."""
    chain = env.robots.lwr()
    B = 512
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    w = env.synth.make_workload(chain, B, 2, seed=9, io_dtype=np.float64)
    rng = np.random.default_rng(9)
    dq = rng.normal(size=(B, 7)) * 0.06
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=4, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    states = env.oc.new_states(B, 7)
    q = w["q"].copy()
    flips = 0
    prev = None
    for t in range(40):
        ctrl = rng.uniform(-1, 1, (B, 4))
        got = eng.step_host(q, null_control=ctrl, want=("qdot_null", "qdot_out", "status"))
        ref = env.oc.cycle_batch(chain, params, q, w["fields"], w["nfields"], null_control=ctrl, states=states)
        _compare(got, ref, TOL64, ("qdot_null", "qdot_out", "status"))
        u = got["qdot_null"] / (ctrl[:, :1] * params.null_gain)
        if prev is not None:
            flips += int(((u * prev).sum(1) < 0).sum())
        prev = u
        q = np.clip(q + dq, 0.95 * chain.q_lo, 0.95 * chain.q_hi)
    assert flips == 0  # the tracked vector never jumps to its negative
    eng.reset_state()
    eng.close()


def test_check_limits_zeroes_the_null_command(env):
    chain = env.robots.lwr()
    B = 256
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER)
    w = env.synth.make_workload(chain, B, 0, seed=10, io_dtype=np.float64)
    w["q"][:, 3] = 0.999 * chain.q_hi[3]  # joint 3 almost on its upper limit
    ctrl = np.zeros((B, 4))
    ctrl[:, 0] = np.linspace(-6, 6, B)
    got, ref = _run_both(env, chain, params, w, np.float64, null_control=ctrl)
    _compare(got, ref, TOL64, ALL)
    stopped = (got["status"] & f.ST_LIMIT_STOP) != 0
    assert 0 < stopped.sum() < B
    assert np.all(got["qdot_null"][stopped] == 0.0)


# ---- mixer and limiter ------------------------------------------------------------------------------------------
def test_mixer_external_channels_and_limiter(env):
    chain = env.robots.lwr()
    B = 512
    f = env.abi
    rng = np.random.default_rng(12)
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER | f.F_LIMITER, mix_w=[1, 1, 0.5, -0.25, 2.0, 1.0], max_vel=0.8)
    w = env.synth.make_workload(chain, B, 3, seed=12, io_dtype=np.float64)
    ext = rng.normal(size=(4, B, 7))
    ctrl = rng.uniform(-1, 1, (B, 4))
    got, ref = _run_both(env, chain, params, w, np.float64, null_control=ctrl, ext=ext)
    _compare(got, ref, TOL64, ALL)
    assert np.abs(got["qdot_out"]).max() <= 0.8 + 1e-12
    assert ((got["status"] & f.ST_LIMITED) != 0).any()


@pytest.mark.parametrize("dt,with_mixer", [(np.float64, True), (np.float32, True), (np.float64, False)])
def test_limiter_speed_of_each_arm(env, dt, with_mixer):
    """Every bridge keeps its own max_vel (bridge:612-623): groups of arms with different limiter speeds in
    one batch, with and without per-arm mixer weights; a batch-wide change afterwards reaches every arm."""
    chain = env.robots.lwr()
    B = 192
    f = env.abi
    flags = f.F_LIMITER | (f.F_MIXER | f.F_NULLSPACE if with_mixer else 0)
    w = env.synth.make_workload(chain, B, 2, seed=19, io_dtype=dt)
    params = f.default_params(flags=flags, max_vel=0.5)
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=4, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    speeds = np.full(B, 0.5)
    speeds[64:128] = 0.05
    speeds[128:] = np.linspace(0.0, 0.3, 64)
    eng.set_max_vel(speeds[64:], first_arm=64)
    mixw = np.tile([1.0, 1.0, 0, 0, 0, 0], (B, 1))
    if with_mixer:
        mixw[::2, 1] = 0.3
        eng.set_mixer_weights(mixw)  # must not disturb the limiter speeds stored beside the weights
    got = eng.step_host(w["q"], want=("qdot_out", "status"))
    tol = 1e-9 if dt == np.float64 else 2e-6
    for b in (0, 1, 63, 64, 65, 127, 128, 129, 150, 191):
        p = f.default_params(flags=flags, max_vel=float(np.float64(dt(speeds[b]))), mix_w=list(mixw[b]))
        ref = env.oc.cycle_batch(chain, p, w["q"][b:b + 1], w["fields"][b:b + 1], w["nfields"][b:b + 1])
        assert np.abs(got["qdot_out"][b] - ref["qdot_out"][0]).max() < tol, b
        assert got["status"][b] == ref["status"][0], b
    assert np.abs(got["qdot_out"]).max(axis=1)[64:128].max() <= 0.05 * (1 + 1e-6)
    assert np.all(got["qdot_out"][128] == 0.0)  # max_vel 0: the arm stands still
    eng.set_params(max_vel=0.02)
    got = eng.step_host(w["q"], want=("qdot_out",))
    assert np.abs(got["qdot_out"]).max() <= 0.02 * (1 + 1e-6)
    with pytest.raises(env.engine.VfikError):
        eng.set_max_vel([-0.1])
    eng.close()


def test_nan_is_flagged_not_fatal(env):
    chain = env.robots.lwr()
    f = env.abi
    w = env.synth.make_workload(chain, 64, 1, seed=1, io_dtype=np.float64)
    w["q"][5, 2] = np.nan
    got, ref = _run_both(env, chain, f.default_params(), w, np.float64, want=("qdot_out", "status"))
    assert got["status"][5] & f.ST_NAN and ref["status"][5] & f.ST_NAN
    ok = np.arange(64) != 5
    assert np.abs(got["qdot_out"][ok] - ref["qdot_out"][ok]).max() < TOL64
    assert not (got["status"][ok] & f.ST_NAN).any()


@pytest.mark.parametrize("tag", ["k6n7", "k6n14"])
def test_vfik_mix_bit_exact_with_reference_golden(env, golden_dir, tag):
    """The mixing kernel alone reproduces the reference's CommandMixer sums bit for bit."""
    from oracle import vfik_numpy as vn
    from test_oracle_golden import _replay_mixer
    g = np.load(os.path.join(golden_dir, "mixer_golden.npz"))
    box = [0.0]
    captured = []

    class Spy(vn.CommandMixer):
        def read(self):
            r = super().read()
            captured.append((np.array(self.last_command), np.array(self.weights)))
            return r

    _replay_mixer(g, tag, lambda *a: Spy(*a, clock=lambda: box[0]), box)
    exp = g[tag + "__expect"]
    K, n, T = int(g[tag + "__K"]), int(g[tag + "__n"]), len(captured)
    chain = env.robots.lwr() if n == 7 else env.robots.lwr_dual14()
    # one "arm" per golden cycle would need per-arm weights; weights are per handle, so loop cycles
    eng = env.engine.Engine(chain, 1, io_dtype=np.float64, max_slots=1)
    d_cmd = eng.dev_alloc(K * n * 8)
    d_out = eng.dev_alloc(n * 8)
    for t in range(T):
        cmd, wts = captured[t]
        eng.h2d(d_cmd, cmd)
        eng.mix(d_cmd, wts, d_out)
        out = np.zeros(n)
        eng.d2h(out, d_out)
        a, b = np.nan_to_num(out, nan=7.0), np.nan_to_num(exp[t], nan=7.0)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), t
    eng.dev_free(d_cmd)
    eng.dev_free(d_out)
    eng.close()


# ---- ABI behaviour -------------------------------------------------------------------------------------------------
def test_partial_field_update_and_capacity_errors(env):
    chain = env.robots.lwr()
    B = 300  # not a multiple of the wave / block size
    w = env.synth.make_workload(chain, B, 3, seed=14, io_dtype=np.float64)
    params = env.abi.default_params()
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=3, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    assert eng.slots_in_use == 3
    # an obstacle moves for arms 100..149 only (one /param "add" per arm in the reference)
    w["fields"]["p"][100:150, 2, :3] += 0.05
    eng.set_fields(w["fields"][100:150], w["nfields"][100:150], first_arm=100)
    got = eng.step_host(w["q"], want=("qdot_out",))
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"])
    _compare(got, ref, TOL64, ("qdot_out",))
    # too many slots for the handle: refused, state unchanged
    big = env.synth.make_workload(chain, 1, 5, seed=1, io_dtype=np.float64)
    with pytest.raises(env.engine.VfikError, match="slots"):
        eng.set_fields(big["fields"], big["nfields"], first_arm=0)
    bad = w["fields"][:1].copy()
    bad["type"][0, 1] = 3  # no primitive 3 in the library (vf:238 "Unknown vector field type")
    with pytest.raises(env.engine.VfikError, match="unknown type"):
        eng.set_fields(bad, w["nfields"][:1])
    got2 = eng.step_host(w["q"], want=("qdot_out",))
    assert np.array_equal(got2["qdot_out"], got["qdot_out"])
    eng.close()


def test_device_pointer_path_on_torch_stream(env):
    """vfik_step with device pointers of torch tensors, launched on torch's current stream."""
    import torch
    chain = env.robots.lwr()
    B = 4096
    w = env.synth.make_workload(chain, B, 8, seed=15, io_dtype=np.float32)
    params = env.abi.default_params()
    eng = env.engine.Engine(chain, B, io_dtype=np.float32, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    dev = torch.device("cuda:0")
    q = torch.from_numpy(w["q"].astype(np.float32)).to(dev)
    out = torch.empty(B, 7, dtype=torch.float32, device=dev)
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    io = eng.make_io(q, qdot_out=out)
    for _ in range(3):
        eng.step(io)
    torch.cuda.synchronize()
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"])
    assert np.abs(out.cpu().numpy().astype(np.float64) - ref["qdot_out"]).max() < TOL32
    ms = eng.time_steps(io, 2, 5)
    assert ms > 0
    eng.close()


# ---- size-independent properties at full size --------------------------------------------------------------------
def test_properties_full_size(env):
    """B = 65 536: (1) speedScale linearity, (2) nullspace command lies in the kernel of J,
    (3) no goal -> no motion, (4) the mixer with weights [1,0,...] returns the vf command."""
    chain = env.robots.lwr()
    B = 65536
    f = env.abi
    w = env.synth.make_workload(chain, B, 8, seed=2, io_dtype=np.float32)
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER, mix_w=[1, 0, 0, 0, 0, 0])
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    ctrl = np.ones((B, 4))
    a = eng.step_host(w["q"], null_control=ctrl, want=("qdot_vf", "qdot_null", "qdot_out"))
    assert np.array_equal(a["qdot_out"], a["qdot_vf"])
    eng.set_params(speed_scale=0.25)
    b = eng.step_host(w["q"], null_control=ctrl, want=("qdot_vf",))
    assert np.abs(b["qdot_vf"] - 0.25 * a["qdot_vf"]).max() < 1e-12
    idx = np.random.default_rng(0).choice(B, 2000, replace=False)
    worst = 0.0
    for i in idx:
        J, _ = env.oc.jacobian(chain, w["q"][i])
        worst = max(worst, float(np.abs(J @ a["qdot_null"][i]).max()))
    assert worst < 1e-9, worst
    assert np.abs(np.linalg.norm(a["qdot_null"], axis=1) - params.null_gain).max() < 1e-9  # unit vector * gain
    w["nfields"][:] = 0
    eng.set_fields(w["fields"], w["nfields"])
    c = eng.step_host(w["q"], want=("qdot_vf",))
    assert np.all(c["qdot_vf"] == 0.0)
    eng.close()


@pytest.mark.parametrize("robot,dt,flags", [("lwr", np.float32, 0), ("lwr", np.float64, 0), ("powercube6", np.float64, 0),
                                            ("lwr", np.float64, 1 | 4), ("lwr", np.float32, 1 | 4 | 8), ("lwr_dual14", np.float32, 1 | 2 | 4),
                                            ("lwr_dual14", np.float64, 1 | 2 | 4), ("lwr_dual14", np.float64, 0)])
def test_lean_kernel_variant_only_qdot_out(env, robot, dt, flags):
    """q -> qdot_out and nothing else selects the LEAN kernel variant (no optional input or output compiled in):
    the same numbers as the full variant, which the same engine runs as soon as a second output is asked for."""
    chain = env.robots.by_name(robot)
    B = 1000
    w = env.synth.make_workload(chain, B, 6, seed=27, io_dtype=dt)
    params = env.abi.default_params(flags=flags, max_vel=0.7)
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    lean = eng.step_host(w["q"], want=("qdot_out", "status"))  # status is part of the LEAN variant too
    eng.reset_state()
    full = eng.step_host(w["q"], want=("qdot_out", "status", "pose"))
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"])
    tol = TOL64 if dt == np.float64 else 2e-6
    assert np.abs(lean["qdot_out"] - ref["qdot_out"]).max() < tol
    assert np.array_equal(lean["status"], ref["status"])
    assert np.abs(lean["qdot_out"].astype(np.float64) - full["qdot_out"]).max() < (1e-12 if dt == np.float64 else 5e-7)
    eng.close()


def test_lean_selection_honours_external_channels(env):
    """Only qdot_out requested, but mixer channels 2..5 carry commands: the launch must not take the LEAN variant
    (which has no external channels compiled in)."""
    chain = env.robots.lwr()
    B = 500
    f = env.abi
    params = f.default_params(flags=f.F_NULLSPACE | f.F_MIXER, mix_w=[1, 1, 0.5, 0.25, 0, 1])
    w = env.synth.make_workload(chain, B, 4, seed=33, io_dtype=np.float64)
    ext = np.random.default_rng(33).normal(size=(4, B, 7))
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    for ch in range(4):
        eng.set_ext_cmd(2 + ch, ext[ch])
    got = eng.step_host(w["q"], want=("qdot_out",))
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], ext_cmd=ext)
    assert np.abs(got["qdot_out"] - ref["qdot_out"]).max() < TOL64
    eng.close()


@pytest.mark.parametrize("robot,nobs", [("lwr_dual14", 9), ("lwr_dual14", 12), ("lwr_dual14", 13), ("lwr_dual14", 16), ("lwr_dual14", 17),
                                        ("lwr_dual14", 20), ("lwr_dual14", 24), ("lwr_dual14", 31), ("lwr10", 14), ("lwr", 19)])
def test_slot_counts_around_the_chunk_boundaries(env, robot, nobs):
    """More slots than one staged chunk (8 with float I/O): counts around every chunk boundary, ragged per arm, on the
    10- and 14-joint kernels (and 7) against the oracle."""
    if robot == "lwr10":  # a 10-joint chain (the library is built for 6, 7, 10, 14): the LWR behind a 3-joint arm
        from vfclik_amd.chain import Chain
        front = Chain.from_dh([(0.1, np.pi / 2, 0.2, 0.0), (0.3, 0.0, 0.0, 0.0), (0.0, np.pi / 2, 0.0, 0.0)], [-2.0] * 3, [2.0] * 3, name="front3")
        chain = front.concat(env.robots.lwr(), name="lwr10")
    else:
        chain = env.robots.by_name(robot)
    B = 200
    w = env.synth.make_workload(chain, B, nobs, seed=nobs, io_dtype=np.float32)
    w["nfields"] = (1 + (np.arange(B) * 7) % (nobs + 1)).astype(np.int32)  # 1 .. nobs+1 fields: ragged
    w["nfields"][:4] = nobs + 1
    flags = env.abi.F_NULLSPACE | env.abi.F_JOINT_LIMIT_TASK | env.abi.F_MIXER if chain.n >= 8 else 0
    params = env.abi.default_params(flags=flags)
    got, ref = _run_both(env, chain, params, w, np.float32, want=("qdot_out", "status"), max_slots=nobs)
    _compare(got, ref, TOL32, ("qdot_out", "status"))


def _goal_and_normal_scene(env, chain, B, dt, n_obstacles, rng, funnel_share=1.0, table_share=0.0):
    """What object_feeder sends for a goal with an approach vector (object_feeder:248-303): attractor (id 1), funnel attractor
    (id 2, force 30, orders 10 / 2), near-goal repeller (id 3, 5 cm up the approach axis), then point obstacles (ids 4 ...) and,
    for table_share of the arms, a table: a hemisphere repeller (ObstacleH, object_feeder:344-353: safe distance, order 5) behind them."""
    f = env.abi
    w = env.synth.make_workload(chain, B, n_obstacles, seed=int(rng.integers(1 << 30)), io_dtype=dt, max_fields=4 + n_obstacles)
    F = w["fields"]
    M = F.shape[1]
    # shift the obstacles behind the funnel and the near-goal repeller
    F[:, 3:3 + n_obstacles] = F[:, 1:1 + n_obstacles].copy()
    goal_p = F["p"][:, 0, [3, 7, 11]]
    axis = F["p"][:, 0, [2, 6, 10]]          # approach along the goal frame's z axis
    F["id"][:, 1], F["type"][:, 1], F["force"][:, 1] = 2, f.FIELD_FUNNEL, 30.0
    F["p"][:, 1] = 0.0
    F["p"][:, 1, 0:3] = goal_p
    F["p"][:, 1, 3:6] = axis
    F["p"][:, 1, 6:10] = [0.15, 10.0, 0.15, 2.0]
    F["id"][:, 2], F["type"][:, 2], F["force"][:, 2] = 3, f.FIELD_REPELLER, -10.0
    F["p"][:, 2] = 0.0
    F["p"][:, 2, 0:3] = goal_p + 0.05 * axis
    F["p"][:, 2, 3:6] = [0.15 + 0.05, 0.001, 5.0]
    k = 3 + n_obstacles
    F["id"][:, k], F["type"][:, k], F["force"][:, k] = 40, f.FIELD_HEMISPHERE, -50.0
    F["p"][:, k] = 0.0
    F["p"][:, k, 0:3] = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B), rng.uniform(-0.6, 0.1, B)], axis=1)
    F["p"][:, k, 3:6] = np.stack([rng.normal(0, 0.1, B), rng.normal(0, 0.1, B), np.ones(B)], axis=1) * rng.uniform(0.5, 2.0, (B, 1))
    F["p"][:, k, 6:8] = [0.05, 5.0]
    F["p"] = F["p"].astype(dt).astype(np.float64)
    w["nfields"][:] = M
    no_funnel = rng.uniform(size=B) >= funnel_share        # some arms: goal + obstacles only (type 0 leaves the entry empty)
    F["type"][no_funnel, 1] = f.FIELD_NULL
    F["type"][rng.uniform(size=B) >= table_share, k] = f.FIELD_NULL
    return w


@pytest.mark.parametrize("robot,B,dt,tol,flags,nobs", [
    ("lwr", 65536, np.float32, TOL32, 0, 5),        # the C3 batch in the goalAndNormal scene, lean
    ("lwr", 5000, np.float64, TOL64, 5, 5),         # default process set, every output (the publishing lean variant)
    ("lwr", 333, np.float64, TOL64, 7, 11),         # more repellers than one staged chunk, joint-limit task
    ("powercube6", 700, np.float32, TOL32, 5, 2),
    ("lwr_dual14", 4200, np.float32, TOL32, 7, 6),
    ("lwr", 64, np.float64, TOL64, 0, 0),           # funnel and near-goal repeller alone
])
@pytest.mark.parametrize("table_share", [0.0, 0.6])
def test_goal_and_normal_scene_on_the_straight_line_path(env, robot, B, dt, tol, flags, nobs, table_share):
    """handlers.go_cart with a normal -> object_feeder's goalAndNormal, with and without a table (ObstacleH): the funnel and the
    hemisphere travel in the aux block and the scene stays on the straight-line field path (vfik_field_path == 2); every output
    against the oracle, arms with and without a funnel / a table in one batch."""
    chain = env.robots.by_name(robot)
    rng = np.random.default_rng(B + nobs)
    w = _goal_and_normal_scene(env, chain, B, dt, nobs, rng, funnel_share=0.8, table_share=table_share)
    params = env.abi.default_params(flags=flags)
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=5 + nobs, params=params)
    eng.set_small_batch_kernel(0)
    eng.set_fields(w["fields"], w["nfields"])
    assert eng.field_path == 2
    want = ("qdot_out", "status") if flags == 0 and B > 10000 else ALL
    # (/control only where the nullspace is one-dimensional: with nullity >= 2 it is the declared gap, vfik_io.null_control)
    ctrl = rng.uniform(-1, 1, (B, 4)) if (flags & 1 and chain.n <= 7) else None
    got = eng.step_host(w["q"], null_control=ctrl, want=want)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], null_control=ctrl)
    _compare(got, ref, tol, want)
    # the same scene with ONE arm outside the pattern (a fractional decay order) takes the general path: same results
    F2 = w["fields"].copy()
    F2["p"][B // 2, 2, 5] = 3.5
    eng.set_fields(F2, w["nfields"])
    assert eng.field_path == 0
    got2 = eng.step_host(w["q"], null_control=ctrl, want=want)
    keep = np.arange(B) != B // 2
    for k in want:
        if k != "status":
            assert np.abs(got2[k][keep].astype(np.float64) - got[k][keep].astype(np.float64)).max() < tol
    # ... and back, by a partial update of that arm alone
    eng.set_fields(w["fields"][B // 2:B // 2 + 1], w["nfields"][B // 2:B // 2 + 1], first_arm=B // 2)
    assert eng.field_path == 2
    eng.close()


@pytest.mark.parametrize("robot,B,dt,tol,flags,nobs,want", [
    ("lwr", 65536, np.float32, TOL32, 0, 8, ("qdot_out", "status")),      # C3, lean
    ("lwr", 20000, np.float32, TOL32, 5, 8, None),                        # default process set, every row (publishing lean)
    ("lwr_dual14", 9000, np.float32, TOL32, 7, 16, ("qdot_out", "status")),   # C5's shape: two chunks of slots
    ("lwr", 3000, np.float64, TOL64, 7, 11, None),                        # float64 I/O: chunks of 4
    ("powercube6", 1000, np.float32, TOL32, 0, 3, ("qdot_out", "status")),
])
def test_uniform_repeller_image(env, monkeypatch, robot, B, dt, tol, flags, nobs, want):
    """Every decay repeller of the batch with the object feeder's safe distance and force (object_feeder:301-302,323,331): the lean
    launches read the uniform image (one quad per repeller, the pair in the batch constants).  Against the oracle; against the
    compact image (VFIK_UNIFORM_IMAGE=0); and a batch in which ONE arm's repeller differs falls back to the compact image."""
    chain = env.robots.by_name(robot)
    want = want or ALL
    w = env.synth.make_workload(chain, B, nobs, seed=31, io_dtype=dt)
    # ragged: a third of the arms with fewer obstacles (down to the goal alone, or nothing at all)
    rng = np.random.default_rng(B)
    few = rng.random(B) < 0.33
    w["nfields"][few] = rng.integers(0, nobs + 1, int(few.sum()))
    params = env.abi.default_params(flags=flags)
    ref = env.oc.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"])
    eng = env.engine.Engine(chain, B, io_dtype=dt, max_slots=nobs, params=params)
    eng.set_small_batch_kernel(0)
    # (first a sub-range only: the arms not yet given a field set must read as empty in the uniform image too)
    eng.set_fields(w["fields"][:B // 2], w["nfields"][:B // 2])
    nf_half = w["nfields"].copy()
    nf_half[B // 2:] = 0
    ref_half = env.oc.cycle_batch(chain, params, w["q"], w["fields"], nf_half, want=("qdot_out", "status"))
    _compare(eng.step_host(w["q"], want=("qdot_out", "status")), ref_half, tol, ("qdot_out", "status"))
    eng.reset_state()
    eng.set_fields(w["fields"], w["nfields"])
    assert eng.field_path == 1 and eng.uniform_repellers
    got = eng.step_host(w["q"], want=want)
    _compare(got, ref, tol, want)
    monkeypatch.setenv("VFIK_UNIFORM_IMAGE", "0")
    eng0 = env.engine.Engine(chain, B, io_dtype=dt, max_slots=nobs, params=params)
    monkeypatch.delenv("VFIK_UNIFORM_IMAGE")
    eng0.set_small_batch_kernel(0)
    eng0.set_fields(w["fields"], w["nfields"])
    assert eng0.field_path == 1 and not eng0.uniform_repellers
    got0 = eng0.step_host(w["q"], want=want)
    _compare(got0, ref, tol, want)
    eng0.close()
    # one arm's third repeller with another safe distance: the compact image for everybody, same results but for that arm
    F2 = w["fields"].copy()
    odd = int(np.nonzero(w["nfields"] == nobs + 1)[0][B // 200])
    F2["p"][odd, min(3, nobs), 4] = np.float64(dt(0.004))
    eng.set_fields(F2[odd:odd + 1], w["nfields"][odd:odd + 1], first_arm=odd)
    assert eng.field_path == 1 and not eng.uniform_repellers
    ref2 = env.oc.cycle_batch(chain, params, w["q"], F2, w["nfields"])
    eng.reset_state()
    _compare(eng.step_host(w["q"], want=want), ref2, tol, want)
    # ... and back; then a batch-wide change of the pair (another force for every repeller): the constants follow
    eng.set_fields(w["fields"][odd:odd + 1], w["nfields"][odd:odd + 1], first_arm=odd)
    assert eng.uniform_repellers
    F3 = w["fields"].copy()
    F3["force"][:, 1:1 + nobs] = -6.0
    F3["p"][:, 1:1 + nobs, 4] = np.float64(dt(0.002))
    eng.set_fields(F3, w["nfields"])
    assert eng.uniform_repellers
    ref3 = env.oc.cycle_batch(chain, params, w["q"], F3, w["nfields"])
    eng.reset_state()
    _compare(eng.step_host(w["q"], want=want), ref3, tol, want)
    # a repeller whose radius + safe distance is NEGATIVE (an attracting "repeller": the law allows it) cannot live in the uniform image,
    # whose kernel clamps that sum at 0 to disarm unused slots: the compact image, same results as the oracle
    F4 = F3.copy()
    F4["p"][odd, min(3, nobs), 3] = np.float64(dt(-0.05))
    eng.set_fields(F4[odd:odd + 1], w["nfields"][odd:odd + 1], first_arm=odd)
    assert eng.field_path == 1 and not eng.uniform_repellers
    ref4 = env.oc.cycle_batch(chain, params, w["q"], F4, w["nfields"])
    eng.reset_state()
    _compare(eng.step_host(w["q"], want=want), ref4, tol, want)
    eng.close()


def test_field_path_classification(env):
    chain = env.robots.lwr()
    f = env.abi
    B = 8
    w = env.synth.make_workload(chain, B, 3, seed=2, io_dtype=np.float64, max_fields=6)
    eng = env.engine.Engine(chain, B, io_dtype=np.float64, max_slots=12)
    eng.set_fields(w["fields"], w["nfields"])
    assert eng.field_path == 1
    F = w["fields"].copy()
    n = w["nfields"].copy()
    F[0, 4]["id"], F[0, 4]["type"], F[0, 4]["force"] = 2, f.FIELD_FUNNEL, 30.0
    F[0, 4]["p"][:10] = [0.3, 0.2, 0.5, 0, 0, 1, 0.15, 10.0, 0.15, 2.0]
    n[0] = 5
    eng.set_fields(F, n)
    assert eng.field_path == 2                     # one funnel with integer orders: the funnel block
    F[0, 5] = F[0, 4]
    F[0, 5]["id"] = 50
    n[0] = 6
    eng.set_fields(F, n)
    assert eng.field_path == 0                     # a second funnel: general
    n[0] = 5
    F[0, 4]["p"][7] = 2.5
    eng.set_fields(F, n)
    assert eng.field_path == 0                     # fractional angle order: general
    F[0, 4]["p"][7] = 10.0
    F[1, 4]["id"], F[1, 4]["type"], F[1, 4]["force"] = 9, f.FIELD_HEMISPHERE, -50.0
    F[1, 4]["p"][:8] = [0.2, -0.1, -0.5, 0.05, -0.02, 1.0, 0.05, 5.0]
    n[1] = 5
    eng.set_fields(F, n)
    assert eng.field_path == 2                     # one hemisphere with an integer order: the aux block as well
    F[1, 4]["p"][7] = 4.5
    eng.set_fields(F, n)
    assert eng.field_path == 0                     # fractional order: general
    F[1, 4]["p"][7] = 5.0
    F[1, 5] = F[1, 4]
    F[1, 5]["id"] = 10
    n[1] = 6
    eng.set_fields(F, n)
    assert eng.field_path == 0                     # a second hemisphere: general
    n[1] = 5
    F[2, 4]["id"], F[2, 4]["type"], F[2, 4]["force"] = 11, f.FIELD_ATTRACTOR, 1.0
    F[2, 4]["p"][:17] = F[2, 0]["p"][:17]
    n[2] = 5
    eng.set_fields(F, n)
    assert eng.field_path == 0                     # a second attractor: general
    eng.close()
