"""Pipelined host path (vfik_submit_host / vfik_wait): the same numbers as the synchronous path, in
submission order (stateful nullspace sign memory included), with several submissions in flight."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, engine, robots, synth
    return dict(oc=oracle_c, abi=_abi, engine=engine, robots=robots, synth=synth)


@pytest.mark.parametrize("dt,pinned", [(np.float64, True), (np.float32, True), (np.float64, False)])
def test_pipeline_matches_the_oracle_in_submission_order(env, dt, pinned):
    abi = env["abi"]
    chain = env["robots"].lwr()
    B, K = 777, 8
    w = env["synth"].make_workload(chain, B, 4, seed=23, io_dtype=dt)
    params = abi.default_params(flags=abi.F_NULLSPACE | abi.F_MIXER)
    eng = env["engine"].Engine(chain, B, io_dtype=dt, max_slots=8, params=params)
    eng.set_fields(w["fields"], w["nfields"])
    rng = np.random.default_rng(4)
    mk = (lambda shape, d=dt: eng.host_array(shape, d)) if pinned else (lambda shape, d=dt: np.zeros(shape, dtype=d))
    qs, ctrls, outs, tickets = [], [], [], []
    q = w["q"].astype(dt)
    for k in range(K):  # K different joint states, all submitted before the first wait
        qk, ck = mk((B, 7)), mk((B, 4))
        qk[:] = np.clip(q + 0.05 * k * rng.normal(size=(B, 7)), 0.9 * chain.q_lo, 0.9 * chain.q_hi)
        ck[:] = rng.uniform(-1, 1, (B, 4))
        o = {"qdot_out": mk((B, 7)), "qdot_null": mk((B, 7)), "pose": mk((B, 16)), "status": mk((B,), np.int32)}
        qs.append(qk); ctrls.append(ck); outs.append(o)
        tickets.append(eng.submit_host(qk, o, null_control=ck))
    assert tickets == list(range(K))
    for t in reversed(tickets):  # waiting out of order is allowed
        eng.wait(t)
    eng.wait(tickets[0])  # and waiting twice is harmless
    states = env["oc"].new_states(B, 7)
    tol = 1e-9 if dt == np.float64 else 2e-5
    for k in range(K):  # the sign memory advanced in submission order
        ref = env["oc"].cycle_batch(chain, params, qs[k].astype(np.float64), w["fields"], w["nfields"],
                                    null_control=ctrls[k].astype(np.float64), states=states)
        assert np.abs(outs[k]["qdot_out"] - ref["qdot_out"]).max() < tol, k
        assert np.abs(outs[k]["qdot_null"] - ref["qdot_null"]).max() < tol, k
        assert np.abs(outs[k]["pose"] - ref["pose"]).max() < (1e-12 if dt == np.float64 else 1e-6)
        assert np.array_equal(outs[k]["status"], ref["status"])
    with pytest.raises(env["engine"].VfikError):
        eng.wait(K + 5)
    with pytest.raises(ValueError):
        eng.submit_host(qs[0][:, :6], outs[0])
    eng.close()


def test_pipeline_and_synchronous_path_interleave(env):
    """A synchronous vfik_step_host between two submissions sees and leaves consistent state."""
    abi = env["abi"]
    chain = env["robots"].lwr()
    B = 300
    w = env["synth"].make_workload(chain, B, 2, seed=5, io_dtype=np.float64)
    eng = env["engine"].Engine(chain, B, io_dtype=np.float64, max_slots=4, params=abi.default_params())
    eng.set_fields(w["fields"], w["nfields"])
    q = eng.host_array((B, 7))
    q[:] = w["q"]
    o1, o2 = {"qdot_out": eng.host_array((B, 7))}, {"qdot_out": eng.host_array((B, 7))}
    t1 = eng.submit_host(q, o1)
    mid = eng.step_host(w["q"])
    t2 = eng.submit_host(q, o2)
    eng.wait(t2)
    eng.wait(t1)
    assert np.array_equal(o1["qdot_out"], mid["qdot_out"]) and np.array_equal(o2["qdot_out"], mid["qdot_out"])
    eng.close()


def test_vfik_step_is_capturable_into_a_hip_graph():
    """A launch-bound caller that replays fixed device buffers can capture K vfik_step launches into ONE hipGraph (what
    bench.py --launch graph does): the captured launches give the results of direct launches, replay after replay."""
    import torch
    import __graft_entry__ as g
    g.build()
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    B = 4096 + 64
    w = synth.make_workload(chain, B, 3, seed=77, io_dtype=np.float32)
    for flags in (0, _abi.F_NULLSPACE | _abi.F_MIXER):
        eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=4, params=_abi.default_params(flags=flags))
        eng.set_fields(w["fields"], w["nfields"])
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            eng.use_stream(s.cuda_stream)
            q = torch.from_numpy(w["q"].astype(np.float32)).cuda()
            out_d = torch.zeros(B, 7, dtype=torch.float32, device="cuda")
            out_g = torch.zeros(B, 7, dtype=torch.float32, device="cuda")
            eng.step(eng.make_io(q, qdot_out=out_d))
            torch.cuda.synchronize()
            io_g = eng.make_io(q, qdot_out=out_g)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                for _ in range(5):
                    eng.step(io_g)
            epoch = eng.launch_epoch
            for _ in range(3):
                out_g.zero_()
                graph.replay()
                torch.cuda.synchronize()
                assert torch.equal(out_g, out_d)
            # what the capture baked in is valid for the handle's launch epoch: replays do not move it, a vfik_set_* call does
            assert eng.launch_epoch == epoch
            eng.set_fields(w["fields"], w["nfields"])
            assert eng.launch_epoch > epoch
        eng.close()


def test_sharded_engine_on_the_gpu_splits_a_batch_over_handles():
    """sharding.ShardedEngine with the real Engine: one process, the global batch split over two handles (both on device 0 here: a
    1-GPU box), global arrays in, global rows out == one handle over the whole batch."""
    import __graft_entry__ as g
    g.build()
    from vfclik_amd import _abi, engine, robots, sharding, synth
    chain = robots.lwr()
    B = 1000 + 37
    params = _abi.default_params(flags=_abi.F_NULLSPACE | _abi.F_MIXER)
    w = synth.make_workload(chain, B, 3, seed=5, io_dtype=np.float64)
    ctrl = np.random.default_rng(3).uniform(-1, 1, (B, 4))
    whole = engine.Engine(chain, B, io_dtype=np.float64, max_slots=4, params=params)
    whole.set_fields(w["fields"], w["nfields"])
    ref = whole.step_host(w["q"], null_control=ctrl, want=("qdot_out", "pose", "status"))
    whole.close()
    sh = sharding.ShardedEngine(chain, B, rank=0, world=1, devices=[0, 0], io_dtype=np.float64, max_slots=4, params=params)
    assert [e.batch for e in sh.engines] == [519, 518]
    sh.set_fields(w["fields"], w["nfields"])
    out = sh.step_global(w["q"], null_control=ctrl, want=("qdot_out", "pose", "status"))
    for k in ("qdot_out", "pose", "status"):
        assert np.array_equal(out[k], ref[k]), k   # per-arm decisions: the split changes nothing, bit for bit
    sh.close()


def test_output_rows_through_unaligned_pointers_and_partial_waves():
    """The LDS-tiled coalesced stores of the published rows (put_rows) need 16-byte aligned output pointers and full waves;
    anything else takes the lane-by-lane stores.  Same results either way: outputs handed in at a 4-byte offset, and a batch
    whose last wave is partial, against the aligned run of the same batch."""
    import torch
    import __graft_entry__ as g
    g.build()
    from vfclik_amd import _abi, engine, robots, synth
    chain = robots.lwr()
    B = 64 * 70 + 19                       # beyond the small-batch kernel's 4 096 arms; last wave partial
    w = synth.make_workload(chain, B, 5, seed=31, io_dtype=np.float32)
    params = _abi.default_params(flags=_abi.F_NULLSPACE | _abi.F_MIXER)
    shapes = {"qdot_out": 7, "qdot_vf": 7, "qdot_null": 7, "pose": 16, "pose_nt": 16, "qdist": 7, "v6": 6, "goal_dist": 2}
    res = []
    for off in (0, 1):
        eng = engine.Engine(chain, B, io_dtype=np.float32, max_slots=6, params=params)
        eng.set_fields(w["fields"], w["nfields"])
        eng.use_stream(torch.cuda.current_stream().cuda_stream)
        q = torch.from_numpy(w["q"].astype(np.float32)).cuda()
        flat = {k: torch.zeros(B * c + off, dtype=torch.float32, device="cuda") for k, c in shapes.items()}
        outs = {k: flat[k][off:] for k in shapes}                      # off = 1: every pointer 4 bytes past a 16-byte boundary
        assert all((outs[k].data_ptr() % 16 == 0) == (off == 0) for k in shapes)
        eng.step(eng.make_io(q, **outs))
        torch.cuda.synchronize()
        res.append({k: outs[k].cpu().numpy().reshape(B, shapes[k]).copy() for k in shapes})
        eng.close()
    for k in shapes:
        assert np.array_equal(res[0][k], res[1][k]), k
    assert np.abs(res[0]["pose"][:, 15] - 1.0).max() == 0.0 and np.abs(res[0]["qdot_out"]).max() > 1e-3


def test_bench_two_ranks_fall_back_to_gloo_when_rccl_cannot_start():
    """`bench.py --gpus 2` on ONE device: RCCL refuses two ranks on one GPU, so this is the rehearsal of a communicator that does not
    come up -- every rank must agree (through the store) on gloo for the barrier and the timing reductions, and the job must still
    print its line.  (The control path itself has no collective.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-device", "--steps", "10", "--warmup", "3", "--reps", "3",
                          "--no-cpu-baseline", "--rollout", "0", "--host-path", "0", "--state", "warm"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 1e9
    assert line["config"]["process_group"].startswith("gloo (RCCL communicator not available"), line["config"]["process_group"]
    assert line["max_abs_err_rad_s"] < 1e-6


def test_sharded_engine_runs_its_handles_concurrently():
    """ShardedEngine.step_host submits the cycle to every handle before it waits for the first (vfclik:88-105: the per-arm process
    sets of the reference run side by side).  On the one GPU of this box: two handles of 4 096 arms on device 0 through
    ShardedEngine(devices=[0, 0]) against ONE such handle -- run one after the other they would take 2 x; the bar is 1.7 x."""
    import time
    import __graft_entry__ as g
    g.build()
    from oracle import oracle_c
    from vfclik_amd import _abi, robots, sharding, synth
    chain = robots.lwr()
    B = 4096
    params = _abi.default_params(flags=_abi.F_NULLSPACE | _abi.F_MIXER)
    w = synth.make_workload(chain, 2 * B, 4, seed=3, io_dtype=np.float32)
    one = sharding.ShardedEngine(chain, B, rank=0, world=1, devices=[0], io_dtype=np.float32, max_slots=4, params=params)
    two = sharding.ShardedEngine(chain, 2 * B, rank=0, world=1, devices=[0, 0], io_dtype=np.float32, max_slots=4, params=params)
    one.set_fields(w["fields"][:B], w["nfields"][:B])
    two.set_fields(w["fields"], w["nfields"])
    q1, q2 = w["q"][:B].astype(np.float32), w["q"].astype(np.float32)
    want = ("qdot_out", "status")
    o1 = one.step_host(q1, want=want)
    o2 = two.step_host(q2, want=want)

    def rate(sh, q, into):
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(100):
                sh.step_host(q, want=want, into=into)
            best = min(best, (time.perf_counter() - t0) / 100)
        return best

    t1, t2 = rate(one, q1, o1), rate(two, q2, o2)
    print("one handle of %d arms: %.1f us per cycle; two handles through ShardedEngine(devices=[0, 0]): %.1f us (%.2f x)" % (B, t1 * 1e6, t2 * 1e6, t2 / t1))
    ref = oracle_c.cycle_batch(chain, params, w["q"], w["fields"], w["nfields"], want=("qdot_out",))
    assert np.abs(o2["qdot_out"] - ref["qdot_out"]).max() < 1e-6 and np.array_equal(o2["qdot_out"][:B], o1["qdot_out"])
    assert t2 < 1.7 * t1, (t1, t2)
    one.close()
    two.close()
