"""ControlCycleBatch's host logic without a GPU (round 4: mailbox-driven polling, array-backed bottles to listened ports only,
write_encoders): a stand-in Engine whose outputs are simple functions of q, so that what is checked is WHICH arms ran, WHICH
bottles went out and that the two ways of feeding joint angles are the same thing."""
import numpy as np
import pytest

from vfclik_amd import ports as yarp
from vfclik_amd import robots, vf_module


class _Engine:
    calls = []

    def __init__(self, chain, B, io_dtype=np.float64, **kw):
        self.batch, self.n, self.io_dtype, self.n_objects = B, chain.n, np.dtype(io_dtype), 0
        self.cols = {"qdot_vf": self.n, "qdot_null": self.n, "qdot_out": self.n, "pose": 16, "pose_nt": 16, "v6": 6, "qdist": self.n,
                     "goal_dist": 2, "q_ref_out": self.n, "track_error": 8}

    def set_fields(self, *a, **k):
        pass

    def set_objects(self, frames):
        self.n_objects = frames.shape[1]

    def set_speed_scale(self, *a, **k):
        pass

    def set_mixer_weights(self, w):
        pass

    def set_ext_cmd(self, ch, arr):
        _Engine.calls.append(("ext", ch, arr.copy()))

    def step_host(self, q, null_control=None, want=(), into=None, active=None, **kw):
        _Engine.calls.append(("step", None if active is None else np.asarray(active).copy()))
        act = np.ones(self.batch, dtype=bool) if active is None else np.asarray(active).astype(bool)
        out = {}
        for k in want:
            if into is not None and k in into:
                arr = into[k]
            elif k == "status":
                arr = np.zeros(self.batch, np.int32)
            elif k == "obj_dist":
                arr = np.zeros((self.batch, self.n_objects, 2))
            else:
                arr = np.zeros((self.batch, self.cols[k]))
            if k not in ("status", "obj_dist"):
                arr[act] = (q[act].sum(axis=1)[:, None] + np.arange(arr.shape[1])) * {"qdot_vf": 1.0, "pose": 2.0}.get(k, 0.5)
            out[k] = arr
        return out

    def close(self):
        pass


@pytest.fixture
def cb(monkeypatch):
    yarp.Network.reset()
    monkeypatch.setattr(vf_module, "Engine", _Engine)
    _Engine.calls = []
    chain = robots.lwr()
    bases = ["/%d/lwr/right" % i for i in range(6)]
    c = vf_module.ControlCycleBatch(chain, bases)
    yield c
    c.close()
    yarp.Network.reset()


def _reader(src, name):
    r = yarp.BufferedPortBottle()
    r.open(name)
    yarp.Network.connect(src, name)
    return r


def _feed(cb, a, q):
    p = cb.ports[a]["encoders"]
    b = p.prepare()
    b.clear()
    for v in q:
        b.addDouble(float(v))
    p.write()


def test_batched_and_per_arm_joint_angles_are_the_same_cycle(cb):
    q = np.random.default_rng(0).uniform(-1, 1, (6, 7))
    out = _reader("/2/lwr/right/vectorField/qdotOut", "/t/qdot2")
    for a in range(6):
        _feed(cb, a, q[a])
    got = cb.cycle()
    assert got.all() and _Engine.calls[-1] == ("step", None)          # every arm fresh: no gate passed down
    first = {k: (v.copy() if v is not None else None) for k, v in cb.last.items()}
    b1 = out.read(False)
    cb.write_encoders(q)
    got = cb.cycle()
    assert got.all()
    for k, v in first.items():
        assert v is None or np.array_equal(v, cb.last[k]), k
    b2 = out.read(False)
    assert b1.size() == b2.size() == 7 and [b1.get(i).asDouble() for i in range(7)] == [b2.get(i).asDouble() for i in range(7)]
    assert b2.get(0).isDouble() and b2.toString().count(" ") == 6       # an array-backed bottle reads like any other
    # a sub-range: only those arms run, the others keep their rows
    cb.write_encoders(q[[1, 4]] + 1.0, arms=[1, 4])
    got = cb.cycle()
    assert list(np.nonzero(got)[0]) == [1, 4]
    step = _Engine.calls[-1]
    assert step[0] == "step" and list(np.nonzero(step[1])[0]) == [1, 4]
    assert np.array_equal(cb.last["qdot_vf"][0], first["qdot_vf"][0]) and not np.array_equal(cb.last["qdot_vf"][1], first["qdot_vf"][1])
    # nothing arrived: no launch at all
    n = len(_Engine.calls)
    assert not cb.cycle().any() and len(_Engine.calls) == n


def test_only_ports_somebody_reads_get_bottles_and_only_ports_with_mail_are_polled(cb, monkeypatch):
    q = np.zeros((6, 7))
    written = []
    orig = yarp.BufferedPortBottle.write_bottle

    def spy(self, bottle, strict=False):
        written.append(self.getName())
        return orig(self, bottle, strict)

    monkeypatch.setattr(yarp.BufferedPortBottle, "write_bottle", spy)
    cb.write_encoders(q)
    cb.cycle()
    assert written == []                                               # nobody listens: nothing is built, nothing is written
    r_pose = _reader("/3/lwr/right/vectorField/pose", "/t/pose3")
    r_mix = _reader("/5/lwr/right/bridge/mixed", "/t/mixed5")
    cb.write_encoders(q)
    cb.cycle()
    assert sorted(written) == ["/3/lwr/right/vectorField/pose", "/5/lwr/right/bridge/mixed"]
    assert r_pose.read(False).size() == 16 and r_mix.read(False).size() == 7
    r_pose.close()                                                      # a closed reader is no reader
    del written[:]
    cb.write_encoders(q)
    cb.cycle()
    assert written == ["/5/lwr/right/bridge/mixed"]
    # polling: reads happen on ports with mail only
    reads = []
    orig_read = yarp.BufferedPortBottle.read

    def spy_read(self, wait=True):
        reads.append(self.getName())
        return orig_read(self, wait)

    monkeypatch.setattr(yarp.BufferedPortBottle, "read", spy_read)
    ctl = yarp.BufferedPortBottle()
    ctl.open("/t/ctl")
    yarp.Network.connect("/t/ctl", "/4/lwr/right/nullspace/control")
    b = ctl.prepare()
    b.clear()
    b.addDouble(0.25)
    ctl.write()
    cb.write_encoders(q)
    cb.cycle()
    assert [r for r in reads if "/t/" not in r] == ["/4/lwr/right/nullspace/control"]
    assert cb.control[4, 0] == 0.25 and not cb.control[[0, 1, 2, 3, 5]].any()


def test_watchdog_zeroes_a_silent_channel_for_every_arm_at_once(cb):
    """command_mixer.py:56-66 on arrays: a /bridge/jointcmd command is kept for guard_time, then zeroed -- without visiting any port."""
    now = [100.0]
    cb.clock = lambda: now[0]
    cb.ext_time[:] = now[0]
    src = yarp.BufferedPortBottle()
    src.open("/t/jc")
    yarp.Network.connect("/t/jc", "/2/lwr/right/bridge/jointcmd")
    b = src.prepare()
    b.clear()
    for v in range(7):
        b.addDouble(0.1 * (v + 1))
    src.write()
    cb.write_encoders(np.zeros((6, 7)))
    cb.cycle()
    assert cb.ext[0, 2, 6] == pytest.approx(0.7) and cb._n_ext_live == 1
    now[0] += 1.0
    cb.write_encoders(np.zeros((6, 7)))
    cb.cycle()
    assert cb.ext[0, 2, 6] == pytest.approx(0.7)
    now[0] += 1.5                                                       # 2.5 s of silence > guard_time 2.0
    cb.write_encoders(np.zeros((6, 7)))
    cb.cycle()
    assert not cb.ext.any() and cb._n_ext_live == 0
    assert [c for c in _Engine.calls if c[0] == "ext"][-1][2].sum() == 0.0
    # a wrong-length bottle is ignored (and warned about)
    b = src.prepare()
    b.clear()
    b.addDouble(1.0)
    src.write()
    cb.write_encoders(np.zeros((6, 7)))
    cb.cycle()
    assert not cb.ext.any()


def test_encoders_reach_outside_readers_of_a_batched_write(cb):
    """bridge:578-582 fans the encoders out to whoever connected (the handlers' /encoders ports): a batched write still delivers there."""
    r = _reader("/1/lwr/right/bridge/encoders", "/t/enc1")
    q = np.arange(42, dtype=float).reshape(6, 7)
    cb.write_encoders(q)
    got = r.read(False)
    assert [got.get(i).asDouble() for i in range(7)] == list(q[1])
    assert cb.cycle().all()
