"""Host layer without a GPU: port semantics, /param acceptance rules (vf:209-275), object-feeder
translation (object_feeder:111-359) and the bottles the handler API sends (handlers.py)."""
import numpy as np
import pytest

from vfclik_amd import _abi
from vfclik_amd import ports as yarp
from vfclik_amd.fields import FieldSets


@pytest.fixture(autouse=True)
def fresh_network():
    yarp.Network.reset()
    yield
    yarp.Network.reset()


def _port(name, strict=False):
    p = yarp.BufferedPortBottle()
    p.open(name)
    p.setStrict(strict)
    return p


def _bottle(*items):
    return yarp.Bottle(items)


# ---- ports -------------------------------------------------------------------------------------------
def test_non_strict_reader_sees_only_the_newest_message():
    w, r = _port("/w"), _port("/r")
    yarp.Network.connect("/w", "/r")
    for k in range(3):
        b = w.prepare()
        b.clear()
        b.addDouble(k)
        w.write()
    assert r.getPendingReads() == 1
    assert r.read(False).get(0).asDouble() == 2.0
    assert r.read(False) is None


def test_strict_reader_is_a_fifo_and_unconnected_writes_are_dropped():
    w, r = _port("/w"), _port("/r", strict=True)
    b = w.prepare()
    b.addInt(7)
    w.write()  # not connected yet: dropped
    yarp.Network.connect("/w", "/r")
    assert yarp.Network.isConnected("/w", "/r") and not yarp.Network.isConnected("/r", "/w")
    for k in range(3):
        b = w.prepare()
        b.addInt(k)
        w.writeStrict()
    assert [r.read(False).get(0).asInt() for _ in range(3)] == [0, 1, 2]
    assert r.read(False) is None


def test_bottle_values_and_nested_lists():
    b = yarp.Bottle()
    b.addString("add")
    b.addInt(4)
    b.addDouble(-10)
    lst = b.addList()
    for v in (0.1, 0.2, 0.3):
        lst.addDouble(v)
    assert b.size() == 4 and b.get(0).toString() == "add" and b.get(1).asInt() == 4
    assert b.get(2).isDouble() and not b.get(2).isInt() and b.get(1).isInt()
    assert b.get(3).asList().size() == 3 and b.get(3).asList().get(2).asDouble() == 0.3
    assert b.get(9).asDouble() == 0.0  # out of range reads are harmless, like YARP
    assert b.toString() == "add 4 -10.0 (0.1 0.2 0.3)"
    c = b.copy()
    c.get(3).asList().addDouble(9.0)
    assert b.get(3).asList().size() == 3  # deep copy: a written bottle cannot be edited by the sender


# ---- /param rules ---------------------------------------------------------------------------------------
def test_param_add_remove_and_ignore_rules():
    fs = FieldSets(2, max_fields=4)
    goal = list(np.eye(4).reshape(16)) + [0.05]
    assert fs.handle_param(0, _bottle("add", 1, 1.0, 1, goal))
    assert fs.handle_param(0, _bottle("add", 4, -10.0, 2, [0.1, 0.2, 0.3, 0.05, 0.001, 5.0]))
    assert fs.sets[0][4] == [-10.0, 2, [0.1, 0.2, 0.3, 0.05, 0.001, 5.0]]
    assert not fs.handle_param(0, _bottle("add", 5, -10.0, 2))          # size != 5 (vf:227,265-266)
    assert not fs.handle_param(0, _bottle("add", 5, -10.0, 3, [1.0]))   # unknown type (vf:238,263-264)
    assert not fs.handle_param(0, _bottle("add", 5, -10.0, 2, [1.0, 2.0]))  # too few parameters
    assert not fs.handle_param(0, _bottle("remove", 9))                 # absent id: nothing (vf:269-273)
    assert not fs.handle_param(0, _bottle("remove", 4, 1))              # size != 2 (vf:268,274-275)
    assert not fs.handle_param(0, _bottle("frobnicate", 1))
    assert fs.handle_param(0, _bottle("add", 4, -10.0, 2, [9.0, 9.0, 9.0, 0.05, 0.001, 5.0]))  # same id: replaced
    assert len(fs.sets[0]) == 2 and fs.sets[0][4][2][0] == 9.0
    assert fs.handle_param(0, _bottle("remove", 4)) and 4 not in fs.sets[0]
    assert fs.dirty == {0} and fs.sets[1] == {}
    rec, cnt = fs.records([0, 1])
    assert list(cnt) == [1, 0] and rec["id"][0, 0] == 1 and rec["type"][0, 0] == 1 and rec["p"][0, 0, 16] == 0.05


def test_capacity_is_enforced_without_raising():
    fs = FieldSets(1, max_fields=2)
    for k in range(3):
        ok = fs.handle_param(0, _bottle("add", 4 + k, -10.0, 2, [0.0, 0.0, 0.0, 0.05, 0.001, 5.0]))
        assert ok == (k < 2)


# ---- object feeder ------------------------------------------------------------------------------------------
def _drain(port):
    out = []
    while True:
        b = port.read(False)
        if b is None:
            return out
        out.append(b.tolist())


def test_object_feeder_translation():
    from vfclik_amd.object_feeder import ObjectFeeder
    base = "/0/lwr/right"
    param_in = _port(base + "/vectorField/param", strict=True)
    of = ObjectFeeder(base)
    user = _port("/user")
    yarp.Network.connect("/user", base + "/ofeeder/object")

    def send(*items):
        b = user.prepare()
        for it in items:
            b.add(it)
        user.writeStrict()
        of.spin_once()

    obst = [1, 0, 0, 0.0, 0, 1, 0, -0.4, 0, 0, 1, 0.4, 0, 0, 0, 1, 0.05, 20]  # old/README.old:75
    send("set", "ObstacleP", 0, [float(x) for x in obst])
    assert _drain(param_in) == []  # no goal yet: nothing is sent (object_feeder:355-359)
    goal = [0, 1, 0, 0, -1, 0, 0, 0.3, 0, 0, 1, 1.1, 0, 0, 0, 1, 0.1]  # old/README.old:69
    send("set", "goal", [float(x) for x in goal])
    msgs = _drain(param_in)
    assert msgs[0] == ["add", 1, 1.0, 1, [float(x) for x in goal]]
    assert msgs[1] == ["remove", 2] and msgs[2] == ["remove", 3]
    assert msgs[3] == ["add", 5, -10.0, 2, [0.0, -0.4, 0.4, 0.05, 0.001, 20.0]]
    hemi = [1, 0, 0, 0, 0, 1, 0, -0.4, 0, 0, 1, 0.3, 0, 0, 0, 1, 0, 0, 1, 0.001, 5]  # old/README.old:78
    send("set", "ObstacleH", 1, [float(x) for x in hemi])
    msgs = _drain(param_in)
    assert msgs[-1] == ["add", 6, -50.0, 4, [0.0, -0.4, 0.3, 0.0, 0.0, 1.0, 0.001, 5.0]]
    send("remove", 0)
    msgs = _drain(param_in)
    assert msgs[0] == ["remove", 5]
    gan = [1, 0, 0, 0.4, 0, -1, 0, -0.4, 0, 0, -1, 0.4, 0, 0, 0, 1, 0, -1, 0, 0.1, 0.15, 0.15]  # old/README.old:72-73
    send("set", "goalAndNormal", [float(x) for x in gan])
    msgs = _drain(param_in)
    assert msgs[0][:4] == ["add", 1, 1.0, 1] and msgs[0][4][16] == 0.15
    assert msgs[1] == ["add", 2, 30.0, 5, [0.4, -0.4, 0.4, 0.0, -1.0, 0.0, 0.1, 10.0, 0.15, 2.0]]
    assert msgs[2][:4] == ["add", 3, -10.0, 2]
    assert np.allclose(msgs[2][4], [0.4, -0.45, 0.4, 0.2, 0.001, 5.0])
    of.close()


# ---- handlers ----------------------------------------------------------------------------------------------------
def test_handle_arm_new_bottles():
    from vfclik_amd.handlers import HandleArmNew
    base = "/0/lwr/right"
    objp, wp, toolp, vfw, ref = (_port(base + s, strict=True) for s in
                                 ("/ofeeder/object", "/bridge/weight", "/vectorField/tool", "/vectorField/weight", "/jpctrl/ref"))
    h = HandleArmNew()
    frame = [float(x) for x in np.eye(4).reshape(16)]
    h.go_cart(frame)
    assert objp.read(False).tolist() == ["set", "goal", frame + [0.1]]       # handlers.py:118-128
    assert wp.read(False).tolist() == [1, 1, 0, 0]                            # set_cartesian_control: cart + null
    h.go_joint([0.1] * 7)
    assert ref.read(False).tolist() == [0.1] * 7
    assert wp.read(False).tolist() == [0, 0, 1, 0]
    h.set_wik_cart_weights([1, 1, 1, 0.1, 0.1, 0.1])
    assert vfw.read(False).tolist() == ["t", 1.0, 1.0, 1.0, 0.1, 0.1, 0.1]
    h.set_wik_joint_weights([0.5] * 7)
    assert vfw.read(False).tolist() == ["j"] + [0.5] * 7
    h.set_tool(frame)
    assert toolp.read(False).tolist() == frame


def test_handle_arm_and_bridge_bottles():
    from vfclik_amd.handlers import HandleArm, HandleBridge, HandleJController
    base = "/lwr/right"
    objp = _port(base + "/ofeeder/object", strict=True)
    wp = _port(base + "/bridge/weight", strict=True)
    vfw = _port(base + "/vectorField/weight", strict=True)
    ref = _port(base + "/jpctrl/ref", strict=True)
    h = HandleArm(base)
    h.gotoPos([0.5, 0.1, 0.9])
    msg = objp.read(False).tolist()
    assert msg[:2] == ["set", "goal"] and msg[2][3] == 0.5 and msg[2][7] == 0.1 and msg[2][11] == 0.9 and msg[2][16] == 0.1
    hb = HandleBridge(base)
    hb.cartesian_controller()
    assert wp.read(False).tolist() == [1, 1, 0, 0]
    hb.joint_controller()
    assert wp.read(False).tolist() == [0, 0, 1, 0]
    hb.set_weights("task", [1, 1, 1, 1, 1, 1])
    assert vfw.read(False).tolist()[0] == "t"
    hj = HandleJController(base)
    res, diff = hj.set_ref_js([0.0] * 7)
    assert res is False and ref.read(False).tolist() == [0.0] * 7


def test_default_params_match_reference_constants():
    p = _abi.default_params()
    assert p.speed_scale == 1.0 and p.null_gain == 0.5 and p.lookahead == 0.3  # vf:136, nullspace:62,121
    assert list(p.mix_w) == [1.0, 1.0, 0.0, 0.0, 0.0, 0.0]                        # bridge:596


def test_tracking_state_majority_vote():
    from vfclik_amd.vf_module import TrackingState
    ts = TrackingState()
    msgs = []
    for k in range(25):   # far from the goal and following the command
        msgs += ts.update(0.5, 30.0, 0.01, 0.01)
    assert msgs == [("xyz", "follow"), ("rot", "follow")]   # reported once, when the vote over 20 samples flips
    msgs = []
    for k in range(25):   # far and NOT following
        msgs += ts.update(0.5, 30.0, 0.5, 0.5)
    assert msgs == [("xyz", "not follow"), ("rot", "not follow")]
    msgs = []
    for k in range(25):   # arrived
        msgs += ts.update(0.001, 0.2, 0.5, 0.5)
    assert msgs == [("xyz", "on goal"), ("rot", "on goal")]


def test_profile_digest_parses_kernel_names():
    """profiles/pmc_traffic.json said "kernel": "KArgs)" -- the name was cut at the last '::'."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("profile_digest", os.path.join(root, "tools", "profile_digest.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    n1 = "void vfik::(anonymous namespace)::cycle_kernel<float, 7, false, true, false, true, 1, -1, false>(std::conditional<SmallArgs<1, false, true>::value, vfik::KLean, vfik::KArgs>::type)"
    assert m.kernel_of(n1) == "cycle_kernel<float, 7, false, true, false, true, 1, -1, false>"
    assert m.kernel_of("void vfik::(anonymous namespace)::cycle_sub8_kernel<double, 7, true>(vfik::KArgs)") == "cycle_sub8_kernel<double, 7, true>"
    n2 = "void vfik::(anonymous namespace)::cycle_kernel_s<float, 7, false, true, false, true, 1, -1, false, false, 1>(void const*, void const*, void*, int*, int, int, int, int, unsigned int, int)"
    assert m.kernel_of(n2) == "cycle_kernel_s<float, 7, false, true, false, true, 1, -1, false, false, 1>"
    import json
    t = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    assert all(v["kernel"].startswith("cycle_") for v in t.values())
