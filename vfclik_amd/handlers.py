"""Client handler API of the reference (/root/reference/src/handlers.py), on the in-process ports.

Same classes, method names, port names and bottle layouts as the reference, so user programs written
against ``vfclik.handlers`` drive a batched :class:`vfclik_amd.vf_module.ControlCycleBatch` unchanged:
``HandleArmNew`` (handlers.py:32-230), ``HandleArm`` (:232-440), ``HandleBridge`` (:443-522),
``HandleJController`` (:525-576).  Wire conventions kept: goal = ("set", "goal", (frame16 + slow-down))
(handlers.py:118-128,297-307); mixer weights [cart, null, joint, 0] (handlers.py:189-204,481-497);
rotation distance arrives in degrees and is converted with pi/180 (handlers.py:175,376).

Differences, all forced by the missing transport: connections are made by name in this process and
never block (``yarp_connect_blocking`` -> ``ports.Network.connect``); methods that are empty stubs in the
reference (``set_vf_tool``, ``set_stiffness`` of HandleArmNew, ``go_xyz``, ``go_rot``, ``get_joint_angles``)
are empty here too.
"""
import logging
import time
from math import pi

import numpy as np

from . import ports as yarp

log = logging.getLogger("vfclik_amd.handlers")


def _open(name, strict=False):
    p = yarp.BufferedPortBottle()
    p.open(name)
    p.setStrict(strict)
    return p


def _doubles(b):
    return [b.get(i).asDouble() for i in range(b.size())]


def _emit(port, items, strict=True):
    """One bottle: floats -> doubles, ints -> ints, str -> strings, list/tuple -> nested list of doubles."""
    b = port.prepare()
    b.clear()
    for it in items:
        if isinstance(it, (list, tuple)):
            sub = b.addList()
            for v in it:
                sub.addDouble(float(v))
        elif isinstance(it, str):
            b.addString(it)
        elif isinstance(it, (int, np.integer)) and not isinstance(it, bool):
            b.addInt(int(it))
        else:
            b.addDouble(float(it))
    port.write(strict)


def _goal_entry(bottle, key=0):
    """(xyz distance, rotation distance in rad) of object `key` in a /dmonitor/distOut bottle, else None."""
    for i in range(bottle.size()):
        line = bottle.get(i).asList()
        if line is not None and line.get(0).asInt() == key:
            return line.get(1).asDouble(), line.get(2).asDouble() * pi / 180.0
    return None


class HandleArmNew:
    def __init__(self, namespace="/0", module_name="/handle_arm", arm_namespace="/0", robot="/lwr", arm="/right", sim=True):
        self.sim = True
        self.module_name = module_name
        self.namespace = namespace
        self.arm_namespace = arm_namespace
        base = arm_namespace + robot + arm
        mine = namespace + module_name + arm
        self.base = base
        # (my port, remote port, direction) -- names of handlers.py:38-58
        wiring = {
            "object": (mine + "/object", base + "/ofeeder/object", "out"),
            "stiffness": (mine + "/stiffness", base + "/robot/stiffness", "out"),
            "pose": (mine + "/pose", base + "/vectorField/pose", "in"),
            "distout": (mine + "/distOut", base + "/dmonitor/distOut", "in"),
            "tool": (mine + "/tool", base + "/vectorField/tool", "out"),
            "bridge_weight": (mine + "/bridge/weight", base + "/bridge/weight", "out"),
            "vf_weight": (mine + "/vectorField/weight", base + "/vectorField/weight", "out"),
            "bridge_encoders": (mine + "/encoders", base + "/bridge/encoders", "in"),
            "joint_ref": (mine + "/joint_ref", base + "/jpctrl/ref", "out"),
            "joint_sim_qin": (mine + "/joint_sim/qin", base + "/joint_sim/qin", "out"),
        }
        for key, (local, remote, direction) in wiring.items():
            port = _open(local)
            setattr(self, key + "_port", port)
            setattr(self, key + "_port_name", local)
            if direction == "out":
                yarp.Network.connect(local, remote)
            else:
                yarp.Network.connect(remote, local)
        self.current_slowdown_distance = 0.1  # handlers.py:107

    def set_sim_arm_q(self, q):
        self._write_yarp_port(self.joint_sim_qin_port, q, strict=True)

    def set_vf_tool(self, toolframe):
        pass

    def set_stiffness(self, stiffness):
        pass

    def go_cart(self, frame):  # handlers.py:118-129
        self.cart_goal = frame
        _emit(self.object_port, ["set", "goal", list(frame) + [self.current_slowdown_distance]])
        self.set_cartesian_control()

    def _write_yarp_port(self, port, data, strict=True):  # handlers.py:132-145
        """The reference adds a value only when its type is EXACTLY float, int or str (`type(i)==float`, handlers.py:136-141):
        NumPy scalars, bools and None are dropped without a word, so `go_joint(numpy_array)` sends an EMPTY bottle there
        (which /jpctrl/ref ignores).  Reproduced value for value and type for type (tests/golden/handlers_wire.json); the
        only addition is the warning."""
        kept = [d for d in data if type(d) in (float, int, str)]
        if len(kept) != len(data):
            log.warning("%d of %d values are not plain float / int / str and were dropped (handlers.py:136-141)",
                        len(data) - len(kept), len(data))
        _emit(port, kept, strict)

    def go_joint(self, angles):
        self.joint_goal = angles
        self._write_yarp_port(self.joint_ref_port, angles)
        self.set_joint_control()

    def go_xyz(self, xyz):
        pass

    def go_rot(self, rot):
        pass

    def get_cart_pose(self):
        bottle = self.pose_port.read(True)
        return _doubles(bottle)

    def get_dist_cart_goal(self):  # handlers.py:163-177
        while True:
            entry = _goal_entry(self.distout_port.read(True))  # object 0 = main goal
            if entry is not None:
                return list(entry)

    def get_dist_joint_goal(self):
        bottle = self.bridge_encoders_port.read(True)
        cur = _doubles(bottle)
        return [i - j for i, j in zip(self.joint_goal, cur)]

    def get_joint_angles(self):
        pass

    def set_controller_mixer(self, cart=True, joint=False, null=False):  # handlers.py:189-204
        data = [1 if cart else 0, 1 if null else 0, 1 if joint else 0, 0]
        self._write_yarp_port(self.bridge_weight_port, data)

    def set_cartesian_control(self):
        self.set_controller_mixer(cart=True, null=True)

    def set_joint_control(self):
        self.set_controller_mixer(cart=False, null=False, joint=True)

    def set_wik_joint_weights(self, joint_weights):
        _emit(self.vf_weight_port, ["j"] + [float(w) for w in joint_weights])

    def set_wik_cart_weights(self, cart_weights):
        _emit(self.vf_weight_port, ["t"] + [float(w) for w in cart_weights])

    def set_tool(self, tool_frame):  # handlers.py:229-230: through the type filter, ints stay ints
        self._write_yarp_port(self.tool_port, tool_frame)


class HandleArm(object):
    def __init__(self, arm_portbasename, namespace="", handlername="/HandlerArm"):
        prename = namespace + arm_portbasename
        full_name = prename + handlername
        self.outp = _open(full_name + "/toObjectFeeder")
        self.stiffness_port = _open(full_name + "/stiffness")
        self.goaldistp = _open(full_name + "/fromGoalDistance")
        self.posep = _open(full_name + "/pose:i")
        yarp.Network.connect(full_name + "/toObjectFeeder", prename + "/ofeeder/object")
        yarp.Network.connect(full_name + "/stiffness", prename + "/robot/stiffness")
        yarp.Network.connect(prename + "/dmonitor/distOut", full_name + "/fromGoalDistance")
        yarp.Network.connect(prename + "/vectorField/pose", full_name + "/pose:i")
        self.current_frame = [1.0, 0.0, 0.0, 0.0, 0.0, 0.1, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]  # handlers.py:260-263
        self.current_slowdown_distance = 0.1
        self.toolp = _open(full_name + "/toToolin", strict=True)
        yarp.Network.connect(full_name + "/toToolin", "%s/vectorField/tool" % prename)
        self.goal_threshold = 0.01

    def setTool(self, toolframe):
        """toolframe: 16 values (the reference takes a PyKDL Frame and flattens it, handlers.py:276-288)."""
        _emit(self.toolp, [float(v) for v in toolframe])

    def set_stiffness(self, stiffness):
        bottle = self.stiffness_port.prepare()
        bottle.clear()
        for i in stiffness:
            bottle.addDouble(i)
        self.stiffness_port.write(True)

    def sendFrame(self):  # handlers.py:297-307
        bout = self.outp.prepare()
        bout.clear()
        bout.addString("set")
        bout.addString("goal")
        frame_list = bout.addList()
        for i in self.current_frame:
            frame_list.addDouble(i)
        frame_list.addDouble(self.current_slowdown_distance)
        self.outp.write(True)

    def _place(self, pos=None, orient=None):
        if pos is not None:
            self.current_frame[3:12:4] = [float(pos[0]), float(pos[1]), float(pos[2])]
        if orient is not None:
            for row in range(3):
                self.current_frame[4 * row:4 * row + 3] = [float(v) for v in orient[3 * row:3 * row + 3]]
        self.sendFrame()

    def gotoPos(self, pos):
        self._place(pos=pos)

    def setOrient(self, orient):
        self._place(orient=orient)

    def getPose(self, blocking=True):
        pose_b = self.posep.read(blocking)
        return _doubles(pose_b) if pose_b is not None else None

    def gotoPose(self, pos, orient):
        self._place(pos=pos, orient=orient)

    def gotoFrame(self, frame, wait=10.0, goal_precision=[], spin=None):
        """frame: 16 values; wait in seconds; goal_precision [trans, rot] (handlers.py:346-387).
        ``spin``: callable run where the reference sleeps 10 ms (the in-process substitute for the other processes).
        As in the reference, whatever the FIRST poll finds is discarded -- it is the report that was already waiting when the
        goal went out (the reference counts the pending reads but does not read them, handlers.py:359-362) -- and `difference`
        is the last report's."""
        self.current_frame[:len(frame)] = [v for v in frame]
        self.sendFrame()
        start = time.time()
        difference, result = np.array([0.0, 0.0]), False
        if len(goal_precision) == 2 and wait > 0.0:
            first_read = True
            while time.time() - start < wait:
                b = self.goaldistp.read(False)
                entry = _goal_entry(b) if (b is not None and not first_read) else None
                if entry is not None:
                    difference = np.array(entry)
                    if entry[0] < goal_precision[0] and entry[1] < goal_precision[1]:
                        result = True
                        break
                first_read = False
                if spin is not None:
                    spin()
                else:
                    time.sleep(0.01)
        return (result, difference)

    def gotThere(self):
        entry = _goal_entry(self.goaldistp.read(True))
        return entry is not None and entry[0] < self.goal_threshold

    def gotoPosBlocking(self, pos, timeout=20):
        self.gotoPos(pos)
        start = time.time()
        for i in range(10):  # ignore the first reports
            self.gotThere()
        while (time.time() - start) < timeout:
            if self.gotThere():
                return True
        return False

    gotoPosBlockingGrasp = gotoPosBlocking  # identical bodies in the reference (handlers.py:404-440)


class HandleBridge(object):
    def __init__(self, arm_portbasename, handlername="HandlerArmBridge", torso=True):
        self.torso = torso
        prename = arm_portbasename
        full_name = prename + "/" + handlername
        self.outp = _open(full_name + "/toBridge_weights")
        if self.torso:
            self.torso_port = _open(full_name + "/to_torso_cjoints")
            yarp.Network.connect(full_name + "/to_torso_cjoints", prename + "/bridge/torso_cjoints:i")
        self.VFW_port = _open(full_name + "/to_VF_weight:o")
        self.encoders_port = _open(full_name + "/encoders:i")
        # the reference connects to "/bridge/weights" (handlers.py:472) although the bridge opens
        # "/bridge/weight" (bridge:570); both names are wired so either spelling reaches the mixer
        yarp.Network.connect(full_name + "/toBridge_weights", prename + "/bridge/weights")
        yarp.Network.connect(full_name + "/toBridge_weights", prename + "/bridge/weight")
        yarp.Network.connect(full_name + "/to_VF_weight:o", prename + "/vectorField/weight")
        yarp.Network.connect(prename + "/bridge/encoders", full_name + "/encoders:i")

    def read_joint_angles(self):
        return _doubles(self.encoders_port.read())

    def _weights(self, vals):
        _emit(self.outp, [int(v) for v in vals])

    def joint_controller(self):
        self._weights([0, 0, 1, 0])

    def cartesian_controller(self):
        self._weights([1, 1, 0, 0])

    def torso_joints(self, cjoints):
        if not self.torso:
            print("There's no torso")
            return
        _emit(self.torso_port, [int(j) for j in cjoints])

    def set_VFW(self, type_of="joint", weights=[1] * 7):  # back compatibility
        print("deprecated, use set_weights instead")
        self.set_weights(type_of, weights)

    def set_weights(self, type_of="joint", weights=[1] * 7):
        _emit(self.VFW_port, ["t" if type_of == "task" else "j"] + [float(w) for w in weights])


class HandleJController(object):
    def __init__(self, arm_portbasename, handlername="HandlerArmJoint"):
        prename = arm_portbasename
        full_name = prename + "/" + handlername
        self.outp = _open(full_name + "/to_js")
        self.inp = _open(full_name + "/q")
        yarp.Network.connect(full_name + "/to_js", prename + "/jpctrl/ref")
        yarp.Network.connect(prename + "/bridge/encoders", full_name + "/q")

    def set_ref_js(self, js, wait=0.0, goal_precision=[], spin=None):  # handlers.py:544-576
        _emit(self.outp, [float(v) for v in js])
        start = time.time()
        difference, result = np.array([0.0] * len(js)), False
        if len(goal_precision) == len(js) and wait != 0.0:
            tol = np.array(goal_precision)
            while wait == -1 or time.time() - start < wait:  # wait == -1: until reached (handlers.py:559)
                b = self.inp.read(False)
                if b:
                    q = np.array(_doubles(b))
                    difference = js - q
                    if (((js - tol) <= q) * ((js + tol) >= q)).all():
                        result = True
                        break
                if spin is not None:
                    spin()
                else:
                    time.sleep(0.01)
        return (result, difference)


def main():
    return False
