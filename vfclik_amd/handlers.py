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
import time
from math import pi

import numpy as np

from . import ports as yarp


def _open(name, strict=False):
    p = yarp.BufferedPortBottle()
    p.open(name)
    p.setStrict(strict)
    return p


def _doubles(b):
    return [b.get(i).asDouble() for i in range(b.size())]


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
        bottle = self.object_port.prepare()
        bottle.clear()
        bottle.addString("set")
        bottle.addString("goal")
        lst = bottle.addList()
        for i in self.cart_goal:
            lst.addDouble(i)
        lst.addDouble(self.current_slowdown_distance)
        self.object_port.writeStrict()
        self.set_cartesian_control()

    def _write_yarp_port(self, port, data, strict=True):  # handlers.py:132-145
        bottle = port.prepare()
        bottle.clear()
        for i in data:
            if type(i) == float or isinstance(i, np.floating):
                bottle.addDouble(float(i))
            elif type(i) == int:
                bottle.addInt(i)
            elif type(i) == str:
                bottle.addString(i)
        if strict:
            port.writeStrict()
        else:
            port.write()

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
            dists = self.distout_port.read(True)
            for i in range(dists.size()):
                item = dists.get(i).asList()
                if item.get(0).asInt() == 0:  # main goal
                    return [item.get(1).asDouble(), item.get(2).asDouble() * pi / 180.0]

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
        bottle = self.vf_weight_port.prepare()
        bottle.clear()
        bottle.addString("j")
        for w in joint_weights:
            bottle.addDouble(w)
        self.vf_weight_port.writeStrict()

    def set_wik_cart_weights(self, cart_weights):
        bottle = self.vf_weight_port.prepare()
        bottle.clear()
        bottle.addString("t")
        for w in cart_weights:
            bottle.addDouble(w)
        self.vf_weight_port.writeStrict()

    def set_tool(self, tool_frame):
        self._write_yarp_port(self.tool_port, [float(x) for x in tool_frame])


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
        bout = self.toolp.prepare()
        bout.clear()
        for i in toolframe:
            bout.addDouble(float(i))
        self.toolp.write(True)

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

    def gotoPos(self, pos):
        self.current_frame[3], self.current_frame[7], self.current_frame[11] = pos[0], pos[1], pos[2]
        self.sendFrame()

    def setOrient(self, orient):
        for i in range(3):
            for j in range(3):
                self.current_frame[j + 4 * i] = orient[j + i * 3]
        self.sendFrame()

    def getPose(self, blocking=True):
        pose_b = self.posep.read(blocking)
        return _doubles(pose_b) if pose_b is not None else None

    def gotoPose(self, pos, orient):
        self.current_frame[3], self.current_frame[7], self.current_frame[11] = pos[0], pos[1], pos[2]
        for i in range(3):
            for j in range(3):
                self.current_frame[j + 4 * i] = orient[j + i * 3]
        self.sendFrame()

    def gotoFrame(self, frame, wait=10.0, goal_precision=[], spin=None):
        """frame: 16 values; wait in seconds; goal_precision [trans, rot] (handlers.py:346-387).
        ``spin``: callable run while waiting (the in-process substitute for the other processes)."""
        for i in range(len(frame)):
            self.current_frame[i] = frame[i]
        self.sendFrame()
        init_time = cur_time = time.time()
        difference = np.array([0.0, 0.0])
        result = False
        while self.goaldistp.getPendingReads():
            self.goaldistp.read(False)
        if len(goal_precision) == 2 and wait > 0.0:
            first_read = True
            while cur_time - init_time < wait:
                if spin is not None:
                    spin()
                b = self.goaldistp.read(False)
                if b and not first_read:
                    for i in range(b.size()):
                        line = b.get(i).asList()
                        if line.get(0).asInt() == 0:
                            pos_dist = line.get(1).asDouble()
                            orient_dist = line.get(2).asDouble() * pi / 180.0
                    difference = np.array([pos_dist, orient_dist])
                    if pos_dist < goal_precision[0] and orient_dist < goal_precision[1]:
                        result = True
                        break
                first_read = False
                if spin is None:
                    time.sleep(0.01)
                cur_time = time.time()
        return (result, difference)

    def gotThere(self):
        b = self.goaldistp.read(True)
        dist = 1000.0
        if b:
            for i in range(b.size()):
                line = b.get(i).asList()
                if line.get(0).asInt() == 0:
                    dist = line.get(1).asDouble()
            if dist < self.goal_threshold:
                return True
        return False

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
        bout = self.outp.prepare()
        bout.clear()
        for v in vals:
            bout.addInt(v)
        self.outp.write(True)

    def joint_controller(self):
        self._weights([0, 0, 1, 0])

    def cartesian_controller(self):
        self._weights([1, 1, 0, 0])

    def torso_joints(self, cjoints):
        if self.torso:
            bout = self.torso_port.prepare()
            bout.clear()
            for i in cjoints:
                bout.addInt(i)
            self.torso_port.write(True)
        else:
            print("There's no torso")

    def set_VFW(self, type_of="joint", weights=[1] * 7):  # back compatibility
        print("deprecated, use set_weights instead")
        self.set_weights(type_of, weights)

    def set_weights(self, type_of="joint", weights=[1] * 7):
        bout = self.VFW_port.prepare()
        bout.clear()
        bout.addString("t" if type_of == "task" else "j")
        for w in weights:
            bout.addDouble(w)
        self.VFW_port.write(True)


class HandleJController(object):
    def __init__(self, arm_portbasename, handlername="HandlerArmJoint"):
        prename = arm_portbasename
        full_name = prename + "/" + handlername
        self.outp = _open(full_name + "/to_js")
        self.inp = _open(full_name + "/q")
        yarp.Network.connect(full_name + "/to_js", prename + "/jpctrl/ref")
        yarp.Network.connect(prename + "/bridge/encoders", full_name + "/q")

    def set_ref_js(self, js, wait=0.0, goal_precision=[], spin=None):  # handlers.py:544-576
        bout = self.outp.prepare()
        bout.clear()
        for i in js:
            bout.addDouble(i)
        self.outp.write(True)
        init_time = cur_time = time.time()
        js = np.asarray(js, dtype=float)
        difference = np.array([0.0] * len(js))
        result = False
        if len(goal_precision) == len(js) and wait != 0.0:
            gp = np.array(goal_precision)
            while (cur_time - init_time < wait) or wait == -1:
                if spin is not None:
                    spin()
                b = self.inp.read(False)
                if b:
                    q = np.array(_doubles(b))
                    difference = js - q
                    if (((js - gp) <= q) * ((js + gp) >= q)).all():
                        result = True
                        break
                if spin is None:
                    time.sleep(0.01)
                cur_time = time.time()
        return (result, difference)


def main():
    return False
