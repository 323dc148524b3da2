"""Object feeder: user-level objects -> numbered vector-field primitives.

Restates the translation of /root/reference/scripts/object_feeder:111-359: ``set goal``,
``set goalAndNormal``, ``set ObstacleP``, ``set ObstacleH`` and ``remove`` bottles on
``<base>/ofeeder/object`` become ``add`` / ``remove`` bottles on ``<base>/vectorField/param`` (and
``add``/``remove`` on ``<base>/dmonitor/objectsIn``).  Field ids, forces and parameter layouts are the
reference's: goal id 1 force +1 type 1; funnel id 2 force +30 type 5; near-goal repeller id 3 force
-10 type 2 at 5 cm; obstacles id 4+k, point force -10 type 2 (safeDist 0.001), hemisphere force -50
type 4.  Nothing is sent until a goal exists (object_feeder:214,355-359).
"""
import logging

import numpy as np

from . import ports as yarp

log = logging.getLogger("vfclik_amd.ofeeder")


def _list(b):
    return [b.get(i).asDouble() for i in range(b.size())]


class ObjectFeeder:
    def __init__(self, base):
        self.base = base
        self.objects = {}  # object 0 is always the goal, the rest are obstacles (object_feeder:89-90)
        self.object_port = yarp.BufferedPortBottle()
        self.object_port.open(base + "/ofeeder/object")
        self.object_port.setStrict(True)
        self.param_port = yarp.BufferedPortBottle()
        self.param_port.open(base + "/ofeeder/param")
        self.objects_out = yarp.BufferedPortBottle()
        self.objects_out.open(base + "/ofeeder/objectOut")     # object_feeder:68
        self.object_f_port = yarp.BufferedPortBottle()          # object_feeder:67,106-110: every /object bottle is forwarded
        self.object_f_port.open(base + "/ofeeder/objectf")
        yarp.Network.connect(base + "/ofeeder/param", base + "/vectorField/param")
        yarp.Network.connect(base + "/ofeeder/objectOut", base + "/dmonitor/objectsIn")

    def close(self):
        for p in (self.object_port, self.param_port, self.objects_out, self.object_f_port):
            p.close()

    # -- helpers ---------------------------------------------------------------------------------
    def _param_add(self, vf_id, force, vf_type, params):
        b = self.param_port.prepare()
        b.clear()
        b.addString("add")
        b.addInt(vf_id)
        b.addDouble(force)
        b.addInt(vf_type)
        lst = b.addList()
        for v in params:
            lst.addDouble(float(v))
        self.param_port.writeStrict()

    def _param_remove(self, vf_id):
        b = self.param_port.prepare()
        b.clear()
        b.addString("remove")
        b.addInt(vf_id)
        self.param_port.writeStrict()

    def _objects_add(self, num, frame16):
        b = self.objects_out.prepare()
        b.clear()
        b.addString("add")
        b.addInt(num)
        lst = b.addList()
        for v in frame16:
            lst.addDouble(float(v))
        self.objects_out.writeStrict()

    # -- one pass of the loop body (object_feeder:96-359) -------------------------------------------
    def spin_once(self):
        handled = 0
        while True:
            b = self.object_port.read(False)
            if b is None:
                return handled
            handled += 1
            self.handle(b)

    def handle(self, b):
        fb = self.object_f_port.prepare()   # object_feeder:106-110
        fb.clear()
        for i in range(b.size()):
            fb.add(b.get(i))
        self.object_f_port.writeStrict()
        if b.size() < 2:
            return
        action = b.get(0).toString()
        if action == "set":
            kind = b.get(1).toString()
            if b.size() == 3:
                params = _list(b.get(2).asList())
                if kind == "goal":  # object_feeder:122-135
                    if len(params) == 16:
                        params.append(0.03)
                    if len(params) == 17:
                        self.objects[0] = params
                    else:
                        log.warning("Wrong number of values, expected 16")
                elif kind == "goalAndNormal":  # object_feeder:136-154
                    if len(params) == 21:
                        params.append(0.03)
                    if len(params) == 22:
                        self.objects[0] = params
                    else:
                        log.warning("Wrong number of values, expected 21")
            elif b.size() == 4:
                num = b.get(2).asInt()
                params = _list(b.get(3).asList())
                if kind == "ObstacleP":  # object_feeder:156-170
                    if len(params) == 18:
                        self.objects[num + 1] = ["ObstacleP"] + params
                    else:
                        log.warning("Wrong number of values, expected 18")
                elif kind == "ObstacleH":  # object_feeder:171-186
                    if len(params) == 21:
                        self.objects[num + 1] = ["ObstacleH"] + params
                    else:
                        log.warning("Wrong number of values, expected 21")
            else:
                log.warning("Wrong number of values, expected 3 or 4")
        elif action == "remove":  # object_feeder:189-210
            num = b.get(1).asInt()
            if num + 1 in self.objects:
                del self.objects[num + 1]
                self._param_remove(5 + num)
                ob = self.objects_out.prepare()
                ob.clear()
                ob.addString("remove")
                ob.addInt(num + 1)
                self.objects_out.writeStrict()
            else:
                log.warning("Object doesn't exist, doing nothing")
                return
        else:
            log.warning("Action not recognized")
        if 0 not in self.objects:  # object_feeder:355-359
            log.info("Not setting repellers, waiting for a goal")
            return
        for num in sorted(self.objects):  # object_feeder:216-354
            p = self.objects[num]
            if num == 0:
                self._objects_add(0, p[:16])
                if len(p) == 17:  # normal goal: attractor, and drop funnel (2) + near-goal repeller (3)
                    self._param_add(1, 1.0, 1, p[:17])
                    self._param_remove(2)
                    self._param_remove(3)
                else:  # goal with approach vector (object_feeder:248-303)
                    self._param_add(1, 1.0, 1, p[:16] + [p[21]])
                    self._param_add(2, 30.0, 5, [p[3], p[7], p[11], p[16], p[17], p[18], p[19], 10.0, p[20], 2.0])
                    axis = np.array(p[16:19])
                    axis = axis / np.linalg.norm(axis)
                    vpos = np.array([p[3], p[7], p[11]]) + axis * 0.05
                    self._param_add(3, -10.0, 2, [vpos[0], vpos[1], vpos[2], p[20] + 0.05, 0.001, 5.0])
            else:
                self._objects_add(num, p[1:17])
                if p[0] == "ObstacleP":  # object_feeder:317-334
                    self._param_add(4 + num, -10.0, 2, [p[4], p[8], p[12], p[17], 0.001, p[18]])
                elif p[0] == "ObstacleH":  # object_feeder:335-354
                    self._param_add(4 + num, -50.0, 4, [p[4], p[8], p[12], p[17], p[18], p[19], p[20], p[21]])
