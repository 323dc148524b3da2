"""The per-cycle control modules of vfclik for B arms behind the reference's port interface.

One :class:`ControlCycleBatch` stands where the reference runs, PER ARM, a ``vf`` process
(/root/reference/scripts/vf), a ``nullspace`` process (scripts/nullspace), a ``debug_jointlimits``
process and the ``CommandMixer`` inside ``bridge`` (scripts/bridge:593-596,626).  It opens the same
ports under each arm's base name (``<namespace><robotarm_portbasename>``, e.g. ``/0/lwr/right``):

    <base>/vectorField/{qIn,qdotOut,param,tool,weight,max_vel,pose,pose_no_tool,pose_in,vector_out}  (vf:69-84)
    <base>/nullspace/{qin,control,qdotout}                                                           (nullspace:142-144)
    <base>/debug/{qin,qdist}                                                                         (debug_jointlimits:46-47)
    <base>/bridge/{encoders,weight,current_weights,jointcmd,mechanismcmd,xtra1cmd,xtra2cmd,mixed}    (bridge:564-573)

:meth:`cycle` does what one iteration of each of those loops does -- poll every input non-blocking,
ignore malformed bottles with a warning, keep sticky state -- and then runs ONE ``vfik_step`` for
the whole batch.  Arms that received no joint angles this cycle publish nothing AND keep their state
(vf:312-313, nullspace:162-163): the launch carries a fresh-q gate (``io.active``), so a silent arm's
nullspace sign memory and its tracking-error history (vf:350-356) do not advance.

Port I/O is per arm and in Python, so this layer is for drop-in use and tests; a caller that already
holds batched arrays uses :meth:`step_arrays` (or ``Engine`` directly) and never touches a bottle.
"""
import logging
import time

import numpy as np

from . import _abi
from . import ports as yarp
from .engine import Engine
from .fields import FieldSets

log = logging.getLogger("vfclik_amd.vf")

CONFIG_MAX_VEL = 0.41  # vf:134
MIX_PORTS = ("vectorfieldcmd", "nullcmd", "jointcmd", "mechanismcmd", "xtra1cmd", "xtra2cmd")  # bridge:593-596


def _bottle_doubles(b):
    return [b.get(i).asDouble() for i in range(b.size())]


def _send(port, values, ints=()):
    b = port.prepare()
    b.clear()
    for v in values:
        b.addDouble(float(v))
    for v in ints:
        b.addInt(int(v))
    port.write()


class TrackingState:
    """Majority vote over the last 20 samples of scripts/monitor_distance (monitor_distance:86-105,
    173-219): 'on goal' / 'follow' / 'not follow' for the xyz and the rotational part, reported only
    when the voted state changes."""
    distanceXYZ_th, track_error_xyz_th = 0.02, 0.1        # monitor_distance:91-94
    distanceOrient_th, track_error_rot_th = 1.0, 0.1      # monitor_distance:95-99 (degrees, radians)
    size = 20                                             # monitor_distance:102
    states = ("on goal", "follow", "not follow")

    def __init__(self):
        self.buffer = []
        self.last = ["on goal", "on goal"]

    @staticmethod
    def _classify(dist, dist_th, err, err_th):
        state = "on goal"
        if dist > dist_th and err > err_th:
            state = "not follow"
        if dist > dist_th and err < err_th:
            state = "follow"
        if dist < dist_th:
            state = "on goal"
        return state

    def update(self, dist_xyz, dist_rot_deg, err_xyz, err_rot):
        """Returns the list of (kind, state) messages to emit this sample (usually empty)."""
        self.buffer.append((self._classify(dist_xyz, self.distanceXYZ_th, err_xyz, self.track_error_xyz_th),
                            self._classify(dist_rot_deg, self.distanceOrient_th, err_rot, self.track_error_rot_th)))
        out = []
        if len(self.buffer) > self.size:
            self.buffer.pop(0)
            for k, kind in enumerate(("xyz", "rot")):
                col = [b[k] for b in self.buffer]
                voted = max(self.states, key=col.count)
                if voted != self.last[k]:
                    self.last[k] = voted
                    out.append((kind, voted))
        return out


class ControlCycleBatch:
    def __init__(self, chain, arm_bases, io_dtype=np.float64, max_fields=16, device=0, nullspace=True,
                 mixer=True, guard_time=2.0, params=None, open_ports=True, clock=time.time, initial_joint_pos=None,
                 limits_fn=None):
        """initial_joint_pos: `config.initial_joint_pos` -- with it the joint P controller runs for every arm from
        the first cycle, commanding kp * (initial_joint_pos - q) until a /jpctrl/ref arrives, as the reference's
        does (joint_p_controller:101); without it an arm has a controller once it got a reference, and until then
        its mixer channel 2 is whatever arrives on /bridge/jointcmd.
        limits_fn: `config.updateJntLimits` / `rob.get_limits()` for robots whose limits depend on the pose
        (joint_p_controller:80,121-125; nullspace:167): called every cycle with q (B, n), returns (lo, hi), each
        (B, n); None = the chain's static limits."""
        self.chain = chain
        self.bases = list(arm_bases)
        self.B = len(self.bases)
        self.n = chain.n
        flags = (_abi.F_NULLSPACE if nullspace else 0) | (_abi.F_MIXER if mixer else 0)
        self.params = params if params is not None else _abi.default_params(flags=flags)
        self.engine = Engine(chain, self.B, io_dtype=io_dtype, max_slots=3 * max_fields, device=device, params=self.params)
        self.fields = FieldSets(self.B, max_fields)
        self.clock = clock
        self.guard_time = guard_time
        self.config_max_vel = float(self.params.max_vel)  # bridge:608: config.max_vel bounds the runtime value
        self.speed = np.full(self.B, float(self.params.speed_scale))
        self.tools = np.tile(np.eye(4).reshape(16), (self.B, 1))  # vf:154
        self._tools_dirty = False
        self.q = np.zeros((self.B, self.n))
        self.control = np.zeros((self.B, _abi.NULL_CONTROLS))  # nullspace:137
        self.mix_w = np.tile(np.array(list(self.params.mix_w)), (self.B, 1))  # bridge:596
        self._mix_dirty = False
        self.ext = np.zeros((4, self.B, self.n))
        self.ext_time = np.full((4, self.B), self.clock())
        self._ext_dirty = [False] * 4
        self.objects = [dict() for _ in range(self.B)]   # monitor_distance:72: id -> frame16, insertion-ordered
        self._objects_key = None
        self._probe_bufs = None
        self.q_ref = np.zeros((self.B, self.n))          # /jpctrl/ref (joint_p_controller:113-118)
        self.has_ref = np.zeros(self.B, dtype=bool)      # arms whose joint controller has a reference
        if initial_joint_pos is not None:                 # joint_p_controller:101: ref = config.initial_joint_pos
            self.q_ref[:] = np.asarray(initial_joint_pos, dtype=float).reshape(1, self.n)
            self.has_ref[:] = True
        self.limits_fn = limits_fn
        self._outs = None                                 # output arrays, reused so that gated arms keep their rows
        self.report_counter = 0  # vf:185,432-435
        self.tracking = [TrackingState() for _ in range(self.B)]
        self.last = {}
        self.ports = []
        if open_ports:
            self._open_ports()

    # -- ports ------------------------------------------------------------------------------------
    def _open_ports(self):
        def mk(name, strict=False):
            p = yarp.BufferedPortBottle()
            p.open(name)
            p.setStrict(strict)
            return p

        for base in self.bases:
            vf, ns, dbg, br = base + "/vectorField", base + "/nullspace", base + "/debug", base + "/bridge"
            d = {
                "qIn": mk(vf + "/qIn"), "qdotOut": mk(vf + "/qdotOut"), "param": mk(vf + "/param", True),
                "tool": mk(vf + "/tool", True), "weight": mk(vf + "/weight", True), "max_vel": mk(vf + "/max_vel", True),
                "pose": mk(vf + "/pose"), "pose_no_tool": mk(vf + "/pose_no_tool"), "pose_in": mk(vf + "/pose_in"),
                "vector_out": mk(vf + "/vector_out"), "goal_out": mk(vf + "/goal_out"),
                "ns_qin": mk(ns + "/qin"), "ns_control": mk(ns + "/control"), "ns_qdotout": mk(ns + "/qdotout"),
                "dbg_qin": mk(dbg + "/qin"), "qdist": mk(dbg + "/qdist"),
                "encoders": mk(br + "/encoders"), "br_weight": mk(br + "/weight", True), "br_max_vel": mk(br + "/max_vel", True),
                "current_weights": mk(br + "/current_weights"), "mixed": mk(br + "/mixed"),
                "track_error": mk(vf + "/track_error"), "distOut": mk(base + "/dmonitor/distOut"),
                "tracking_state": mk(base + "/dmonitor/tracking_state"),
                "objectsIn": mk(base + "/dmonitor/objectsIn", True),
                "jp_ref": mk(base + "/jpctrl/ref"), "jp_at_goal": mk(base + "/jpctrl/at_goal"),
            }
            for k in MIX_PORTS[2:]:
                d[k] = mk(br + "/" + k)
            self.ports.append(d)
            # the wiring the reference modules make themselves (vf:129-131, bridge:578-582, nullspace:153-154)
            yarp.Network.connect(br + "/encoders", vf + "/qIn")
            yarp.Network.connect(br + "/encoders", ns + "/qin")
            yarp.Network.connect(br + "/encoders", dbg + "/qin")

    def close(self):
        for d in self.ports:
            for p in d.values():
                p.close()
        self.ports = []
        self.engine.close()

    # -- the polling half of the reference loops ----------------------------------------------------
    def _poll(self):
        got_q = np.zeros(self.B, dtype=bool)
        now = self.clock()
        for a, d in enumerate(self.ports):
            for b in self._drain(d["max_vel"]):  # vf:197-207
                v = b.get(0).asDouble()
                if 0.0 <= v <= CONFIG_MAX_VEL:
                    self.speed[a] = v
                    self.engine.set_speed_scale([v], first_arm=a)
                else:
                    log.warning("arm %d: speedScale not between 0.0 and config_max_vel, ignoring", a)
            while True:  # strict port: every /param message counts (vf:210-275)
                b = d["param"].read(False)
                if b is None:
                    break
                self.fields.handle_param(a, b)
            for b in self._drain(d["weight"]):  # vf:296-309
                if b.size() >= 1:
                    self._handle_weight(a, b)
            for b in self._drain(d["tool"]):  # vf:321-326: sticky, 16 values or ignored
                if b.size() == 16:
                    self.tools[a] = _bottle_doubles(b)
                    self._tools_dirty = True
            b = d["ns_control"].read(False)  # nullspace:169-173
            if b is not None:
                vals = _bottle_doubles(b)[: _abi.NULL_CONTROLS]
                self.control[a, :] = 0.0
                self.control[a, : len(vals)] = vals
            for b in self._drain(d["br_weight"]):  # command_mixer.py:48-53
                for i in range(min(b.size(), _abi.MIX_CHANNELS)):
                    self.mix_w[a, i] = b.get(i).asDouble()
                self._mix_dirty = True
            for b in self._drain(d["br_max_vel"]):  # bridge:612-623: the limiter's joint speed
                v = b.get(0).asDouble()
                if 0.0 <= v <= self.config_max_vel:
                    self.engine.set_max_vel([v], first_arm=a)
                else:
                    log.warning("arm %d: value of max_vel not between 0.0 and config.max_vel, ignoring", a)
            for ch, name in enumerate(MIX_PORTS[2:]):  # command_mixer.py:56-69
                b = d[name].read(False)
                if b and b.size() == self.n:
                    self.ext[ch, a] = _bottle_doubles(b)
                    self.ext_time[ch, a] = now
                    self._ext_dirty[ch] = True
                elif now - self.ext_time[ch, a] > self.guard_time:
                    if self.ext[ch, a].any():
                        self.ext[ch, a] = 0.0
                        self._ext_dirty[ch] = True
                elif b:
                    log.warning("arm %d: wrong length for data bottle on %s", a, name)
            for b in self._drain(d["objectsIn"]):  # monitor_distance:111-129
                if b.size() >= 2:
                    if b.get(0).toString() == "add" and b.size() == 3:
                        lst = b.get(2).asList()
                        if lst is not None and lst.size() == 16:
                            self.objects[a][b.get(1).asInt()] = [lst.get(i).asDouble() for i in range(16)]
                    if b.get(0).toString() == "remove":
                        self.objects[a].pop(b.get(1).asInt(), None)
            b = d["jp_ref"].read(False)  # joint_p_controller:113-118
            if b and b.size() == self.n:
                self.q_ref[a] = _bottle_doubles(b)
                self.has_ref[a] = True
            b = d["qIn"].read(False)  # vf:312-313
            d["ns_qin"].read(False)
            d["dbg_qin"].read(False)
            if b and b.size() == self.n:
                self.q[a] = _bottle_doubles(b)
                got_q[a] = True
        return got_q

    @staticmethod
    def _drain(port):
        """All pending bottles of a strict (configuration) port, oldest first.  The reference reads one
        per loop iteration at ~1 kHz and so reaches the same final state a few iterations later."""
        out = []
        while True:
            b = port.read(False)
            if b is None:
                return out
            out.append(b)

    def _handle_weight(self, arm, b):
        """'t' + 6 or 'j' + n doubles (vf:164-179,296-309): the weights of THIS arm's vf process."""
        kind = b.get(0).asString()
        n_vars = 6 if kind == "t" else self.n if kind == "j" else None
        if n_vars is None:
            return
        if b.size() != n_vars + 1:
            log.warning("arm %d: wrong size of %s weights, ignored", arm, kind)
            return
        w = [[b.get(i + 1).asDouble() for i in range(n_vars)]]
        self.engine.set_arm_weights(first_arm=arm, **({"wy": w} if kind == "t" else {"wq": w}))

    def _push_state(self):
        self.fields.flush(self.engine)
        if self._tools_dirty:
            self.engine.set_tool(self.tools, per_arm=True)
            self._tools_dirty = False
        if self._mix_dirty:
            self.engine.set_mixer_weights(self.mix_w)
            self._mix_dirty = False
        for ch in range(4):
            if self._ext_dirty[ch]:
                self.engine.set_ext_cmd(2 + ch, self.ext[ch])
                self._ext_dirty[ch] = False

    # -- one control cycle ---------------------------------------------------------------------------
    def cycle(self):
        got_q = self._poll()
        self._push_state()
        if not got_q.any():
            self.probe()
            return got_q
        # the joint P controller feeds /bridge/jointcmd (joint_p_controller:78): for arms with a reference the fused
        # controller IS mixer channel 2; a NaN row tells the kernel that the arm has none, and its channel 2 stays
        # the external command that arrived on /bridge/jointcmd
        ref = None
        if self.has_ref.any():
            ref = np.where(self.has_ref[:, None], self.q_ref, np.nan)
        lo = hi = None
        if self.limits_fn is not None:  # limits of THIS cycle (nullspace:167, joint_p_controller:80)
            lo, hi = self.limits_fn(self.q)
        want = ("qdot_vf", "qdot_null", "qdot_out", "pose", "pose_nt", "v6", "qdist", "status", "goal_dist", "track_error") \
            + (("q_ref_out",) if ref is not None else ())
        # the distance monitor's object frames live on the device and are rewritten only when /dmonitor/objectsIn changed them
        n_obj = self._push_objects()
        if n_obj:
            want += ("obj_dist",)
        # only the arms whose q arrived run their cycle: the others keep their state and publish nothing.
        # ONE call = copies in, the cycle kernel, the tracking-error estimator (vf:349-428) and the distance monitor
        # (monitor_distance:148-167) on the cycle's own device results, copies out, ONE synchronisation.
        # self._outs: PRIVATE persistent arrays the launch writes into, so that the rows of gated arms keep what their last
        # own cycle left there; what is published (`out`, self.last) are copies -- a consumer that holds last cycle's
        # dictionary (a logger, a previous-against-current comparison) never sees it change underneath
        if self._outs is not None and "obj_dist" in self._outs and (not n_obj or self._outs["obj_dist"].shape[1] != n_obj):
            del self._outs["obj_dist"]
        res = self.engine.step_host(self.q, null_control=self.control, q_ref=ref, want=want, active=got_q, q_lo=lo, q_hi=hi,
                                    into=self._outs)
        if self._outs is None:
            self._outs = {}
        self._outs.update(res)   # (q_ref_out joins once some arm has a reference; its key then stays)
        out = {k: v.copy() for k, v in res.items()}
        if ref is not None:  # the controller keeps the CLAMPED reference (joint_p_controller:121)
            upd = got_q & self.has_ref
            self.q_ref[upd] = out["q_ref_out"][upd]
        else:
            out["q_ref_out"] = None  # (the key set of `last` does not vary between cycles)
        out["track_error"] = out["track_error"].astype(np.float64)
        dists = out["object_dist"] = out.pop("obj_dist").astype(np.float64) if n_obj else None
        self.last = out
        self.report_counter += 1
        report = self.report_counter > 20  # vf:432-435
        if report:
            self.report_counter = 0
        for a in np.nonzero(got_q)[0]:
            d = self.ports[a]
            _send(d["pose"], out["pose"][a])                 # vf:341
            _send(d["pose_no_tool"], out["pose_nt"][a])      # vf:342
            _send(d["qdotOut"], out["qdot_vf"][a])           # vf:462-466
            _send(d["ns_qdotout"], out["qdot_null"][a])      # nullspace:180-184
            _send(d["qdist"], 100.0 * out["qdist"][a])       # debug_jointlimits:68-73
            _send(d["mixed"], out["qdot_out"][a])            # what bridge.set_vel receives (bridge:626)
            _send(d["current_weights"], self.mix_w[a])       # bridge:627
            if self.has_ref[a]:                               # joint_p_controller:139-146
                _send(d["jp_at_goal"], [], ints=[1 if out["status"][a] & _abi.ST_JOINT_AT_GOAL else 0])
            if report:
                _send(d["vector_out"], out["v6"][a])         # vf:437-442
                if 1 in self.fields.sets[a]:                  # vf:444-453: the goal's parameters
                    _send(d["goal_out"], self.fields.sets[a][1][2])
            te = out["track_error"][a]
            if te.any():                                      # from the 6th frame on (vf:354,418-428)
                _send(d["track_error"], te[:7], ints=[int(te[7])])
            if self.objects[a]:                               # monitor_distance:156-172: one entry per object
                b = d["distOut"].prepare()
                b.clear()
                for slot, oid in enumerate(self.objects[a]):
                    item = b.addList()
                    item.addDouble(float(oid))                # monitor_distance:165 sends the id as a double
                    item.addDouble(float(dists[a, slot, 0]))
                    item.addDouble(float(dists[a, slot, 1]))
                d["distOut"].write()
                if 0 in self.objects[a]:                      # the goal: tracking state vote (monitor_distance:168-219)
                    slot = list(self.objects[a]).index(0)
                    for kind, state in self.tracking[a].update(dists[a, slot, 0], dists[a, slot, 1], te[0], te[1]):
                        sb = d["tracking_state"].prepare()
                        sb.clear()
                        sb.addString(kind)
                        sb.addString(state)
                        d["tracking_state"].writeStrict()
        self.probe()
        return got_q

    def _push_objects(self):
        """The distance monitor's object dictionary (monitor_distance:72,111-129) as device state: [B][O][16] with the objects
        of every arm in the order of its dictionary, uploaded only when a message changed it.  Returns O (0: no arm knows
        any object)."""
        O = max(len(o) for o in self.objects)
        if O == 0:
            return 0
        key = [tuple((oid, tuple(f)) for oid, f in objs.items()) for objs in self.objects]
        if key != self._objects_key:
            frames = np.tile(np.eye(4).reshape(16), (self.B, O, 1))
            for a, objs in enumerate(self.objects):
                for slot, oid in enumerate(objs):
                    frames[a, slot] = objs[oid]
            self.engine.set_objects(frames)
            self._objects_key = key
        return O

    def probe(self):
        """The visualisation probe of scripts/vf (vf:469-503): for every arm with a 16-value bottle waiting on
        /pose_in, the arm's field at that pose goes out on /vector_out.  Runs outside the control cycle, as
        the reference's probe runs whether or not joint angles arrived."""
        asked = {}
        for a, d in enumerate(self.ports):
            b = d["pose_in"].read(False)
            if b is not None and b.size() == 16:
                asked[a] = _bottle_doubles(b)
        if not asked:
            return asked
        self._push_state()
        e = self.engine
        esz = e.io_dtype.itemsize
        if self._probe_bufs is None:
            self._probe_bufs = (e.dev_alloc(self.B * 16 * esz), e.dev_alloc(self.B * 6 * esz))
        d_pose, d_v6 = self._probe_bufs
        poses = np.tile(np.eye(4).reshape(16), (self.B, 1)).astype(e.io_dtype)
        for a, p in asked.items():
            poses[a] = p
        e.h2d(d_pose, poses)
        e.probe_field(d_pose, d_v6)
        v6 = np.zeros((self.B, 6), dtype=e.io_dtype)
        e.d2h(v6, d_v6)
        for a in asked:
            _send(self.ports[a]["vector_out"], v6[a])
        return asked

    def step_arrays(self, q, null_control=None, want=("qdot_out",), active=None, q_lo=None, q_hi=None):
        """Array path: one cycle for the whole batch without any bottle -- the way to drive thousands of arms
        (the per-arm port polling of :meth:`cycle` is a Python loop, meant for drop-in use and tests)."""
        self._push_state()
        return self.engine.step_host(q, null_control=null_control, want=want, active=active, q_lo=q_lo, q_hi=q_hi)
