#!/opt/conda/bin/python3.9
"""Golden records from the Python-3-clean BLOCKS of reference files that do not parse as a whole (round 4).

Run ONLY in the build container (the reference tree does not exist on the GPU box):

    /opt/conda/bin/python3.9 tests/golden/make_golden_blocks.py

`make_golden.py` harvests whole files that parse as Python 3.  scripts/vf, scripts/bridge, scripts/object_feeder and
src/handlers.py are Python 2 (a `print` statement each: vf:190, bridge:142, object_feeder:85, handlers.py:105), so `ast.parse`
refuses them -- but the pieces listed below are Python-3-clean text.  They are located by their header line, their indented
block is taken AS TEXT, dedented and compiled in a fresh namespace that holds nothing but duck-typed recording ports, the
numpy names the file imports, and Python-2 spellings of three builtins (`xrange`, and `map` / `zip` returning lists).  Same
discipline as make_golden.py: no module of the reference is imported, none of its module-level code runs (signal handlers,
yarp initialisation, port creation, the other loops), nothing is written next to it, and only arrays / JSON records leave.

What is executed, and what each record pins:
  * scripts/vf:163-179            get_weight_matrix                    -> weights_golden.npz   (A2: oracle + host weight parser)
  * scripts/bridge:182-210        LWR_Bridge.set_vel                   -> bridge_golden.npz    ((f)-1: oracle limiter / command
                                                                          form, the GPU's fused limiter + LWR command form)
  * src/handlers.py:109-230       HandleArmNew methods (not __init__)  -> handlers_wire.json   ((b): bottle sequences, value for
    :276-440, :476-520, :544-576  HandleArm, HandleBridge, HandleJController methods              value and type for type)
  * scripts/monitor_distance:168-219  the goal's tracking-state vote   -> tracking_state_golden.json ((f)-3: TrackingState,
                                                                          including the "rot" message's payload)
  * scripts/object_feeder:93-359  the feeder's loop                    -> feeder_wire.json     ((f)-2 / A1: /param and
                                                                          /objectsIn sequences for goal, goalAndNormal,
                                                                          ObstacleP, ObstacleH, remove)
Stand-ins for names that live outside the reference tree are listed where they are set (`frame_to_list`, `length`, `dprint`):
each is the trivial function its call site implies and is named in the record's "standins".
"""
import builtins
import json
import os
import re
import sys
import textwrap

sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# text-block extraction
# ----------------------------------------------------------------------------
def _lines(path):
    with open(path, "r") as f:
        return f.read().split("\n")


def _indent(s):
    return len(s) - len(s.lstrip())


def _find(lines, pattern, lo=0, hi=None):
    rx = re.compile(pattern)
    for k in range(lo, len(lines) if hi is None else hi):
        if rx.match(lines[k]):
            return k
    raise RuntimeError("pattern not found: %r" % pattern)


def _block(lines, start):
    """[start, end): the header at `start` (which may continue over several lines) and everything indented deeper.  Blank and
    comment-only lines do not end a block; trailing ones are trimmed."""
    ind = _indent(lines[start])
    end = start + 1
    while end < len(lines):
        s = lines[end]
        if s.strip() and not s.lstrip().startswith("#") and _indent(s) <= ind and not s.lstrip().startswith(")"):
            break
        end += 1
    while end > start + 1 and (not lines[end - 1].strip() or lines[end - 1].lstrip().startswith("#")):
        end -= 1
    return start, end


def _def_source(path, name, cls=None):
    """Source text (dedented) of top-level function `name`, or of method `name` of class `cls`."""
    lines = _lines(path)
    lo, hi = 0, len(lines)
    if cls is not None:
        lo, hi = _block(lines, _find(lines, r"^class %s\b" % re.escape(cls)))
    ind = r"^" if cls is None else r"^\s+"
    s, e = _block(lines, _find(lines, ind + r"def %s\(" % re.escape(name), lo, hi))
    return textwrap.dedent("\n".join(lines[s:e])) + "\n", (s + 1, e)


def _class_from_methods(path, cls, methods, ns):
    """A class called `cls` whose methods are the reference's method blocks, compiled in `ns` (their globals)."""
    spans = {}
    body = {}
    for m in methods:
        src, span = _def_source(path, m, cls)
        spans[m] = span
        scratch = {}
        exec(compile(src, "%s:%s.%s" % (path, cls, m), "exec"), ns, scratch)
        body[m] = scratch[m]
    return type(cls, (object,), body), spans


def _attr_literal(path, cls, attr):
    """The literal a class's __init__ assigns to self.<attr> (read as text, evaluated with ast.literal_eval)."""
    import ast
    lines = _lines(path)
    lo, hi = _block(lines, _find(lines, r"^class %s\b" % re.escape(cls)))
    k = _find(lines, r"^\s+self\.%s\s*=" % re.escape(attr), lo, hi)
    text = lines[k].split("=", 1)[1]
    while True:
        try:
            return ast.literal_eval(text.strip())
        except SyntaxError:
            k += 1
            text += lines[k]


_PY2 = {
    "xrange": range,
    "map": lambda f, *a: list(builtins.map(f, *a)),
    "zip": lambda *a: list(builtins.zip(*a)),
}


# ----------------------------------------------------------------------------
# recording ports / bottles: exactly the surface the blocks touch
# ----------------------------------------------------------------------------
class RecValue:
    def __init__(self, tag, v):
        self.tag, self.v = tag, v

    def asDouble(self):
        return float(self.v)

    def asInt(self):
        return int(self.v)

    def asString(self):
        return str(self.v)

    def toString(self):
        return str(self.v)

    def asList(self):
        return self.v if self.tag == "l" else None

    def isDouble(self):
        return self.tag == "d"

    def isInt(self):
        return self.tag == "i"


class RecBottle:
    """Records, per element, WHICH add* call made it: 'd' addDouble, 'i' addInt, 's' addString, 'l' addList."""

    def __init__(self, items=None):
        self.items = []
        for it in items or []:
            self._add_py(it)

    def _add_py(self, it):
        if isinstance(it, RecValue):
            self.items.append(it)
        elif isinstance(it, (list, tuple)):
            self.items.append(RecValue("l", RecBottle(it)))
        elif isinstance(it, str):
            self.items.append(RecValue("s", it))
        elif isinstance(it, int):
            self.items.append(RecValue("i", it))
        else:
            self.items.append(RecValue("d", float(it)))

    def clear(self):
        self.items = []

    def addDouble(self, v):
        self.items.append(RecValue("d", float(v)))

    def addInt(self, v):
        self.items.append(RecValue("i", int(v)))

    def addString(self, v):
        self.items.append(RecValue("s", str(v)))

    def addList(self):
        b = RecBottle()
        self.items.append(RecValue("l", b))
        return b

    def add(self, v):
        self.items.append(v)

    def size(self):
        return len(self.items)

    def get(self, i):
        return self.items[i]

    def toString(self):
        return " ".join(v.toString() for v in self.items)

    def __bool__(self):
        return True

    def dump(self):
        return [[v.tag, v.v.dump() if v.tag == "l" else v.v] for v in self.items]


class RecPort:
    def __init__(self, name, log, script=None):
        self.name, self.log = name, log
        self.script = list(script or [])   # bottles a read() returns, oldest first; None entries = nothing this time
        self.out = RecBottle()

    def prepare(self):
        self.out = RecBottle()
        return self.out

    def _emit(self, how):
        self.log.append({"port": self.name, "write": how, "bottle": self.out.dump()})

    def write(self, strict=False):
        self._emit("write(True)" if strict else "write()")

    def writeStrict(self):
        self._emit("writeStrict()")

    def read(self, wait=True):
        if not self.script:
            return None
        return self.script.pop(0)

    def getPendingReads(self):
        return 0


def _jsonable(v):
    if isinstance(v, np.ndarray):
        return {"ndarray": v.tolist()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return {"npscalar": type(v).__name__, "value": v.item()}
    return v


# ----------------------------------------------------------------------------
# A2: get_weight_matrix (vf:163-179)
# ----------------------------------------------------------------------------
def make_weights_golden():
    path = os.path.join(REF, "scripts", "vf")
    src, span = _def_source(path, "get_weight_matrix")
    warned = []
    ns = dict(_PY2, zeros=np.zeros, dprint=lambda *a: warned.append(a))
    exec(compile(src, path, "exec"), ns)
    gwm = ns["get_weight_matrix"]
    rng = np.random.default_rng(164)
    NV = 14
    kinds, nvars, lens, vals, isint, oks, Ws = [], [], [], [], [], [], []
    cases = []
    for n_vars, kind in ((6, "t"), (7, "j"), (14, "j")):
        for ln in (n_vars, n_vars - 1, n_vars + 1, 0, 1):
            for rep in range(3):
                cases.append((kind, n_vars, ln, rep))
    for kind, n_vars, ln, rep in cases:
        w = rng.uniform(0.0, 2.0, ln)
        ints = np.zeros(ln, dtype=bool)
        if rep == 1 and ln:            # integer weights arrive as ints: asDouble() of an int (handlers send [1]*7 defaults)
            w = np.round(w)
            ints[:] = True
        if rep == 2 and ln > 2:
            w[1] = 0.0
            w[2] = -0.5                # the function does not validate values
        b = RecBottle([kind] + [int(x) if i else float(x) for x, i in zip(w, ints)])
        W = gwm(b, n_vars)
        kinds.append(0 if kind == "t" else 1)
        nvars.append(n_vars)
        lens.append(ln)
        vals.append(np.pad(w, (0, NV + 1 - ln), constant_values=np.nan))
        isint.append(np.pad(ints, (0, NV + 1 - ln)))
        oks.append(W is not None)
        full = np.full((NV, NV), np.nan)
        if W is not None:
            full[:n_vars, :n_vars] = np.asarray(W)
        Ws.append(full)
    np.savez(os.path.join(OUT, "weights_golden.npz"), kind=np.array(kinds), n_vars=np.array(nvars), length=np.array(lens),
             values=np.stack(vals), is_int=np.stack(isint), accepted=np.array(oks), W=np.stack(Ws))
    print("weights_golden.npz: %d cases (%d refused), vf:%d-%d" % (len(cases), len(cases) - sum(oks), span[0], span[1]))


# ----------------------------------------------------------------------------
# (f)-1: LWR_Bridge.set_vel (bridge:182-210)
# ----------------------------------------------------------------------------
def make_bridge_golden():
    path = os.path.join(REF, "scripts", "bridge")
    log = []
    quiet = dict(vars(builtins))
    quiet["print"] = lambda *a, **k: None
    ns = dict(_PY2, __builtins__=quiet, max_vel=0.0, direct_control=False)
    cls, spans = _class_from_methods(path, "LWR_Bridge", ["set_vel"], ns)
    rng = np.random.default_rng(182)
    NV = 14
    rec = {k: [] for k in ("n", "qdot", "last_q", "last_qcmded", "max_vel", "direct", "cmd")}
    for n in (7, 14, 6):
        for c in range(48):
            br = cls.__new__(cls)
            br.nJoints = n
            br.qcmd_port = RecPort("qcmd", log)
            max_vel = float(rng.choice([0.05, 0.2, 0.41, 1.0]))
            scale = [0.01, 0.3, 3.0][c % 3]                 # well below, around and well above max_vel
            qdot = rng.normal(0.0, scale, n)
            direct = (c % 8) == 7
            if c % 8 == 6:
                direct, qdot = True, np.zeros(n)            # what the bridge loop can produce: every weight 0 -> mixer sum 0
            if c % 12 == 3:
                qdot[rng.integers(n)] = max_vel             # exactly AT the limit: `>` is strict (bridge:191)
            if c % 12 == 9:
                qdot[:] = 0.0                               # leading_vel 0: no division (bridge:191-194)
            br.last_q = rng.uniform(-2.0, 2.0, n).tolist()
            br.last_qcmded = (np.array(br.last_q) + rng.normal(0.0, 0.01, n)).tolist()
            ns["max_vel"], ns["direct_control"] = max_vel, direct
            del log[:]
            br.set_vel(qdot.tolist())
            (sent,) = log
            assert sent["write"] == "write()" and all(t == "d" for t, _ in sent["bottle"])
            pad = lambda v: np.pad(np.asarray(v, dtype=float), (0, NV - n), constant_values=np.nan)  # noqa: E731
            rec["n"].append(n)
            rec["qdot"].append(pad(qdot))
            rec["last_q"].append(pad(br.last_q))
            rec["last_qcmded"].append(pad(br.last_qcmded))
            rec["max_vel"].append(max_vel)
            rec["direct"].append(direct)
            rec["cmd"].append(pad([v for _, v in sent["bottle"]]))
    np.savez(os.path.join(OUT, "bridge_golden.npz"), **{k: np.asarray(v) for k, v in rec.items()})
    print("bridge_golden.npz: %d cases, bridge:%d-%d" % (len(rec["n"]), spans["set_vel"][0], spans["set_vel"][1]))


# ----------------------------------------------------------------------------
# (b): handler methods -> bottle sequences
# ----------------------------------------------------------------------------
class _YarpValue:
    """`yarp.Value.asDouble` is used unbound (handlers.py:160,331,479,565)."""
    asDouble = staticmethod(lambda v: v.asDouble())


class _Yarp:
    Value = _YarpValue


class _Clock:
    def __init__(self):
        self.now = 100.0

    def time(self):
        return self.now

    def sleep(self, s):
        self.now += s


def _dist_bottle(entries):
    """/dmonitor/distOut: one list (id, xyz distance, rotation distance in degrees) per object (monitor_distance:163-166)."""
    return RecBottle([[float(i), float(a), float(b)] for i, a, b in entries])


def make_handlers_wire():
    path = os.path.join(REF, "src", "handlers.py")
    quiet = dict(vars(builtins))
    quiet["print"] = lambda *a, **k: None
    clock = _Clock()
    ns = dict(_PY2, __builtins__=quiet, yarp=_Yarp, time=clock, sleep=clock.sleep, pi=np.pi, array=np.array,
              frame_to_list=lambda f: list(f))
    records = []
    spans = {}

    def run(cls_name, cls, setup, method, args, kwargs=None, scripts=None):
        log = []
        obj = cls.__new__(cls)
        for attr, val in setup.items():
            setattr(obj, attr, val(log) if callable(val) else val)
        for attr, script in (scripts or {}).items():
            getattr(obj, attr).script = list(script)
        clock.now = 100.0
        ret = getattr(obj, method)(*args, **(kwargs or {}))
        records.append({"class": cls_name, "method": method, "args": _jsonable(list(args)), "kwargs": _jsonable(kwargs or {}),
                        "reads": {a: [None if b is None else b.dump() for b in s] for a, s in (scripts or {}).items()},
                        "writes": log, "returns": _jsonable(ret), "clock_advanced": clock.now - 100.0})

    def port(name):
        return lambda log: RecPort(name, log)

    frame = [0.0, -1.0, 0.0, 0.45, 1.0, 0.0, 0.0, -0.2, 0.0, 0.0, 1.0, 0.6, 0.0, 0.0, 0.0, 1.0]
    ident_ints = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    q7 = [0.1, -0.2, 0.3, 1.1, -0.5, 0.7, 0.0]

    # ---- HandleArmNew (handlers.py:109-230) ----
    m_new = ["set_sim_arm_q", "set_vf_tool", "set_stiffness", "go_cart", "_write_yarp_port", "go_joint", "go_xyz", "go_rot",
             "get_cart_pose", "get_dist_cart_goal", "get_dist_joint_goal", "get_joint_angles", "set_controller_mixer",
             "set_cartesian_control", "set_joint_control", "set_wik_joint_weights", "set_wik_cart_weights", "set_tool"]
    New, spans["HandleArmNew"] = _class_from_methods(path, "HandleArmNew", m_new, ns)
    new_setup = {k: port(k) for k in ("object_port", "joint_ref_port", "bridge_weight_port", "vf_weight_port", "tool_port",
                                      "joint_sim_qin_port", "pose_port", "distout_port", "bridge_encoders_port")}
    new_setup["current_slowdown_distance"] = _attr_literal(path, "HandleArmNew", "current_slowdown_distance")
    for args in ([frame], [ident_ints], [np.array(frame)]):
        run("HandleArmNew", New, new_setup, "go_cart", args)
    for args in ([q7], [[1, 0, 2, 0, 0, 0, 0]], [np.array(q7)], [[0.5, 1, "x", True, None, 2.5, np.float64(3.0), np.int64(4)]]):
        run("HandleArmNew", New, new_setup, "go_joint", args)
    for cart in (True, False):
        for joint in (True, False):
            for null in (True, False):
                run("HandleArmNew", New, new_setup, "set_controller_mixer", [], {"cart": cart, "joint": joint, "null": null})
    run("HandleArmNew", New, new_setup, "set_controller_mixer", [])
    run("HandleArmNew", New, new_setup, "set_cartesian_control", [])
    run("HandleArmNew", New, new_setup, "set_joint_control", [])
    run("HandleArmNew", New, new_setup, "set_wik_joint_weights", [[1.0, 0.5, 0.25, 1.0, 2.0, 1.0, 0.1]])
    run("HandleArmNew", New, new_setup, "set_wik_joint_weights", [[1, 1, 1, 0, 1, 1, 1]])
    run("HandleArmNew", New, new_setup, "set_wik_cart_weights", [[1.0, 1.0, 1.0, 0.1, 0.1, 0.1]])
    run("HandleArmNew", New, new_setup, "set_wik_cart_weights", [np.array([1.0, 1.0, 0.0, 0.5, 0.5, 0.5])])
    for args in ([frame], [ident_ints], [np.array(frame)]):
        run("HandleArmNew", New, new_setup, "set_tool", args)
    run("HandleArmNew", New, new_setup, "set_sim_arm_q", [q7])
    run("HandleArmNew", New, new_setup, "set_sim_arm_q", [np.array(q7)])
    for stub in ("set_vf_tool", "set_stiffness", "go_xyz", "go_rot"):
        run("HandleArmNew", New, new_setup, stub, [[1.0, 2.0, 3.0]])
    run("HandleArmNew", New, new_setup, "get_joint_angles", [])
    run("HandleArmNew", New, new_setup, "get_cart_pose", [], scripts={"pose_port": [RecBottle(frame)]})
    run("HandleArmNew", New, new_setup, "get_dist_cart_goal", [],
        scripts={"distout_port": [_dist_bottle([(4, 0.3, 20.0), (5, 0.2, 10.0)]), _dist_bottle([(5, 0.7, 1.0), (0, 0.125, 45.0)])]})
    setup_goal = dict(new_setup, joint_goal=q7)
    run("HandleArmNew", New, setup_goal, "get_dist_joint_goal", [], scripts={"bridge_encoders_port": [RecBottle([0.0, 0.1, 0.2, 1.0, -1.0, 0.7, 0.5])]})

    # ---- HandleArm (handlers.py:276-440) ----
    m_arm = ["setTool", "set_stiffness", "sendFrame", "gotoPos", "setOrient", "getPose", "gotoPose", "gotoFrame", "gotThere",
             "gotoPosBlocking", "gotoPosBlockingGrasp"]
    Arm, spans["HandleArm"] = _class_from_methods(path, "HandleArm", m_arm, ns)

    def arm_setup():
        return {"outp": port("outp"), "stiffness_port": port("stiffness_port"), "goaldistp": port("goaldistp"),
                "posep": port("posep"), "toolp": port("toolp"),
                "current_frame": list(_attr_literal(path, "HandleArm", "current_frame")),
                "current_slowdown_distance": _attr_literal(path, "HandleArm", "current_slowdown_distance"),
                "goal_threshold": _attr_literal(path, "HandleArm", "goal_threshold")}

    orient = [0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    run("HandleArm", Arm, arm_setup(), "sendFrame", [])
    run("HandleArm", Arm, arm_setup(), "gotoPos", [[0.5, -0.25, 0.75]])
    run("HandleArm", Arm, arm_setup(), "gotoPos", [[1, 0, 2]])
    run("HandleArm", Arm, arm_setup(), "setOrient", [orient])
    run("HandleArm", Arm, arm_setup(), "gotoPose", [[0.5, -0.25, 0.75], orient])
    run("HandleArm", Arm, arm_setup(), "set_stiffness", [[200.0, 200.0, 100.0, 10.0, 10.0, 10.0]])
    run("HandleArm", Arm, arm_setup(), "setTool", [frame])        # frame_to_list stand-in: the 16 values as they are
    run("HandleArm", Arm, arm_setup(), "getPose", [], scripts={"posep": [RecBottle(frame)]})
    run("HandleArm", Arm, arm_setup(), "gotThere", [], scripts={"goaldistp": [_dist_bottle([(4, 0.001, 0.0), (0, 0.02, 3.0)])]})
    run("HandleArm", Arm, arm_setup(), "gotThere", [], scripts={"goaldistp": [_dist_bottle([(0, 0.0099, 90.0)])]})
    run("HandleArm", Arm, arm_setup(), "gotThere", [], scripts={"goaldistp": [_dist_bottle([(3, 0.0, 0.0)])]})
    run("HandleArm", Arm, arm_setup(), "gotoFrame", [frame], {"wait": 0.0})
    run("HandleArm", Arm, arm_setup(), "gotoFrame", [frame], {"wait": 1.0, "goal_precision": [0.01, 0.05]},
        scripts={"goaldistp": [_dist_bottle([(0, 0.001, 0.1)]),      # arrives in the FIRST poll: discarded (handlers.py:366-381)
                               None,
                               _dist_bottle([(0, 0.2, 30.0)]),
                               _dist_bottle([(0, 0.009, 3.0)]),      # xyz ok, rotation 3 deg = 0.052 rad: not yet
                               _dist_bottle([(0, 0.009, 2.0)])]})    # both inside
    run("HandleArm", Arm, arm_setup(), "gotoFrame", [frame], {"wait": 0.05, "goal_precision": [0.01, 0.05]},
        scripts={"goaldistp": [None, _dist_bottle([(0, 0.2, 30.0)]), None, None, None, None, None, None]})   # times out
    run("HandleArm", Arm, arm_setup(), "gotoFrame", [frame], {"wait": 1.0, "goal_precision": [0.01, 0.05]},
        scripts={"goaldistp": [None, _dist_bottle([(0, 0.001, 0.1)])]})  # first poll empty: the first REPORT counts
    near = _dist_bottle([(0, 0.005, 0.0)])
    far = _dist_bottle([(0, 0.5, 0.0)])
    run("HandleArm", Arm, arm_setup(), "gotoPosBlocking", [[0.5, -0.25, 0.75]], {"timeout": 20},
        scripts={"goaldistp": [near] * 10 + [far, far, near]})       # the first ten reports are ignored (handlers.py:409-411)
    run("HandleArm", Arm, arm_setup(), "gotoPosBlockingGrasp", [[0.5, -0.25, 0.75]], {"timeout": 20},
        scripts={"goaldistp": [far] * 12 + [near]})

    # ---- HandleBridge (handlers.py:476-520) ----
    m_br = ["read_joint_angles", "joint_controller", "cartesian_controller", "torso_joints", "set_VFW", "set_weights"]
    Br, spans["HandleBridge"] = _class_from_methods(path, "HandleBridge", m_br, ns)
    br_setup = {"outp": port("outp"), "torso_port": port("torso_port"), "VFW_port": port("VFW_port"),
                "encoders_port": port("encoders_port"), "torso": True}
    run("HandleBridge", Br, br_setup, "joint_controller", [])
    run("HandleBridge", Br, br_setup, "cartesian_controller", [])
    run("HandleBridge", Br, br_setup, "torso_joints", [[0, 2]])
    run("HandleBridge", Br, dict(br_setup, torso=False), "torso_joints", [[0, 2]])
    run("HandleBridge", Br, br_setup, "set_weights", [])
    run("HandleBridge", Br, br_setup, "set_weights", ["task", [1.0, 1.0, 1.0, 0.2, 0.2, 0.2]])
    run("HandleBridge", Br, br_setup, "set_weights", ["joint", [1.0, 0.5, 1.0, 1.0, 1.0, 1.0, 0.0]])
    run("HandleBridge", Br, br_setup, "set_weights", ["anything else", [2, 2, 2, 2, 2, 2, 2]])
    run("HandleBridge", Br, br_setup, "set_VFW", ["task", [1.0, 1.0, 1.0, 0.5, 0.5, 0.5]])
    run("HandleBridge", Br, br_setup, "read_joint_angles", [], scripts={"encoders_port": [RecBottle(q7)]})

    # ---- HandleJController (handlers.py:544-576) ----
    JC, spans["HandleJController"] = _class_from_methods(path, "HandleJController", ["set_ref_js"], ns)
    jc_setup = {"outp": port("outp"), "inp": port("inp")}
    run("HandleJController", JC, jc_setup, "set_ref_js", [q7])
    run("HandleJController", JC, jc_setup, "set_ref_js", [[0, 1, 0, 1, 0, 1, 0]])
    run("HandleJController", JC, jc_setup, "set_ref_js", [np.array(q7)])
    tol = [0.01] * 7
    run("HandleJController", JC, jc_setup, "set_ref_js", [np.array(q7)], {"wait": 1.0, "goal_precision": tol},
        scripts={"inp": [None, RecBottle([v + 0.02 for v in q7]), RecBottle([v + 0.0099 for v in q7])]})
    run("HandleJController", JC, jc_setup, "set_ref_js", [np.array(q7)], {"wait": 0.03, "goal_precision": tol},
        scripts={"inp": [RecBottle([v - 0.5 for v in q7])] + [None] * 6})
    run("HandleJController", JC, jc_setup, "set_ref_js", [np.array(q7)], {"wait": -1, "goal_precision": tol},
        scripts={"inp": [None] * 5 + [RecBottle(q7)]})
    run("HandleJController", JC, jc_setup, "set_ref_js", [np.array(q7)], {"wait": 1.0, "goal_precision": [0.01] * 3})

    doc = {"source": "src/handlers.py", "spans": {c: {m: list(s) for m, s in d.items()} for c, d in spans.items()},
           "standins": {"frame_to_list": "list(frame): the 16 values as they are (arcospyu, absent)",
                        "time/sleep": "a clock that sleep() advances", "print": "silenced",
                        "xrange/map/zip": "Python-2 forms (range; map and zip returning lists)"},
           "tags": {"d": "addDouble", "i": "addInt", "s": "addString", "l": "addList"},
           "records": records}
    with open(os.path.join(OUT, "handlers_wire.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("handlers_wire.json: %d records" % len(records))


# ----------------------------------------------------------------------------
# (f)-3: the goal's tracking-state vote (monitor_distance:168-219)
# ----------------------------------------------------------------------------
def make_tracking_state_golden():
    import ast
    path = os.path.join(REF, "scripts", "monitor_distance")
    lines = _lines(path)
    s, e = _block(lines, _find(lines, r"^\s+if i == 0:"))
    code = compile(textwrap.dedent("\n".join(lines[s:e])) + "\n", path, "exec")
    log = []
    ns = dict(_PY2, i=0, tracking_state_port=RecPort("tracking_state", log))
    for name in ("distanceXYZ_th", "track_error_xyz_th", "distanceOrient_th", "track_error_rot_th", "tracking_buffer",
                 "tracking_buffer_size", "last_tracking_xyz_state", "last_tracking_rot_state", "tracking_xyz_state",
                 "tracking_rot_state"):
        k = _find(lines, r"^%s\s*=" % name)
        ns[name] = ast.literal_eval(lines[k].split("=", 1)[1].strip())
    consts = {k: ns[k] for k in ("distanceXYZ_th", "track_error_xyz_th", "distanceOrient_th", "track_error_rot_th", "tracking_buffer_size")}
    rng = np.random.default_rng(168)
    T = 400
    # phases: far + not following -> far + following (xyz first, rot later) -> on goal -> rot leaves alone -> noisy border
    samples, messages = [], []
    for t in range(T):
        if t < 60:
            row = [0.5, 40.0, 0.5, 0.5]
        elif t < 100:
            row = [0.4, 30.0, 0.01, 0.5]          # xyz follows, rot does not: the two votes differ
        elif t < 150:
            row = [0.3, 20.0, 0.01, 0.01]
        elif t < 200:
            row = [0.01, 0.5, 0.3, 0.3]           # on goal whatever the errors
        elif t < 250:
            row = [0.01, 15.0, 0.3, 0.3]          # rotation pushed away, xyz stays on goal
        elif t < 300:
            row = [0.3, 0.2, 0.3, 0.01]           # xyz not following, rot on goal
        else:
            row = [float(rng.choice([0.01, 0.3])), float(rng.choice([0.5, 5.0])), float(rng.choice([0.05, 0.2])), float(rng.choice([0.05, 0.2]))]
        if t in (120, 121):
            row[0] = consts["distanceXYZ_th"]     # exactly ON a threshold: every comparison is strict -> state keeps its default
        if t in (122, 123):
            row[2] = consts["track_error_xyz_th"]
        ns["distanceXYZ"], ns["distanceOrient"], ns["track_error_xyz"], ns["track_error_rot"] = row
        del log[:]
        exec(code, ns)
        samples.append([float(x) for x in row])
        for m in log:
            assert m["write"] == "writeStrict()"
            messages.append([t, m["bottle"][0][1], m["bottle"][1][1]])
    doc = {"source": "scripts/monitor_distance:%d-%d" % (s + 1, e), "constants": consts,
           "columns": ["distanceXYZ", "distanceOrient_deg", "track_error_xyz", "track_error_rot"],
           "samples": samples, "messages": messages}
    with open(os.path.join(OUT, "tracking_state_golden.json"), "w") as f:
        json.dump(doc, f)
    print("tracking_state_golden.json: %d samples, %d messages" % (T, len(messages)))


# ----------------------------------------------------------------------------
# (f)-2 / A1: the object feeder's loop (object_feeder:93-359)
# ----------------------------------------------------------------------------
class _Stop:
    def __init__(self):
        self.flag = False

    def __bool__(self):
        return self.flag


class _Py2IntDict(dict):
    """`objects = {}` (object_feeder:88) as CPython 2 iterates it: keys are small non-negative ints, hash(i) == i, slot = i & mask,
    so iteration is in ascending key order whatever the insertion order.  (A Python 3 dict would iterate in insertion order and
    send the obstacles of "obstacles before any goal" ahead of the goal.)"""

    def __iter__(self):
        return iter(sorted(dict.keys(self)))


class _ScriptedIn(RecPort):
    """/object: hands out the scripted bottles one per loop iteration, then ends the loop."""

    def __init__(self, name, log, script, stop):
        RecPort.__init__(self, name, log, script)
        self.stop = stop

    def read(self, wait=False):
        if not self.script:
            self.stop.flag = True
            return None
        b = self.script.pop(0)
        self.log.append({"port": self.name, "read": b.dump()})
        return b


def make_feeder_wire():
    path = os.path.join(REF, "scripts", "object_feeder")
    lines = _lines(path)
    s, e = _block(lines, _find(lines, r"^while not stop:"))
    code = compile("\n".join(lines[s:e]) + "\n", path, "exec")
    rng = np.random.default_rng(93)

    def frame_at(x, y, z):
        return [1.0, 0.0, 0.0, x, 0.0, 1.0, 0.0, y, 0.0, 0.0, 1.0, z, 0.0, 0.0, 0.0, 1.0]

    goal16 = [0.0, -1.0, 0.0, 0.45, 1.0, 0.0, 0.0, -0.2, 0.0, 0.0, 1.0, 0.6, 0.0, 0.0, 0.0, 1.0]
    normal21 = goal16 + [0.0, 0.0, -2.0, 0.4, 0.15]                 # axis (not unit length), cut angle, cut distance
    obst = lambda k: frame_at(*rng.uniform(-0.5, 0.5, 3).round(3).tolist()) + [0.05 + 0.01 * k, 5.0]   # noqa: E731
    table = frame_at(0.0, 0.0, 0.1) + [0.0, 0.0, 1.0, 0.08, 5.0]
    scenarios = {
        "obstacles_before_any_goal": [["set", "ObstacleP", 0, obst(0)], ["set", "ObstacleP", 1, obst(1)], ["set", "goal", goal16 + [0.1]]],
        "goal_16_values_gets_default_slowdown": [["set", "goal", goal16]],
        "goal_wrong_length": [["set", "goal", goal16[:12]], ["set", "goal", goal16 + [0.1]], ["set", "goal", goal16[:5]]],
        "goal_then_obstacles_then_remove": [["set", "goal", goal16 + [0.1]], ["set", "ObstacleP", 0, obst(0)],
                                            ["set", "ObstacleP", 1, obst(1)], ["set", "ObstacleH", 2, table],
                                            ["remove", 1], ["remove", 1], ["remove", 7], ["set", "ObstacleP", 0, obst(3)]],
        "goal_and_normal": [["set", "goalAndNormal", normal21], ["set", "ObstacleP", 0, obst(0)], ["set", "ObstacleH", 1, table],
                            ["set", "goalAndNormal", normal21 + [0.2]], ["set", "goal", goal16 + [0.1]]],
        "malformed": [["set", "goal", goal16 + [0.1]], ["set", "ObstacleP", 0, obst(0)[:17]], ["set", "ObstacleH", 0, table[:20]],
                      ["set", "goalAndNormal", normal21[:20]], ["set", "goal"], ["fly", "goal", goal16], ["set"],
                      ["set", "teapot", goal16], ["set", "ObstacleX", 0, obst(0)], ["set", "goal", 1, 2, 3]],
    }
    doc = {"source": "scripts/object_feeder:%d-%d" % (s + 1, e),
           "standins": {"length": "Euclidean norm (vfl.vfl.length, absent; object_feeder:293 divides the axis by it)",
                        "dprint": "silenced", "yarp_ctrl.update / yarp.Time_delay": "no-ops",
                        "objects": "dict iterating in ascending key order (CPython 2's order for small int keys)"},
           "note": "`for objectNum in objects` (object_feeder:216): the dictionary handed to the block iterates in ascending key "
                   "order, as CPython 2 does for small non-negative int keys (stand-in `_Py2IntDict`).",
           "tags": {"d": "addDouble", "i": "addInt", "s": "addString", "l": "addList"}, "scenarios": {}}
    for name, script in scenarios.items():
        log = []
        stop = _Stop()

        class _Ctl:
            update = staticmethod(lambda: None)

        class _Y:
            Time_delay = staticmethod(lambda s: None)
            Bottle = RecBottle

        ns = dict(_PY2, stop=stop, yarp_ctrl=_Ctl, yarp=_Y, dprint=lambda *a: None, array=np.array,
                  length=lambda v: float(np.sqrt(np.dot(v, v))), objects=_Py2IntDict(), init_pose=False,
                  objectPort=_ScriptedIn("object", log, [RecBottle(m) for m in script], stop),
                  paramPort=RecPort("param", log), object_f_port=RecPort("objectf", log), objectOutPort=RecPort("objectOut", log))
        exec(code, ns)
        doc["scenarios"][name] = {"events": _jsonable(log), "objects_left": sorted(ns["objects"])}
    with open(os.path.join(OUT, "feeder_wire.json"), "w") as f:
        json.dump(doc, f)
    print("feeder_wire.json: %s" % {k: len(v["events"]) for k, v in doc["scenarios"].items()})


if __name__ == "__main__":
    make_weights_golden()
    make_bridge_golden()
    make_handlers_wire()
    make_tracking_state_golden()
    make_feeder_wire()
