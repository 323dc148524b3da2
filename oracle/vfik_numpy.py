"""NumPy float64 restatement of vfclik's per-cycle loop, one arm per call.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/vfik_oracle.h for the same rule on the C restatement).  It is written from the
reference's call sites and keeps their shape on purpose -- a ``Lafik``-like kinematics object,
``VectorField`` / ``ScalarField`` closures composed with ``+`` and ``*``, a mini PyKDL -- so that
it can be timed as "the reference-style CPU loop" and read side by side with
/root/reference/scripts/vf, scripts/nullspace, scripts/debug_jointlimits and
src/command_mixer.py (cited as file:line below).

Pinning (details in DESIGN.md):
  * ``CommandMixer``            -- pinned by tests/golden/mixer_golden.npz (bit-exact)
  * ``restrict / Nullspace``    -- pinned by tests/golden/nullspace_golden.npz
  * everything that the reference imports from vfl / arcospyu / PyKDL (field primitives,
    normCart, FK, Jacobian, getIKV, distToCenter) -- PARITY UNPINNED: those packages are not in the
    reference tree nor on this machine, and no version is pinned (setup.py:6-20).  The formulas
    below are this build's own definitions (DESIGN.md "Spec").
"""
import math

import numpy as np
from numpy.linalg import norm, pinv, svd

EPS_LEN = 1e-12
D_FLOOR = 1e-9
MAG_CAP = 1e6


# ---------------------------------------------------------------------------------------------
# mini PyKDL: only what scripts/vf touches (vf:329-332, 356, 456-459)
# ---------------------------------------------------------------------------------------------
class Frame:
    def __init__(self, M=None, p=None):
        self.M = np.eye(3) if M is None else np.asarray(M, dtype=float)
        self.p = np.zeros(3) if p is None else np.asarray(p, dtype=float)

    def __mul__(self, other):  # PyKDL Frame * Frame  (vf:330)
        return Frame(self.M @ other.M, self.M @ other.p + self.p)


class Twist:
    def __init__(self, vel, rot):
        self.vel = np.asarray(vel, dtype=float)
        self.rot = np.asarray(rot, dtype=float)

    def RefPoint(self, v_base_AB):  # KDL: Twist(vel + rot x AB, rot)   (vf:459)
        return Twist(self.vel + np.cross(self.rot, v_base_AB), self.rot)


def rot_log(R, Rg):
    """Rotation vector (base frame) that takes R to Rg: log(Rg R^T) = KDL diff(R, Rg)."""
    E = Rg @ R.T
    a = 0.5 * np.array([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]])
    c = 0.5 * (np.trace(E) - 1.0)
    s = norm(a)
    th = math.atan2(s, c)
    if s < 1e-4 and c < 0.0:  # near pi: use the symmetric part  c I + (1-c) a a^T
        k = int(np.argmax(np.diag(E)))
        ax = np.zeros(3)
        ax[k] = math.sqrt(max((E[k, k] - c) / (1.0 - c), 0.0))
        for j in range(3):
            if j != k:
                ax[j] = 0.5 * (E[j, k] + E[k, j]) / ((1.0 - c) * ax[k])
        if ax @ a < 0.0:
            ax = -ax
        return ax / norm(ax) * th
    if s < EPS_LEN:
        return np.zeros(3)
    return a / s * th


def diff(Fa, Fb):  # PyKDL.diff(F_a, F_b, dt=1)   (vf:331,356)
    return Twist(Fb.p - Fa.p, rot_log(Fa.M, Fb.M))


def listToKdlFrame(l):  # arcospyu helper used at vf:329
    A = np.asarray(l, dtype=float).reshape(4, 4)
    return Frame(A[:3, :3].copy(), A[:3, 3].copy())


def kdlFrameToList(F):  # vf:332,342
    A = np.eye(4)
    A[:3, :3] = F.M
    A[:3, 3] = F.p
    return A.reshape(16).tolist()


# ---------------------------------------------------------------------------------------------
# vfl stand-in: VectorField / ScalarField algebra (vf:150-151, 278-292) and the primitives
# ---------------------------------------------------------------------------------------------
class VectorField:
    def __init__(self, f):
        self.getVector = f

    def __add__(self, other):
        f, g = self.getVector, other.getVector
        return VectorField(lambda pos: f(pos) + g(pos))

    def __mul__(self, k):
        f = self.getVector
        return VectorField(lambda pos: f(pos) * k)

    def normCart(self):
        f = self.getVector

        def g(pos):
            v = np.array(f(pos), dtype=float)
            nt, nr = norm(v[0:3]), norm(v[3:6])
            v[0:3] = v[0:3] / nt if nt > EPS_LEN else 0.0
            v[3:6] = v[3:6] / nr if nr > EPS_LEN else 0.0
            return v

        return VectorField(g)


class ScalarField:
    def __init__(self, f):
        self.getScalar = f

    def __mul__(self, other):
        f, g = self.getScalar, other.getScalar
        return ScalarField(lambda pos: f(pos) * g(pos))


def _pos_of(pos16):
    return np.array([pos16[3], pos16[7], pos16[11]], dtype=float)


class _Prim:
    rot_slowdown = 0.3

    def setParams(self, params):
        self.p = [float(x) for x in params]

    def getVector(self, pos):
        return np.zeros(6)

    def getScalar(self, pos):
        return np.ones(2)


class NullField(_Prim):  # type 0 (vf:148-149)
    pass


class PointAttractor(_Prim):  # type 1: frame16 + slow-down distance (object_feeder:236-241)
    def _err(self, pos):
        G = listToKdlFrame(self.p[:16])
        F = listToKdlFrame(pos)
        d = G.p - F.p
        r = rot_log(F.M, G.M)
        return d, r

    def getVector(self, pos):
        d, r = self._err(pos)
        D, th = norm(d), norm(r)
        v = np.zeros(6)
        if D > EPS_LEN:
            v[0:3] = d / D
        if th > EPS_LEN:
            v[3:6] = r / th
        return v

    def getScalar(self, pos):
        d, r = self._err(pos)
        D, th = norm(d), norm(r)
        ds = self.p[16]
        s0 = min(1.0, D / ds) if ds > 0.0 else 1.0
        s1 = min(1.0, th / self.rot_slowdown) if self.rot_slowdown > 0.0 else 1.0
        return np.array([s0, s1])


class DecayRepeller(_Prim):  # type 2: x y z radius safeDist order (object_feeder:326-333)
    def getVector(self, pos):
        d = np.array(self.p[0:3]) - _pos_of(pos)
        D = max(norm(d), D_FLOOR)
        with np.errstate(over="ignore"):
            m = min(float(np.float64((self.p[3] + self.p[4]) / D) ** np.float64(self.p[5])), MAG_CAP)
        return np.concatenate([m * d / D, np.zeros(3)])


class HemisphereRepeller(_Prim):  # type 4: x y z nx ny nz safeDist order (object_feeder:344-353)
    def getVector(self, pos):
        nrm = np.array(self.p[3:6])
        nn = norm(nrm)
        if nn <= EPS_LEN:
            return np.zeros(6)
        nh = nrm / nn
        h = float((_pos_of(pos) - np.array(self.p[0:3])) @ nh)
        with np.errstate(over="ignore"):
            m = min(float(np.float64(self.p[6] / max(h, D_FLOOR)) ** np.float64(self.p[7])), MAG_CAP)
        return np.concatenate([-m * nh, np.zeros(3)])


class FunnelAttractor(_Prim):  # type 5: x y z ax ay az cutAngle angleOrder cutDist distOrder (:270-279)
    def getVector(self, pos):
        axis = np.array(self.p[3:6])
        an = norm(axis)
        if an <= EPS_LEN:
            return np.zeros(6)
        ah = axis / an
        w = _pos_of(pos) - np.array(self.p[0:3])
        along = float(w @ ah)
        perp = w - along * ah
        P, dist = norm(perp), norm(w)
        phi = math.atan2(P, along)
        ga = min(1.0, (phi / self.p[6]) ** self.p[7]) if self.p[6] > 0.0 else 1.0
        with np.errstate(over="ignore"):
            gd = min(1.0, float(np.float64(self.p[8] / max(dist, D_FLOOR)) ** np.float64(self.p[9])))
        return np.concatenate([-perp / max(P, D_FLOOR) * ga * gd, np.zeros(3)])


def vectorFieldLibrary(rot_slowdown=0.3):  # vfl.vfl.vectorFieldLibrary()  (vf:146)
    _Prim.rot_slowdown = rot_slowdown
    return {0: NullField, 1: PointAttractor, 2: DecayRepeller, 4: HemisphereRepeller, 5: FunnelAttractor}


def build_total_field(vectorFields, vfDB):
    """The rebuild at vf:276-293: force-weighted sum, scalar-field product, normCart.
    vectorFields: {id: [force, type, params]}.  Iterates in ascending id."""
    vftemp = vfDB[0]()
    vftemp.setParams([])
    totalVF = VectorField(vftemp.getVector)
    totalSF = ScalarField(vftemp.getScalar)
    for vfNum in sorted(vectorFields):
        force, vfType, params = vectorFields[vfNum]
        fn = vfDB[vfType]()
        fn.setParams(params)
        totalVF = totalVF + VectorField(fn.getVector) * force
        totalSF = totalSF * ScalarField(fn.getScalar)
    return totalVF.normCart(), totalSF


# ---------------------------------------------------------------------------------------------
# arcospyu.robot_tools.Lafik stand-in (vf:153,305-318,461; nullspace:166-175; debug_jointlimits:62-67)
# ---------------------------------------------------------------------------------------------
class Lafik:
    def __init__(self, chain_B, jtype, q_lo, q_hi, lam=0.1):
        """chain_B: (n+1, 3, 4) fixed transforms of the z-normal form (include/vfik_types.h)."""
        self.B = np.asarray(chain_B, dtype=float).reshape(-1, 3, 4)
        self.numJnts = self.B.shape[0] - 1
        self.jtype = list(jtype)
        self.joint_limits = [(float(a), float(b)) for a, b in zip(q_lo, q_hi)]
        self.lam = float(lam)
        self.tweights = np.eye(6)
        self.jweights = np.eye(self.numJnts)
        self.jnt_pos = [0.0] * self.numJnts
        self._fk()

    def _fk(self):
        X = Frame(self.B[0][:, :3], self.B[0][:, 3])
        self._z, self._o = [], []
        for i in range(self.numJnts):
            self._z.append(X.M[:, 2].copy())
            self._o.append(X.p.copy())
            qi = self.jnt_pos[i]
            if self.jtype[i] == 0:
                c, s = math.cos(qi), math.sin(qi)
                Jz = Frame(np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]), np.zeros(3))
            else:
                Jz = Frame(np.eye(3), np.array([0.0, 0.0, qi]))
            X = X * Jz * Frame(self.B[i + 1][:, :3], self.B[i + 1][:, 3])
        self.kdlframe = X
        self.frame = kdlFrameToList(X)

    @property
    def jntsList(self):
        return list(self.jnt_pos)

    @jntsList.setter
    def jntsList(self, q):  # vf:316
        self.jnt_pos = [float(x) for x in q]
        self._fk()

    def get_limits(self):  # nullspace:167
        return [list(l) for l in self.joint_limits]

    def jac_list(self):  # nullspace:175 -- 6 x n, flange reference point, base frame
        self._fk()
        pe = self.kdlframe.p
        J = np.zeros((6, self.numJnts))
        for i in range(self.numJnts):
            if self.jtype[i] == 0:
                J[0:3, i] = np.cross(self._z[i], pe - self._o[i])
                J[3:6, i] = self._z[i]
            else:
                J[0:3, i] = self._z[i]
        return J.tolist()

    def set_tweights(self, W):  # vf:305
        self.tweights = np.asarray(W, dtype=float)

    def set_jweights(self, W):  # vf:309
        self.jweights = np.asarray(W, dtype=float)

    def getIKV(self, vel, rot):  # vf:461 -- weighted damped least squares (north_star)
        J = np.array(self.jac_list())
        Wy, Wq = self.tweights, self.jweights
        Jw = Wy @ J @ Wq
        A = Jw @ Jw.T + self.lam ** 2 * np.eye(6)
        y = np.linalg.solve(A, Wy @ np.concatenate([np.asarray(vel, float), np.asarray(rot, float)]))
        return (Wq @ Jw.T @ y).tolist()

    @staticmethod
    def distToCenter(limit, q):  # debug_jointlimits:67
        mid, half = 0.5 * (limit[0] + limit[1]), 0.5 * (limit[1] - limit[0])
        return abs(q - mid) / half


# ---------------------------------------------------------------------------------------------
# scripts/nullspace restated (restrict :75-79, nullspace :95-107, move_in_nullspace :110-117,
# check_limits :120-131).  Same numpy calls as the reference, state kept in an object instead of
# module globals (:91-92).
# ---------------------------------------------------------------------------------------------
def restrict(P, J):
    pJ = np.asarray(P) @ np.asarray(J)
    return np.eye(pJ.shape[1]) - pinv(pJ) @ pJ


class Nullspace:
    def __init__(self, nJoints):
        self.nJoints = nJoints
        self.sig = [1] * nJoints
        self.lastvec = np.zeros((nJoints, nJoints))

    def nullspace(self, P, J):
        B = restrict(P, J)
        u, s, vh = svd(B.T)
        u = u.copy()
        i = 0
        while i < self.nJoints and s[i] >= 1e-8:
            if norm(self.sig[i] * u[:, i] - self.lastvec[:, i]) > norm(self.sig[i] * u[:, i] + self.lastvec[:, i]):
                self.sig[i] = -self.sig[i]
            u[:, i] = u[:, i] * self.sig[i]
            self.lastvec[:, i] = u[:, i]
            i += 1
        return u[:, 0:i].T

    def move_in_nullspace(self, P, J, control):
        ns = self.nullspace(P, J)
        n = min(self.nJoints, len(control), ns.shape[0])
        qdot = np.zeros(self.nJoints)
        for i in range(n):
            qdot = qdot + ns[i, :] * control[i]
        return [float(qdot[i]) for i in range(self.nJoints)], ns.shape[0]


def check_limits(q, qdot, limits, scale=0.3):
    margin = 0.0
    n = len(limits)
    for i in range(n):
        d = q[i] + scale * qdot[i]
        if d < limits[i][0] + margin or d > limits[i][1] - margin:
            return [0.0] * n, True
    return list(qdot), False


# ---------------------------------------------------------------------------------------------
# src/command_mixer.py restated (CommandMixer :32-82) with an injectable clock
# ---------------------------------------------------------------------------------------------
class CommandMixer:
    def __init__(self, ports, weight_port, n, guard_time, weights, clock=None):
        import time as _time
        self.clock = clock if clock is not None else _time.time
        self.nChannels = n
        self.ports = ports
        self.weight_port = weight_port
        self.weights = [0.0] * len(ports) if len(ports) != len(weights) else weights
        self.guard_time = guard_time
        self.last_command = [[0.0] * n for _ in ports]
        self.last_command_time = [self.clock()] * len(ports)

    def read(self):
        if self.weight_port:
            bottle = self.weight_port.read(False)
            if bottle:
                for i in range(min(bottle.size(), len(self.ports))):
                    self.weights[i] = bottle.get(i).asDouble()
        for p in range(len(self.ports)):
            bottle = self.ports[p].read(False)
            if bottle and bottle.size() == self.nChannels:
                self.last_command_time[p] = self.clock()
                self.last_command[p] = [bottle.get(i).asDouble() for i in range(self.nChannels)]
            elif self.clock() - self.last_command_time[p] > self.guard_time:
                self.last_command[p] = [0.0] * self.nChannels
        result = [0.0] * self.nChannels
        for v, w in zip(self.last_command, self.weights):
            for i in range(len(v)):
                result[i] += v[i] * w
        return result


def get_weight_matrix(wbottle, n_vars):  # vf:164-179; pinned by tests/golden/weights_golden.npz
    """Diagonal (n_vars, n_vars) weight matrix from a ('t' | 'j', w_0 ... w_{n_vars-1}) bottle, None for any other size."""
    if wbottle.size() != n_vars + 1:
        return None
    W = np.zeros((n_vars, n_vars))
    for i in range(n_vars):
        W[i, i] = wbottle.get(i + 1).asDouble()
    return W


def limiter(qdot, max_vel):  # LWR_Bridge.set_vel, bridge:188-195; pinned by tests/golden/bridge_golden.npz
    lead = max(abs(v) for v in qdot)
    ratio = max_vel / lead if lead > max_vel else 1.0
    return [v * ratio for v in qdot], lead > max_vel


def lwr_command(qdot_lim, last_q, last_qcmded, direct_control):  # LWR_Bridge.set_vel, bridge:198-203
    cmd = len(qdot_lim) * [0.0]
    for i in range(len(qdot_lim)):
        if direct_control:
            cmd[i] = qdot_lim[i]
        else:
            cmd[i] = -last_qcmded[i] + last_q[i] + qdot_lim[i]
    return cmd


def joint_p_controller(ref, q, limits, kp, delta):
    """scripts/joint_p_controller: check_limits (:89-99), outqdot = (ref - q) * kp (:127-128) and the
    /at_goal flag (:134-138; the error is compared signed, as written there)."""
    ref_out = list(ref)
    for i in range(len(limits)):
        if ref[i] < limits[i][0]:
            ref_out[i] = limits[i][0]
        elif ref[i] > limits[i][1]:
            ref_out[i] = limits[i][1]
    error = np.asarray(ref_out, dtype=float) - np.asarray(q, dtype=float)
    outqdot = error * kp
    all_reached = True
    for x in error:
        all_reached = all_reached and bool(x < delta)
    return outqdot.tolist(), all_reached


# ---------------------------------------------------------------------------------------------
# one control cycle of one arm, in the order of the reference processes
# ---------------------------------------------------------------------------------------------
ST_NAN, ST_LIMIT_STOP, ST_NULL_AMBIGUOUS, ST_LIMITED = 1, 2, 4, 8
F_NULLSPACE, F_JOINT_LIMIT_TASK, F_MIXER, F_LIMITER = 1, 2, 4, 8


class ArmCycle:
    """State that one arm's vf + nullspace + bridge processes keep between cycles."""

    def __init__(self, chain_B, jtype, q_lo, q_hi, params):
        self.p = dict(params)
        self.lafik = Lafik(chain_B, jtype, q_lo, q_hi, lam=self.p["lambda"])
        self.lafik.set_tweights(np.diag(self.p["wy"]))
        self.lafik.set_jweights(np.diag(self.p["wq"][: self.lafik.numJnts]))
        self.vfDB = vectorFieldLibrary(self.p["rot_slowdown"])
        self.vectorFields = {}
        self.totalVF, self.totalSF = build_total_field({}, self.vfDB)
        self.oldtoolFrame = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]  # vf:154
        self.ns = Nullspace(self.lafik.numJnts)

    def set_fields(self, vectorFields):
        self.vectorFields = dict(vectorFields)
        self.totalVF, self.totalSF = build_total_field(self.vectorFields, self.vfDB)

    def cycle(self, q, tool=None, null_control=None, ext_cmd=None):
        p, lafik, n = self.p, self.lafik, self.lafik.numJnts
        flags = p["flags"]
        status = 0
        # ---- scripts/vf:311-347 ----
        lafik.jntsList = q
        kdlframe = lafik.kdlframe
        toolFrame = tool if (tool is not None and len(tool) == 16) else self.oldtoolFrame
        self.oldtoolFrame = toolFrame
        newkdlframe = kdlframe * listToKdlFrame(toolFrame)
        dframe = diff(newkdlframe, kdlframe)
        frame = kdlFrameToList(newkdlframe)
        velvector = self.totalVF.getVector(frame)
        scalars = self.totalSF.getScalar(frame)
        velPos = p["speed_scale"] * scalars[0] * velvector[0:3]
        velRot = p["speed_scale"] * scalars[1] * velvector[3:6]
        # ---- scripts/vf:455-461 ----
        tw = Twist(velPos, velRot).RefPoint(dframe.vel)
        qdot_vf = lafik.getIKV(tw.vel, tw.rot)
        # ---- scripts/nullspace:162-184 ----
        qdot_null = [0.0] * n
        if flags & F_NULLSPACE:
            control = list(null_control) if null_control is not None else [0.0] * 4
            J = np.array(lafik.jac_list())
            qn, rank = self.ns.move_in_nullspace(np.eye(6), J, control)
            if rank >= 2:
                status |= ST_NULL_AMBIGUOUS
            if flags & F_JOINT_LIMIT_TASK:
                lo = np.array([l[0] for l in lafik.joint_limits])
                hi = np.array([l[1] for l in lafik.joint_limits])
                mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
                z = -p["jl_gain"] * (np.asarray(q, float) - mid) / half ** 2
                qn = (np.asarray(qn) + restrict(np.eye(6), J) @ z).tolist()
            qn, stopped = check_limits(list(q), qn, lafik.get_limits(), p["lookahead"])
            if stopped:
                status |= ST_LIMIT_STOP
            qdot_null = [v * p["null_gain"] for v in qn]
        # ---- src/command_mixer.py:78-82 ----
        if flags & F_MIXER:
            cmds = [list(qdot_vf), list(qdot_null)]
            for ch in range(4):
                cmds.append(list(ext_cmd[ch]) if ext_cmd is not None else [0.0] * n)
            out = [0.0] * n
            for v, w in zip(cmds, p["mix_w"]):
                for i in range(n):
                    out[i] += v[i] * w
        else:
            out = list(qdot_vf)
        if flags & F_LIMITER:
            out, lim = limiter(out, p["max_vel"])
            if lim:
                status |= ST_LIMITED
        if any(math.isnan(v) for v in out):
            status |= ST_NAN
        qdist = [lafik.distToCenter(l, qi) for qi, l in zip(q, lafik.joint_limits)]  # debug_jointlimits:65-67
        return dict(qdot_vf=np.array(qdot_vf), qdot_null=np.array(qdot_null), qdot_out=np.array(out),
                    pose=np.array(frame), pose_nt=np.array(kdlFrameToList(kdlframe)),
                    v6=np.concatenate([velPos, velRot]), qdist=np.array(qdist), status=status)


# ---------------------------------------------------------------------------------------------
# observers
# ---------------------------------------------------------------------------------------------
class TrackingError:
    """The tracking-error estimator inside scripts/vf (vf:156-160,349-428), one arm."""
    cmd_buffer_size, frame_list_size, check_delay = 4, 5, 4  # vf:158-160

    def __init__(self):
        self.frame_list, self.cmd_buffer = [], []

    def update(self, newkdlframe, velPos, velRot):
        """Returns the 8 values of the /track_error bottle, or None while the history is short."""
        self.cmd_buffer.append([np.asarray(velPos, float), np.asarray(velRot, float)])
        if len(self.cmd_buffer) > self.cmd_buffer_size:
            self.cmd_buffer.pop(0)
        self.frame_list.append(newkdlframe)
        if len(self.frame_list) <= self.frame_list_size:
            return None
        self.frame_list.pop(0)
        ext = diff(self.frame_list[-2], self.frame_list[-1])
        cmd = self.cmd_buffer[len(self.cmd_buffer) - self.check_delay]
        ext_vel_mag, cmd_vel_mag = norm(ext.vel), norm(cmd[0])
        ext_vel = ext.vel / ext_vel_mag if ext_vel_mag > 0 else np.array([1.0, 0, 0])
        cmd_vel = cmd[0] / cmd_vel_mag if cmd_vel_mag > 0 else np.array([1.0, 0, 0])
        vel_diff_angle = abs(math.acos(min(1.0, max(-1.0, float(cmd_vel @ ext_vel)))))
        ext_rot_mag, cmd_rot_mag = norm(ext.rot), norm(cmd[1])
        ext_rot = ext.rot / ext_rot_mag if ext_rot_mag > 0 else np.array([1.0, 0, 0])
        cmd_rot = cmd[1] / cmd_rot_mag if cmd_rot_mag > 0 else np.array([1.0, 0, 0])
        rot_diff_angle = abs(math.acos(min(1.0, max(-1.0, float(cmd_rot @ ext_rot)))))
        loop_freq, tracking_th = 150, 0.10  # vf:408-409
        ev, er = ext_vel_mag * loop_freq, ext_rot_mag * loop_freq
        cr, cv = cmd_rot_mag / 5.0, cmd_vel_mag * 1.0
        ext_int_diff = abs((cv + cr) - (ev + er))
        return np.array([vel_diff_angle, rot_diff_angle, ev, er, cv, cr, ext_int_diff, float(ext_int_diff < tracking_th)])


def object_distances(pose16, objects):
    """scripts/monitor_distance:148-167: for every object of the dictionary (id -> 16-list), in dictionary
    order, [id, xyz distance, orientLength in degrees] -- the entries of one /dmonitor/distOut bottle."""
    F = listToKdlFrame(pose16)
    out = []
    for i in objects:
        G = listToKdlFrame(objects[i])
        out.append([i, norm(F.p - G.p), 180.0 * norm(rot_log(F.M, G.M)) / math.pi])
    return out


def goal_distance(pose16, goal16):
    """The object-0 entry of /dmonitor/distOut: xyz distance and orientLength in degrees
    (scripts/monitor_distance:76-84,161-167)."""
    F, G = listToKdlFrame(pose16), listToKdlFrame(goal16)
    return norm(F.p - G.p), 180.0 * norm(rot_log(F.M, G.M)) / math.pi
