/* vfik_types.h -- plain-C data layouts shared by the C-ABI (include/vfik.h), the HIP
 * implementation (vfclik_amd/csrc) and the CPU oracle (oracle/).  No logic lives here.
 *
 * Every layout is the batched form of something the reference keeps per arm-process:
 *   vfik_field   <- one entry of vf's `vectorFields{id: [force, type, params]}`
 *                   (/root/reference/scripts/vf:246-258; parameter layouts by type from
 *                   scripts/object_feeder:229-354)
 *   vfik_chain   <- what `Lafik(config)` hides (scripts/vf:153): the serial chain, the joint
 *                   limits (scripts/nullspace:167, scripts/debug_jointlimits:66)
 *   vfik_params  <- speedScale (vf:134-137,197-207), IK weights (vf:295-309), nullspace gain
 *                   and look-ahead (nullspace:62,121), mixer weights (bridge:593-596),
 *                   bridge max_vel (bridge:69,188-193)
 */
#ifndef VFIK_TYPES_H
#define VFIK_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFIK_MAX_JOINTS 16
#define VFIK_MAX_PARAMS 17   /* type 1 carries frame16 + slow-down distance */
#define VFIK_MIX_CHANNELS 6  /* bridge:593-596: vectorfield, null, joint, mechanism, xtra1, xtra2 */
#define VFIK_NULL_CONTROLS 4 /* nullspace:137,144 "expects four-float-bottles" */

/* field primitive type codes = keys of vfl.vfl.vectorFieldLibrary() used by the reference
 * (vf:148,238; object_feeder:236,268,288,324,342) */
enum {
    VFIK_FIELD_NULL = 0,        /* seed field, no parameters (vf:148-151) */
    VFIK_FIELD_ATTRACTOR = 1,   /* frame16 + slowdown (object_feeder:236-241) */
    VFIK_FIELD_REPELLER = 2,    /* x y z radius safeDist order (object_feeder:326-333) */
    VFIK_FIELD_HEMISPHERE = 4,  /* x y z nx ny nz safeDist order (object_feeder:344-353) */
    VFIK_FIELD_FUNNEL = 5       /* x y z ax ay az cutAngle angleOrder cutDist distOrder (:270-279) */
};

/* One vector-field primitive of one arm, in the reference's own parameter layout. */
typedef struct vfik_field {
    int32_t id;     /* vf's dictionary key (object_feeder: 1 goal, 2 funnel, 3 near-goal, 4+k obstacles) */
    int32_t type;   /* VFIK_FIELD_* */
    double force;   /* >0 attracts, <0 repels (vf:236) */
    double p[VFIK_MAX_PARAMS];
} vfik_field;

/* Serial chain in z-normal form: T_ee(q) = B[0] * Jz(q_1) * B[1] * ... * Jz(q_n) * B[n],
 * Jz(q) = Rot_z(q) for jtype 0 (revolute), Trans_z(q) for jtype 1 (prismatic).
 * B[i] is a row-major 3x4 [R | p].  Any KDL-style chain (joint then fixed tip frame per
 * segment, arbitrary joint axis) is brought to this form on the host (vfclik_amd/chain.py). */
typedef struct vfik_chain {
    int32_t n;
    int32_t jtype[VFIK_MAX_JOINTS];
    double B[VFIK_MAX_JOINTS + 1][12];
    double q_lo[VFIK_MAX_JOINTS];
    double q_hi[VFIK_MAX_JOINTS];
} vfik_chain;

/* feature flags (vfik_params.flags) */
enum {
    VFIK_F_NULLSPACE = 1u << 0, /* run the nullspace module (launcher option --no_nullspace absent, vfclik:72-79) */
    VFIK_F_JOINT_LIMIT_TASK = 1u << 1, /* project -jl_gain*grad(Phi_limits) into the nullspace (north_star C5) */
    VFIK_F_MIXER = 1u << 2,     /* q_out = CommandMixer.read() over the 6 channels (command_mixer.py:78-82) */
    VFIK_F_LIMITER = 1u << 3    /* bridge velocity limiter on the mixed command (bridge:188-195) */
};

typedef struct vfik_params {
    double speed_scale;     /* vf speedScale, runtime range [0, 0.41] via /max_vel (vf:134,197-207) */
    double lambda;          /* DLS damping of getIKV (vf:461); build-defined default 0.1 */
    double rot_slowdown;    /* rad; rotational analogue of the goal's slow-down distance (build-defined) */
    double null_gain;       /* nullspace:62 gain = 0.5 */
    double lookahead;       /* nullspace:121 scale = 0.3 */
    double jl_gain;         /* gain of the joint-limit task (build-defined, C5) */
    double max_vel;         /* bridge max joint speed (bridge:69); used with VFIK_F_LIMITER */
    double wy[6];           /* task-space weights ('t' message, vf:301-305) */
    double wq[VFIK_MAX_JOINTS]; /* joint-space weights ('j' message, vf:306-309) */
    double mix_w[VFIK_MIX_CHANNELS]; /* mixer weights, initial [1,1,0,0,0,0] (bridge:596) */
    uint32_t flags;
    uint32_t reserved;
    double jp_kp;           /* joint P controller gain (joint_p_controller:55-56, config.jpctrl_kp; 1.5 in the source) */
    double jp_delta;        /* joint P controller "reached" threshold, rad (joint_p_controller:57: 0.087) */
} vfik_params;

/* per-arm status bits written by vfik_step */
enum {
    VFIK_ST_NAN = 1 << 0,            /* a NaN reached a joint command (command_mixer.py:71-75 only prints) */
    VFIK_ST_LIMIT_STOP = 1 << 1,     /* nullspace look-ahead crossed a joint limit -> null command zeroed (nullspace:120-131) */
    VFIK_ST_NULL_AMBIGUOUS = 1 << 2, /* nullity != 1: the reference's SVD basis is not unique; /control is NOT honoured for
                                        this arm in this cycle (a declared functional gap, see vfik_io.null_control) */
    VFIK_ST_LIMITED = 1 << 3,        /* bridge limiter scaled the command (bridge:190-191) */
    VFIK_ST_JOINT_AT_GOAL = 1 << 4   /* /jpctrl/at_goal (joint_p_controller:134-146): every ref_i - q_i < jp_delta
                                        (signed, as the reference compares it); only with io->q_ref */
};

#ifdef __cplusplus
}
#endif
#endif /* VFIK_TYPES_H */
