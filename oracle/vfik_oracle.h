/* vfik_oracle.h -- CPU restatement of vfclik's per-cycle control path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call
 * anything in oracle/.  The product (vfclik_amd/, libvfik_hip.so) never does.
 *
 * Pinning status (DESIGN.md section "Oracle"):
 *   - vfo_mix              : pinned bit-exact by tests/golden/mixer_golden.npz (outputs of the
 *                            reference's CommandMixer.read run in the build container)
 *   - vfo_restrict, vfo_nullspace_basis, vfo_move_in_nullspace, vfo_check_limits :
 *                            pinned by tests/golden/nullspace_golden.npz (outputs of the
 *                            reference's scripts/nullspace functions)
 *   - FK / Jacobian / field primitives / normCart / getIKV / distToCenter : PARITY UNPINNED.
 *     Their arithmetic lives in vfl, arcospyu and PyKDL, none of which is in the reference tree,
 *     on this disk, or version-pinned by the reference (setup.py:6-20).  The definitions here are
 *     this build's own, written from the reference's call sites; see DESIGN.md section "Spec".
 */
#ifndef VFIK_ORACLE_H
#define VFIK_ORACLE_H

#include "../include/vfik_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A3: forward kinematics; T is row-major 3x4; z/o (n x 3) are joint axes / origins in base. */
void vfo_fk(const vfik_chain* c, const double* q, double T[12], double* z, double* o);
/* geometric Jacobian at the flange, base coordinates, row-major 6 x n */
void vfo_jacobian(const vfik_chain* c, const double* q, double* J, double T[12]);
/* A5: summed field at pose F (row-major 3x4): vec6 after normCart, sc2 = product of scalar fields */
void vfo_field_eval(const vfik_field* f, int nf, const double F[12], double rot_slowdown,
                    double vec6[6], double sc2[2]);
/* A7: weighted damped least squares */
void vfo_ikv(const double* J, int n, const double tw[6], const double wy[6], const double* wq,
             double lambda, double* qdot);
/* A10: B = I - pinv(J) J  (P = I6), row-major n x n */
void vfo_restrict(const double* J, int n, double* Bout);
/* A11: stateful basis; returns r = number of rows written to basis (r x n, row-major) */
int vfo_nullspace_basis(const double* J, int n, double* lastvec /*n x n, column i = vector i*/,
                        int* sig /*n*/, double* basis);
/* A12 */
void vfo_move_in_nullspace(const double* J, int n, const double* control, int ncontrol,
                           double* lastvec, int* sig, double* qdot, int* rank_out);
/* A13: returns 1 when the command was zeroed */
int vfo_check_limits(const double* q, double* qdot, const double* lo, const double* hi, int n,
                     double scale);
/* A14 */
void vfo_dist_to_center(const double* q, const double* lo, const double* hi, int n, double* d);
/* A15: result_i = sum_k cmd[k][i] * w[k], accumulated in channel order, no contraction */
void vfo_mix(const double* cmd /*K x n*/, const double* w, int K, int n, double* out);
/* bridge limiter (bridge:188-195); returns 1 when scaled */
int vfo_limiter(double* qdot, int n, double max_vel);
/* joint P controller (joint_p_controller:89-99,124-128,134-138): out = kp * (clamp(ref, lo, hi) - q);
 * returns 1 when every signed error is below delta (the /at_goal flag) */
int vfo_joint_p(const double* ref, const double* q, const double* lo, const double* hi, int n,
                double kp, double delta, double* out);
/* LWR command form (bridge:199-203): cmd = qdot_lim when direct_control, else -q_cmded + q + qdot_lim */
void vfo_lwr_cmd(const double* qdot_lim, const double* q, const double* q_cmded, int n, int direct, double* cmd);

/* per-arm persistent state of the nullspace module (nullspace:91-92) */
typedef struct vfo_state {
    double lastvec[VFIK_MAX_JOINTS * VFIK_MAX_JOINTS];
    int sig[VFIK_MAX_JOINTS];
} vfo_state;
void vfo_state_init(vfo_state* s, int n);

typedef struct vfo_out {
    double* qdot_vf;    /* n   : /vectorField/qdotOut (vf:462-466) */
    double* qdot_null;  /* n   : /nullspace/qdotout (nullspace:180-184), gain applied */
    double* qdot_out;   /* n   : mixed (+limited) command, or qdot_vf when the mixer is off */
    double* pose;       /* 16  : /pose  (T_tip, vf:341) */
    double* pose_nt;    /* 16  : /pose_no_tool (T_ee, vf:342) */
    double* v6;         /* 6   : speedScale*scalars*normCart(sum)  (vf:346-347) */
    double* qdist;      /* n   : distToCenter (debug_jointlimits:65-67), NOT multiplied by 100 */
    int* status;
} vfo_out;

/* one control cycle of one arm: vf (A3-A8) [+ nullspace A10-A13] [+ mixer A15] [+ limiter] */
void vfo_cycle(const vfik_chain* c, const vfik_params* p, const double tool[16],
               const vfik_field* fields, int nfields, const double* q,
               const double* null_control /*4 or NULL*/, const double* ext_cmd /*4 x n or NULL*/,
               vfo_state* st /*may be NULL when VFIK_F_NULLSPACE is off*/, vfo_out* out);

/* batch driver (OpenMP over arms when built with -fopenmp).  All arrays are batch-major AoS:
 * q[B][n], tool[B][16] (tool_stride 0 = shared), fields[B][max_fields], nfields[B],
 * null_control[B][4] or NULL, ext_cmd[4][B][n] or NULL, outputs [B][..] or NULL.
 * q_lo / q_hi [B][n] or NULL: the arm's joint limits of THIS cycle, which the reference re-reads every cycle
 * (nullspace:167 rob.get_limits(), joint_p_controller:80 config.updateJntLimits(cur_pos)); NULL = the chain's.
 * active[B] or NULL: an arm with active[b] == 0 got no joint angles this cycle -- its loop body does not run
 * (vf:312-313, nullspace:162-163, debug_jointlimits:61): no output row written, state untouched. */
void vfo_cycle_batch(const vfik_chain* c, const vfik_params* p, int B, const double* tool,
                     int tool_stride, const vfik_field* fields, int max_fields, const int* nfields,
                     const double* q, const double* null_control, const double* ext_cmd,
                     vfo_state* st, double* qdot_vf, double* qdot_null, double* qdot_out,
                     double* pose, double* pose_nt, double* v6, double* qdist, int* status,
                     int nthreads, const double* q_lo, const double* q_hi, const int* active);
int vfo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
