/* vfik.h -- C-ABI of the MI355X-native batched closed-loop IK path (libvfik_hip.so).
 *
 * The reference (arcoslab/vfclik) has no FFI: its per-cycle path is a set of Python processes
 * (scripts/vf, scripts/nullspace, scripts/debug_jointlimits, src/command_mixer.py inside
 * scripts/bridge) that exchange YARP bottles.  This header is the boundary a maintainer would bind
 * (ctypes) to replace the body of those loops for B arms at once.  Each entry
 * point names the reference code it stands in for (file:line under the reference tree).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no torch / HIP types in any signature (a HIP stream is
 *     passed as void*).
 *   - every function returns 0 on success or a negative VFIK_E_* code; vfik_last_error() gives
 *     the text for the calling thread.  Bad input is never fatal -- like the reference, which
 *     logs and ignores malformed bottles (scripts/vf:176-179,204-207,264-266).
 *   - per-arm numeric trouble (NaN, limit stop, ...) is reported in status[B] (VFIK_ST_*),
 *     never by failing the call (src/command_mixer.py:71-75 only prints).
 *   - batch arrays are batch-major: q[B][n], qdot[B][n], pose[B][16], ...  Element type is the
 *     handle's io dtype (float for 32, double for 64).  Arithmetic is always float64.
 *   - one host thread per handle; one handle per device (shards are independent).
 *   - there is no CPU implementation behind this ABI.  Without a GPU vfik_create fails.
 */
#ifndef VFIK_H
#define VFIK_H

#include <stddef.h>
#include <stdint.h>

#include "vfik_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VFIK_ABI_VERSION 5

enum {
    VFIK_OK = 0,
    VFIK_E_ARG = -1,         /* bad argument (size, NULL, range); nothing was changed */
    VFIK_E_HIP = -2,         /* a HIP runtime call failed */
    VFIK_E_UNSUPPORTED = -3, /* e.g. a joint count the library was not built for */
    VFIK_E_STATE = -4        /* call order (chain not set, ...) */
};

typedef struct vfik_handle vfik_handle;

int vfik_abi_version(void);
/* sizeof(vfik_field), sizeof(vfik_chain), sizeof(vfik_params), sizeof(vfik_io) as this library was built:
 * a binding compares them with its own mirrors before the first call */
void vfik_struct_sizes(size_t out[4]);
const char* vfik_last_error(void);
/* number of visible HIP devices (0 without a GPU; never initialises a context) */
int vfik_device_count(void);
/* joint counts this build has kernels for, as a bit mask (bit n set = n joints supported) */
uint32_t vfik_supported_joints(void);

/* One handle = the state that the reference spreads over one vf, one nullspace, one
 * debug_jointlimits and one bridge/CommandMixer process PER ARM, for `batch` arms on one GPU:
 * field sets (vf:145), sticky tool frames (vf:154), IK weights, nullspace sign state
 * (nullspace:91-92), mixer weights and last commands (command_mixer.py:39-44).
 * io_dtype: 32 or 64.  max_slots: capacity of the per-arm field list in device slots
 * (a repeller takes 1, hemisphere/funnel 2, an extra attractor 3; the first attractor is free). */
vfik_handle* vfik_create(int device, int io_dtype, int n_joints, int max_slots, int batch);
void vfik_destroy(vfik_handle* h);

/* Run on the caller's HIP stream (e.g. torch's current stream) instead of the handle's own.
 * The pointer is used as given: NULL selects the device's default (null) stream. */
int vfik_set_stream(vfik_handle* h, void* hip_stream);

/* Lafik(config) (vf:153, nullspace:60, debug_jointlimits:56): chain geometry + joint limits. */
int vfik_set_chain(vfik_handle* h, const vfik_chain* chain);

/* speedScale (/max_vel, vf:197-207), 't'/'j' weights (/weight, vf:295-309), nullspace gain and
 * look-ahead (nullspace:62,121), mixer weights (/bridge/weight, command_mixer.py:48-53), bridge
 * max_vel (bridge:612-623), feature flags.  Range checks of the ports are done by the host layer. */
int vfik_set_params(vfik_handle* h, const vfik_params* p);

/* Per-arm speedScale: what each arm's vf keeps after a /max_vel message (vf:197-207).  values[n_arms]
 * for arms [first_arm, first_arm + n_arms).  vfik_set_params.speed_scale writes one value to all arms. */
int vfik_set_speed_scale(vfik_handle* h, int first_arm, int n_arms, const double* values);

/* /tool (vf:321-326): 16 doubles row-major, shared by the batch (per_arm = 0) or tool16[B][16].  ONE tool for the batch keeps the
 * launches of an all-revolute chain on the kernels built for such chains (the lean and publishing-lean float32-I/O
 * variants apply it themselves); per-arm tools take the general variants.  A per-arm array whose rows are all equal IS a
 * shared tool and is stored as one (with the values rounded to the I/O type, as the per-arm image would hold them). */
int vfik_set_tool(vfik_handle* h, const double* tool16, int per_arm);

/* Field sets of arms [first_arm, first_arm + n_arms): the result of the add/remove bookkeeping of
 * vf:209-275 -- fields[n_arms][max_fields] in the reference's parameter layouts, counts[n_arms].
 * Packing (ascending id; lowest-id attractor -> goal block) happens here, on the host, only when a
 * message arrived, exactly when the reference rebuilds totalVF (vf:276-293). */
int vfik_set_fields(vfik_handle* h, int first_arm, int n_arms, const vfik_field* fields,
                    int max_fields, const int32_t* counts);

/* Per-arm IK weights: what each arm's vf process keeps after a /weight message (vf:164-179,295-309): 't' + 6
 * task-space weights -> wy[n_arms][6], 'j' + n joint-space weights -> wq[n_arms][n]; either may be NULL
 * (unchanged).  Arms never written use vfik_params.wy / wq; a later vfik_set_params that CHANGES wy or wq
 * is batch-wide again and replaces every arm's own weights.  A call for the WHOLE batch with both arrays whose rows are all equal is a
 * batch-wide setting too (stored in the handle's parameters, the arms' own weights dropped): batch-wide weights keep the launches of an
 * all-revolute chain of up to 7 joints on the kernels built for it, per-arm weights take the general variants. */
int vfik_set_arm_weights(vfik_handle* h, int first_arm, int n_arms, const double* wy, const double* wq);

/* Per-arm mixer weights, w[n_arms][6]: what each arm's bridge keeps after a /bridge/weight message
 * (command_mixer.py:48-53; handlers send [cart, null, joint, 0], handlers.py:189-204).  NULL returns every
 * arm's weights to the batch-wide vfik_params.mix_w (per-arm limiter speeds of vfik_set_max_vel stay).  A
 * vfik_set_params that CHANGES vfik_params.mix_w writes the new weights to every arm.  While every arm's bridge state (these weights and the
 * limiter speed of vfik_set_max_vel) is the same, launches read it from the batch constants as if it had never been set per arm, and keep
 * the kernel variants without per-arm options (every handler sends the same [cart, null, joint, 0] at start-up: handlers.py:189-204,481-497). */
int vfik_set_mixer_weights(vfik_handle* h, int first_arm, int n_arms, const double* w);

/* Per-arm limiter speed: what each arm's bridge keeps after a /bridge/max_vel message (bridge:612-623; the
 * reference accepts 0 <= v <= config.max_vel, the caller applies that rule).  Used with VFIK_F_LIMITER.  A
 * vfik_set_params that CHANGES vfik_params.max_vel writes the new value to every arm. */
int vfik_set_max_vel(vfik_handle* h, int first_arm, int n_arms, const double* values);

/* Last command of mixer channel 2..5 (jointcmd, mechanismcmd, xtra1cmd, xtra2cmd; bridge:593-596)
 * for the whole batch: host array cmd[B][n] in the io dtype, or NULL to zero the channel, which is
 * also what the watchdog does after guard_time of silence (command_mixer.py:64-66). */
int vfik_set_ext_cmd(vfik_handle* h, int channel, const void* cmd_host);

/* Forget the nullspace sign memory (sig = 1, lastvec = 0; nullspace:91-92). */
int vfik_reset_state(vfik_handle* h);

/* Buffers of one control cycle.  NULL = not wanted / not supplied. */
typedef struct vfik_io {
    const void* q;            /* in  [B][n]   /qIn, /nullspace/qin, /debug/qin (vf:312, nullspace:162) */
    const void* null_control; /* in  [B][4]   /nullspace/control (nullspace:169-173); NULL = zeros.  HONOURED ONLY where the
                                 nullspace is one-dimensional (7 joints at a regular pose: element 0 scales the unique
                                 basis vector, nullspace:110-117).  With nullity >= 2 -- every chain of 8+ joints, a
                                 7-joint arm at a rank-deficient pose -- the reference moves along whatever basis LAPACK's
                                 SVD returns inside the nullspace, which cannot be restated: the arm then gets
                                 VFIK_ST_NULL_AMBIGUOUS, /control contributes nothing, and qdot_null carries the
                                 joint-limit task alone.  What IS reproduced there is the subspace: the reference's
                                 command lies in null(J) and the projector used here leaves it unchanged
                                 (tests/test_oracle_golden.py, the reference's own n = 14 outputs) */
    void* qdot_vf;            /* out [B][n]   /vectorField/qdotOut (vf:462-466) */
    void* qdot_null;          /* out [B][n]   /nullspace/qdotout (nullspace:180-184) */
    void* qdot_out;           /* out [B][n]   mixed (+limited) command (bridge:626); = qdot_vf without the mixer */
    void* pose;               /* out [B][16]  /pose (vf:341) */
    void* pose_nt;            /* out [B][16]  /pose_no_tool (vf:342) */
    void* v6;                 /* out [B][6]   field twist before RefPoint (vf:346-347; /vector_out) */
    void* qdist;              /* out [B][n]   distToCenter (debug_jointlimits:65-67), not x100 */
    int32_t* status;          /* out [B]      VFIK_ST_* */
    void* goal_dist;          /* out [B][2]   xyz distance and rotation angle in DEGREES to the goal: the object-0
                                 entry of /dmonitor/distOut (monitor_distance:76-84,161-172) */
    const void* q_ref;        /* in  [B][n]   /jpctrl/ref (joint_p_controller:113-118); NULL = no joint controller.
                                 An arm whose row starts with NaN has no controller either: its channel 2 stays the
                                 external /bridge/jointcmd command of vfik_set_ext_cmd.
                                 With VFIK_F_MIXER the controller's output kp*(clamp(ref,limits) - q)
                                 (joint_p_controller:89-99,124-128) IS mixer channel 2 (/bridge/jointcmd,
                                 joint_p_controller:78) and an external channel-2 command is not read */
    const void* q_cmded;      /* in  [B][n]   the LWR's echo of its last commanded position (bridge:168-172);
                                 NULL = velocity command.  With it qdot_out is the LWR command form
                                 -q_cmded + q + qdot_lim (bridge:199-203) unless all mixer weights of the arm
                                 are 0 ("direct_control", bridge:604).  vfik_step only */
    /* ---- ABI 3 ---- */
    const int32_t* active;    /* in  [B]      fresh-q gate, NULL = every arm.  The reference advances an arm only when that
                                 arm's own joint angles arrived (vf:312-313, nullspace:162-163, debug_jointlimits:61): an
                                 arm with active[b] == 0 publishes NOTHING this cycle -- no output row of it is written
                                 (status included) and its nullspace sign memory (nullspace:91-92) stays as it was */
    const void* q_lo;         /* in  [B][n]   joint limits of THIS cycle, per arm; NULL (both) = the chain's static limits. */
    const void* q_hi;         /*              The reference re-reads the limits every cycle because some robots' limits
                                 depend on the configuration (nullspace:167 `rob.get_limits()`, joint_p_controller:80
                                 `config.updateJntLimits(cur_pos)`, :121-125).  Used by check_limits (nullspace:120-131), the
                                 joint controller's clamp (joint_p_controller:79-89), distToCenter (debug_jointlimits:66-67),
                                 the joint-limit task and the rollout's clamp.  lo < hi is the caller's business */
    void* q_ref_out;          /* out [B][n]   the joint controller's reference after its clamp: the reference KEEPS the clamped
                                 value (joint_p_controller:121 `ref = check_limits(ref, indata)`), so with limits that move
                                 a host feeds this back as the next cycle's q_ref.  Only with q_ref */
    /* ---- ABI 4: the observers of the cycle, advanced by the SAME call on the cycle's own device results (no host round
     * trip: the reference's vf computes its tracking error inside the loop body, vf:349-428, and monitor_distance reads
     * the pose vf just published, monitor_distance:148-172).  vfik_step / vfik_step_host / vfik_submit_host only ---- */
    void* track_error;        /* out [B][8]   the /track_error bottle (vf:418-427): vel_diff_angle, rot_diff_angle,
                                 ext_vel_mag_corr, ext_rot_mag_corr, cmd_vel_mag_corr, cmd_rot_mag_corr, ext_int_diff,
                                 arm_tracking; zeros until the 6th frame (vf:354).  Requesting it advances the handle's
                                 per-arm history by one frame (gated arms: nothing, as in vfik_track_error) */
    void* obj_dist;           /* out [B][n_objects][2]  xyz distance and rotation angle in DEGREES between the tool pose and
                                 every object frame of vfik_set_objects -- the entries of /dmonitor/distOut
                                 (monitor_distance:156-167).  Needs vfik_set_objects */
} vfik_io;

/* One control cycle for the whole batch -- the loop bodies of vf:311-466, nullspace:162-184,
 * debug_jointlimits:61-73 and command_mixer.py:78-82 (+ bridge:188-195) in ONE kernel launch.
 * Device pointers; asynchronous on the handle's stream. */
int vfik_step(vfik_handle* h, const vfik_io* io);
/* Same with host pointers: copies in, runs, copies out, synchronises. */
int vfik_step_host(vfik_handle* h, const vfik_io* io);
int vfik_sync(vfik_handle* h);

/* Pipelined host path.  The reference's modules exchange Python lists over ports every cycle
 * (vf:312-315,462-466): for a host that keeps q and qdot in its own memory the copies, not the kernel,
 * bound the rate.  vfik_submit_host is vfik_step_host without the wait; outputs are in host memory after
 * vfik_wait(ticket); up to 3 submissions are in flight; kernels run in submission order on the handle's
 * stream.  With PINNED buffers (vfik_host_alloc, or the caller's own hipHostMalloc / hipHostRegister / torch
 * pin_memory) the kernel writes qdot across PCIe itself (zero-copy); q is read the same way when nothing
 * else is in flight, and moved by the copy engine, overlapping the previous kernel, when something is.
 * With pageable buffers copy-in, kernel and copy-out are staged on three streams.  The io buffers must stay
 * untouched until vfik_wait.  Device pointers are accepted as well (then it is vfik_step + an event). */
void* vfik_host_alloc(vfik_handle* h, size_t bytes);
int vfik_host_free(vfik_handle* h, void* p);
int vfik_submit_host(vfik_handle* h, const vfik_io* io, long* ticket);
int vfik_wait(vfik_handle* h, long ticket);

/* Closed-loop rollout: n_cycles control cycles in ONE launch.  After each cycle the
 * commanded velocity (what io->qdot_out reports) is integrated, q <- q + dt * qdot_out -- the role of
 * the kinematic simulator `joint_sim` that `vfclik -s` wires behind the bridge (scripts/vfclik:99-103,
 * scripts/bridge:136-139) -- and optionally clamped to the joint limits.  io->q is the start
 * configuration; the outputs named in io are those of the LAST cycle; q_out[B][n] (may be NULL) receives
 * the joint angles after it.  Field sets, tools, weights and /control stay fixed during the launch,
 * as they do between two messages in the reference; the nullspace sign memory advances every cycle.
 * Chains of up to 7 joints run all cycles inside one kernel; longer chains (no registers left for
 * loop-carried state) run n_cycles single-cycle launches that integrate q on the way out -- same results.
 * Device pointers, asynchronous; vfik_rollout_host takes host pointers and synchronises. */
int vfik_rollout(vfik_handle* h, const vfik_io* io, int n_cycles, double dt, int clamp_to_limits, void* q_out);
int vfik_rollout_host(vfik_handle* h, const vfik_io* io, int n_cycles, double dt, int clamp_to_limits, void* q_out);

/* Tracking-error estimator of scripts/vf (vf:349-428) for the batch: feed it, once per cycle, the tool
 * poses and field twists that vfik_step produced (device pointers pose[B][16], v6[B][6]); out[B][8] gets
 * vel_diff_angle, rot_diff_angle, ext_vel_mag_corr, ext_rot_mag_corr, cmd_vel_mag_corr, cmd_rot_mag_corr,
 * ext_int_diff, arm_tracking (the /track_error bottle, vf:418-427); zeros until the 6th frame (vf:354).
 * Per-arm history (previous frame, last 4 commands) lives in the handle; vfik_track_reset clears it.
 * active[B] (device, may be NULL = every arm): the estimator sits inside vf's `if qInBottle` block (vf:312-313,349),
 * so an arm without fresh joint angles appends no frame and no command -- its history and its out row stay. */
int vfik_track_error(vfik_handle* h, const void* pose, const void* v6, void* out, const int32_t* active);
int vfik_track_reset(vfik_handle* h);

/* Field probe of scripts/vf (vf:469-503, /pose_in -> /vector_out, "for visualizing"): every arm's field set
 * evaluated at a pose handed in instead of the arm's forward kinematics; device pose[B][16] -> device
 * v6[B][6] = speedScale * scalars * normCart(sum) (vf:491-494).  Asynchronous on the handle's stream. */
int vfik_probe_field(vfik_handle* h, const void* pose, void* v6);

/* Distance monitor of scripts/monitor_distance (monitor_distance:76-84,148-167) for the batch: device
 * pose[B][16] (what vfik_step wrote to io->pose), device frames[B][max_objects][16] (the object frames of
 * /dmonitor/objectsIn, object_feeder:215-227,306-315; unused slots may hold anything finite), device
 * out[B][max_objects][2] = xyz distance, rotation angle in DEGREES -- one /dmonitor/distOut entry each
 * (monitor_distance:161-167).  Asynchronous on the handle's stream. */
int vfik_object_distances(vfik_handle* h, const void* pose, const void* frames, int max_objects, void* out);

/* The object frames of the distance monitor (scripts/monitor_distance keeps a dictionary id -> frame fed by
 * /dmonitor/objectsIn, monitor_distance:72,111-129; object_feeder:215-227,306-315) as device state of the handle, rewritten
 * only when a message changed them: host frames[n_arms][n_objects][16] doubles (row-major 4x4; unused slots: any finite
 * frame, e.g. identity) for the arms [first_arm, first_arm + n_arms).  n_objects (1..4096) is one number for the batch; a
 * call with another n_objects than the handle holds must cover every arm.  io->obj_dist of a cycle call then gets every
 * arm's distances to its objects, computed on the device from the pose of that cycle.  Arms gated off by io->active get no new
 * pose, hence no new distances either: their rows of obj_dist (like those of track_error) keep the caller's content (ABI 5; until
 * then they were computed from the arm's previous pose).  The device buffers behind io->track_error / io->obj_dist are allocated
 * at the first call that asks for them -- a host that CAPTURES vfik_step into a hipGraph makes one such call outside the capture
 * first (VFIK_E_STATE otherwise). */
int vfik_set_objects(vfik_handle* h, int first_arm, int n_arms, const double* frames, int n_objects);

/* CommandMixer.read's weighted sum on its own (command_mixer.py:78-82): device cmds[K][B][n],
 * host weights[K], device out[B][n].  Bit-exact with the reference's left-to-right sum. */
int vfik_mix(vfik_handle* h, const void* cmds, const double* weights, int K, void* out);

/* Device memory for hosts that do not bring their own allocator (torch tensors work as well). */
void* vfik_dev_alloc(vfik_handle* h, size_t bytes);
int vfik_dev_free(vfik_handle* h, void* p);
int vfik_memcpy_h2d(vfik_handle* h, void* dst_dev, const void* src_host, size_t bytes);
int vfik_memcpy_d2h(vfik_handle* h, void* dst_host, const void* src_dev, size_t bytes);

/* hipEvent timing of `steps` back-to-back vfik_step launches on the handle's stream, after
 * `warmup` untimed ones: total elapsed ms -> *ms_total.  (bench.py: kernel time on the stream the
 * kernel really runs on.) */
int vfik_time_steps(vfik_handle* h, const vfik_io* io, int warmup, int steps, float* ms_total);

/* Small batches -- what vfclik itself runs is a handful of arms, each with its vf, nullspace and debug process and the bridge's
 * mixer (scripts/vfclik:88-105).  Launches the eight-lanes-per-arm kernel serves (revolute chain of up to 7 joints; no tool or ONE tool,
 * and IK weights shared by the batch; goal + integer-order decay repellers; with or without the nullspace module, joint-limit task, mixer, limiter,
 * /control; outputs qdot_out, qdot_vf, qdot_null, pose, pose_nt, qdist, status; no gate, no per-arm limits or weights) take it
 * instead of one lane per arm up to the batch size where the same-box A/B stops winning (profiles/r04_latency_small_*.txt;
 * DESIGN.md section 5.8): 4096 arms when the per-cycle rows are published; launches that ask for qdot_out alone (with or without the
 * nullspace module) stay on one lane per arm since round 4.  vfik_set_small_batch_kernel sets ONE threshold for all three cases (0 = never; the environment variable
 * VFIK_SUB8_MAX_BATCH does the same when the handle is created).  vfik_small_batch_launches: how many launches took that kernel
 * so far. */
int vfik_set_small_batch_kernel(vfik_handle* h, int max_batch);
long vfik_small_batch_launches(vfik_handle* h);

/* introspection for tests / DESIGN.md: slots in use, bytes of device state */
int vfik_slots_in_use(vfik_handle* h);
/* Which field path the handle's current field sets select for a cycle launch (decided when they are packed, vfik_set_fields):
 * 1 = straight-line: every arm is goal + decay repellers with integer orders (what object_feeder sends for point obstacles,
 * object_feeder:317-334: order 5 throughout; since ABI 5 the orders may differ between obstacles and between arms, as in
 * old/README.old:75, `ObstacleP ... 0.05 20`, beside the feeder's own order-5 near-goal repeller -- vfik_mixed_orders); 2 = straight-line with an aux block: as 1, and arms may carry ONE funnel attractor and ONE hemisphere
 * repeller with integer decay orders -- the goalAndNormal scene (object_feeder:248-303: attractor + approach funnel + near-goal
 * repeller + obstacles) and a surface (ObstacleH, object_feeder:344-353);
 * 0 = general: anything else (further attractors, several funnels or hemispheres, fractional orders or orders >= 128), entry by entry. */
int vfik_field_path(vfik_handle* h);
/* 1 when every decay repeller of the batch carries the same safe distance and the same force -- what the object feeder sends
 * (0.001 and -10 for every point obstacle and for the near-goal repeller: object_feeder:301-302,323,331): on field path 1 the lean
 * launches then read one quad (x y z radius) per repeller instead of 24 bytes, the pair from the batch constants.  0 otherwise. */
int vfik_uniform_repellers(vfik_handle* h);
/* ABI 5.  1 when the decay repellers of the batch have integer orders that are not all the same: lean and publishing-lean
 * single-cycle launches of chains without a tool / weights then read one order byte per repeller beside the compact image and stay
 * on field path 1 / 2 (a wave whose 64 arms agree slot by slot pays scalar control flow only; a wave with an odd arm per-lane
 * selects); every other launch of such a batch (rollouts, per-arm options) takes the general path.  0 otherwise -- also with the
 * environment variable VFIK_MIXED_ORDERS=0, which restores the behaviour of ABI 4 (differing orders -> field path 0). */
int vfik_mixed_orders(vfik_handle* h);
/* ABI 5.  A counter that moves with every call that can change what a cycle launch bakes in at enqueue time -- the kernel variant and
 * its scalar arguments: field path, uniform / compact / order-plane image, slots in use, flags, PLAIN or not, per-arm option buffers
 * (tool, weights, mixer state, external commands), the small-batch thresholds -- i.e. every vfik_set_* call, vfik_reset_state and
 * vfik_set_small_batch_kernel.  A host that CAPTURED vfik_step / vfik_rollout launches into a hipGraph compares the value at capture
 * with the current one before a replay: a graph captured under another epoch may launch a stale variant (e.g. the uniform-image
 * kernel after one arm's (safe distance, force) pair changed) and must be re-captured.  Data the launch reads through pointers that
 * stay (q, the field images' CONTENT, nullspace state, external commands once allocated) needs no re-capture; the epoch moves anyway:
 * it is conservative. */
long vfik_launch_epoch(vfik_handle* h);
/* ABI 5, introspection.  1 when the chain set by vfik_set_chain matches a Denavit-Hartenberg pattern the lean float32-I/O kernels (and
 * the eight-lanes-per-arm kernel of small batches, either I/O type) are built for -- for 7 joints the KUKA LWR 4+ (vfclik's default robot, scripts/vfclik:42): a = 0 on every link, alpha = +-pi/2 on six,
 * d = 0 on three; for 14 joints two of them in series; for 6 joints the arm of vfclik_amd/robots.py -- and launches may take the
 * variants in which those links cost no arithmetic (all-revolute chain; no tool or ONE tool for the batch; IK weights shared by the batch); 0 otherwise: every
 * chain runs, the general DH form is the fallback.  VFIK_DH_PATTERN=0 in the environment switches the specialisation off. */
int vfik_dh_pattern(vfik_handle* h);
size_t vfik_device_bytes(vfik_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* VFIK_H */
