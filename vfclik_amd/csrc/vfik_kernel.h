// vfik_kernel.h -- kernel argument block shared by vfik_kernel.hip (device) and vfik_abi.cpp (host).
#pragma once
#include <cstddef>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vfik_types.h"

// joint counts the library is built for (one fully unrolled kernel each)
#ifndef VFIK_NJ_LIST
#define VFIK_NJ_LIST X(6) X(7) X(10) X(14)
#endif

namespace vfik {

// Device data layout (DESIGN.md "Data layout in HBM").  T = io dtype (float | double), B = batch,
// Bpad = B rounded up to 64.  A "quad" is T[4] (16 or 32 bytes); a quad PLANE is Bpad quads, one per
// arm, so that the 64 lanes of a wave read 1 KiB (2 KiB) of contiguous memory with one (two)
// 16-byte-per-lane request(s).
//   q, qdot_*, qdist      [B][n]      batch-major (the reference's bottles: n doubles per arm)
//   pose, pose_nt         [B][16]
//   goal                  4 planes    frame rows 0,1,2 | (present, slow-down, force, speedScale of the arm)
//   funnel                3 planes    the arm's funnel attractor (object_feeder:262-279: the approach cone of a goal with a normal), if it is
//                                     the only non-repeller entry of the set: (x y z ax | ay az cutAngle angleOrder | cutDist distOrder force present)
//   hemisphere            3 planes    likewise the arm's ONE hemisphere repeller (object_feeder:344-353): (x y z nx | ny nz safeDist order | force present - -)
//   slots_fast            3 ceil(S/2) planes  decay repellers only, two slots in three quads: (x0 y0 z0 r0 | s0 f0 x1 y1 | z1 r1 s1 f1);
//                                     what the straight-line field path reads (24 instead of 32 bytes a slot at float I/O)
//   orders                ceil(S/16) planes of 16 BYTES per arm (whatever T): byte m % 16 of plane m / 16 = integer decay order of compact-image
//                                     slot m (< 128; unused slots carry 5): read by the MIXO kernel variants when the batch's orders differ
//   slots                 2S planes   slot m = planes 2m, 2m+1 = (p0 p1 p2 p3 | p4 p5 force type);
//                                     type -1 = continuation of the previous slot (p6..p11 / p12..p16),
//                                     type 0 = empty
//   tool (per-arm only)   3 planes    frame rows 0,1,2 (a shared tool lives in KConst)
//   mixw (per-arm only)   2 planes    mixer weights w0..w3 | w4 w5 - -  (shared weights live in KConst)
//   lastvec               [(n + 4) / 4][Bpad] quads of FLOAT: the unique basis vector of the last cycle (n values), then
//                                     sig as +-1 / +-2 (nullspace:91-92; vfik_kernel.hip); chains of up to 7 joints only
//   ext                   [4][B][n]   last commands of mixer channels 2..5
// Batch-shared constants: chain geometry, limits and parameters.  They live in DEVICE memory (one
// copy per handle, rewritten only by vfik_set_chain / vfik_set_params) and are read through the
// scalar cache; the kernarg block stays small.  Measured: with the 2.3 KB of constants inside the
// kernarg segment every batch of s_loads was a long-latency fetch in the middle of the kinematics.
template <int NJ>
struct KConst {
    // ---- kinematics block: copied into LDS by every wave (one or two 1-KiB requests) and read from
    // there as vector operands.  As scalar (SGPR) operands the ~90 doubles did not fit the 100 SGPRs:
    // the compiler hoisted every s_load to the top and spilled to VGPR lanes (v_writelane/readlane).
    // Chain in Denavit-Hartenberg form, derived on the host from the z-normal form of vfik_chain
    // (vfik_kernel.hip: kconst_fill_t): T_ee = base * prod_i [ Screw_z(q_i) Tx(a_i) Rx(alpha_i) ] * Screw_z(tail)
    double base[12];  // B[0], row-major 3x4
    // c = crev*cos(q+off) + cprs, s = crev*sin(q+off) + sprs, displacement = qd*q + d: revolute joints have
    // (crev, cprs, sprs, qd) = (1, 0, 0, 0), prismatic ones (0, cos off, sin off, 1) -- arithmetic blends
    struct DH { double off, crev, cprs, sprs, qd, d, a, ca, sa, pad; } dh[NJ];   // dh[0].pad: the batch's uniform repeller FORCE (below)
    double tail_c, tail_s, tail_e;  // trailing z-screw of the last fixed transform
    // safe distance of every decay repeller of the batch when they all share one (and one force, dh[0].pad): the uniform repeller
    // image of the straight-line path then carries (x y z radius) per slot only (vfik_abi.cpp: pack_fields, vfik_set_fields)
    double rep_safe;
    // ---- everything else: read through the scalar cache.  Ordered by use on the lean paths: the first members share the 1-KiB
    // rows that every wave copies to LDS for the kinematics block, so the L2 has them by the time the scalar loads ask -- with
    // one handle after another (inputs from HBM) the first use of lambda2 / rot_slow cost the wave an HBM round trip (round 3).
    double speed, lambda2, rot_slow, null_gain, lookahead, max_vel;
    double cos_slow;  // cos(rot_slow): rotation angles with a smaller cosine need no atan2 (scalar = 1)
    double jp_kp, jp_delta;  // joint P controller (joint_p_controller:55-57)
    double jl_gain;          // gain of the joint-limit task (jl_k is jl_gain / half^2 of the STATIC limits; per-cycle limits use this)
    double mix_w[VFIK_MIX_CHANNELS];
    // shared tool frame (rows 0..2 of the 4x4; per-arm tools are a device array).  In the rows every wave copies to LDS: with the inputs
    // cold a scalar load from the tail of the block is an HBM round trip of its own (measured on a flag there: C3 +7 %, profiles/r04_ab_tool.txt)
    double tool[12];
    double wy[6];    // the batch's IK weights (/weight, vf:295-309), in the copied rows for the same reason
    double wq[NJ];
    double q_lo[NJ];
    double q_hi[NJ];
    double q_mid[NJ];     // (lo + hi) / 2
    double inv_half[NJ];  // 2 / (hi - lo)
    double jl_k[NJ];      // jl_gain / half^2
    unsigned prismatic_mask;
    unsigned pad0;
    static constexpr int KIN_BYTES = (12 + 10 * NJ + 4 + 10 + VFIK_MIX_CHANNELS + 12 + 6 + NJ) * 8;  // through wq: the block every wave copies to LDS
    static constexpr int KIN_ROWS = (KIN_BYTES + 1023) / 1024;  // 1-KiB LDS rows / requests
};
// The device image of the constants is KConst<NJ> padded to a multiple of 1 KiB, then the 1-KiB sin / cos table
// ((sin, cos)(k pi/32), k = 0..63) that every wave copies to LDS with the kinematics block.
// where the host patches the uniform repeller pair into the device image (vfik_abi.cpp: write_uniform_pair)
#define VFIK_KCONST_REP_FORCE_OFF ((12 + 9) * 8)
#define VFIK_KCONST_REP_SAFE_OFF(nj) ((12 + 10 * (nj) + 3) * 8)
static_assert(offsetof(KConst<7>, rep_safe) == VFIK_KCONST_REP_SAFE_OFF(7) && offsetof(KConst<14>, rep_safe) == VFIK_KCONST_REP_SAFE_OFF(14), "KConst::rep_safe");
static_assert(offsetof(KConst<7>, dh) + offsetof(KConst<7>::DH, pad) == VFIK_KCONST_REP_FORCE_OFF, "KConst::dh[0].pad");
static_assert(offsetof(KConst<7>, wq) + 7 * 8 == KConst<7>::KIN_BYTES && offsetof(KConst<14>, wq) + 14 * 8 == KConst<14>::KIN_BYTES, "KConst::wq closes the LDS-copied block");
static_assert(KConst<7>::KIN_BYTES <= 1024 && KConst<6>::KIN_BYTES <= 1024 && KConst<10>::KIN_BYTES <= 2048 && KConst<14>::KIN_BYTES <= 2048, "the copied block keeps its row count");
template <int NJ> struct KTab { static constexpr int OFFSET = ((int)sizeof(KConst<NJ>) + 1023) / 1024 * 1024; };

// DH patterns the lean kernels are built for (cycle_body, DHP): per joint count, bit i of
//   SWAP: link i has a = 0 and alpha = +pi/2 EXACTLY as the kernel sees it (ca == 0, sa == 1: the host snaps |ca| < 1e-15),
//   NONE: a = 0 and alpha = 0 (ca == 1, sa == 0),     D0: the link's z-offset d is 0.
// Pattern 1 per joint count: the KUKA LWR 4+ (vfclik's default robot), two of them in series (BASELINE's C5), the 6-joint arm of
// robots.py.  A chain qualifies when its own masks contain the pattern's (kconst_fill reports them; vfik_abi.cpp decides).
//   OFF0 / OFFPI: the joint's angle offset in the DH form is an even / odd multiple of pi (the joint angle itself enters the
//   sin / cos, an odd multiple negates both),     BASE_I: the base frame B[0] is the identity (joint 1 starts from unit vectors).
template <int NJ, int DHP> struct DhPattern { static constexpr unsigned SWAP = 0, NONE = 0, D0 = 0, OFF0 = 0, OFFPI = 0; static constexpr bool BASE_I = false; };
template <> struct DhPattern<7, 1> { static constexpr unsigned SWAP = 0x3Fu, NONE = 0x40u, D0 = 0x2Au, OFF0 = 0x15u, OFFPI = 0x6Au; static constexpr bool BASE_I = true; };
template <> struct DhPattern<14, 1> {
    static constexpr unsigned SWAP = 0x1FBFu, NONE = 0x2040u, D0 = 0x152Au, OFF0 = 0xA95u, OFFPI = 0x356Au;
    static constexpr bool BASE_I = true;
};
template <> struct DhPattern<6, 1> { static constexpr unsigned SWAP = 0x1Du, NONE = 0x20u, D0 = 0x16u, OFF0 = 0x27u, OFFPI = 0x18u; static constexpr bool BASE_I = true; };
// the pattern id of a chain with these masks (0: none built for it)
int dh_pattern_of(int nj, unsigned swap, unsigned none, unsigned d0, unsigned off0, unsigned offpi, bool base_identity);

// Chains longer than this have no registers left for loop-carried state: their rollout is a sequence of
// single-cycle launches that integrate q on the way out (vfik_abi.cpp), not the ROLL kernel variant.
#define VFIK_ROLL_MAX_NJ 7

struct KArgs {
    int B;
    int Bpad;         // B rounded up to 64: pitch of the quad planes
    int slots_used;
    int fast_order;   // >= 0: every used slot of every arm is a decay repeller (or empty) with this integer
                      // decay order (what object_feeder sends for point obstacles); -1: general path
    unsigned flags;
    int tool_stride;  // 0: one tool for the batch (KConst::tool); else per-arm tool quads ([3][Bpad])
    int plain;        // 0: the general variants; else the PLAIN kernels (vfik_kernel.hip), 1 + (the batch's shared tool is not the identity ? 1 : 0) + (its IK weights are not all one ? 2 : 0)
    int block;        // threads per block of the launch (read from here: blockDim.x costs its own scalar load)
    const void* q;
    const void* goal;
    const void* slots;
    const void* slots_fast;  // compact repeller image of the straight-line path: 3 quad planes per pair of slots (vfik_abi.cpp, pack_fields)
    const void* tool;
    const void* null_control;
    const void* ext;
    const void* mixw;  // per-arm mixer weights, 2 quad planes, or NULL (KConst::mix_w for every arm)
    const double* wts;    // per-arm IK weights [6 + n][Bpad] (wy rows, then wq rows), or NULL (KConst::wy / wq)
    const void* q_ref;    // [B][n] joint P controller reference (mixer channel 2), or NULL
    const void* q_cmded;  // [B][n] LWR echo of the commanded position (bridge:199-203 command form), or NULL
    float* lastvec;
    void* qdot_vf;
    void* qdot_null;
    void* qdot_out;
    void* pose;
    void* pose_nt;
    void* v6;
    void* qdist;
    void* goal_dist;  // [B][2] distance / rotation angle (degrees) to the goal block, or NULL
    int* status;
    unsigned long long* stamps;  // diagnostic builds only (-DVFIK_STAMPS): [waves][8] s_memtime values
    const void* kc;              // KConst<n> in device memory
    // closed-loop rollout (ROLL kernel variant): n_cycles control cycles per launch, q += dt * qdot_out
    void* q_out;                 // [B][n] joint angles after the last cycle, or NULL
    double dt;
    int n_cycles;                // 0: ordinary single-cycle launch
    int clamp;                   // keep q inside [q_lo, q_hi] after each integration step
    int status_or;               // OR the status bits into what a.status already holds (cycle 2.. of a stepped rollout)
    // ABI 3
    const int* active;           // [B] fresh-q gate (vf:312-313, nullspace:162-163): 0 = the arm stores nothing this launch; NULL = all
    const void* q_lo;            // [B][n] this cycle's joint limits per arm (nullspace:167, joint_p_controller:80), or NULL:
    const void* q_hi;            //        the chain's static limits of KConst
    void* q_ref_out;             // [B][n] the joint controller's reference after its clamp (joint_p_controller:121), or NULL
    int sub8_max_batch;          // batches up to this size take the eight-lanes-per-arm kernel when the launch is lean (0: never)
    int sub8_max_batch_ns;       // ... with the nullspace module, qdot_out / status only
    int sub8_max_batch_full;     // ... when the launch asks for more than qdot_out (the rows the per-arm processes publish every cycle)
    int n_simd;                  // SIMDs of the device (4 per CU): launches of at most that many waves are one wave per SIMD
    const void* funnel;          // aux block, 6 quad planes: funnel (x y z ax | ay az cutAngle angleOrder | cutDist distOrder force present) and
                                 // hemisphere (x y z nx | ny nz safeDist order | force present - -), or unused
    int slots_used_fast;         // slots of the COMPACT repeller image in use (an arm's funnel is not a slot there)
    int has_funnel;              // some arm's field set has a funnel attractor or a hemisphere repeller (straight-line path: the FUN kernel variants)
    const void* arena;           // the handle's state arena [goal | kconst | lastvec | slots_fast | slots] (arena_layout), or NULL
    int uni;                     // 1: every decay repeller of the batch shares one safe distance and one force (KConst::rep_safe, dh[0].pad): lean launches read the uniform image
    int uni_planes;              // quad planes of the compact image = offset of the uniform image behind slots_fast
    int waves2;                  // 1: lean straight-line float launches of more than n_simd waves take the two-waves-per-SIMD build (VFIK_TWO_WAVES=0: never)
    int pers;                    // 1: lean straight-line launches of more than n_simd waves take the persistent kernel (VFIK_PERSISTENT=0: never)
    // round 4: decay repellers whose INTEGER orders differ (README.old:75 documents order 20 beside the feeder's 5, object_feeder:302)
    const void* orders;          // order planes: 16 bytes per arm and plane = the decay orders of 16 slots of the compact image, one byte each
    int mixed;                   // 1: the batch's repellers do not share one order -- the straight-line path reads `orders` (MIXO kernel variants)
    int dhp;                     // DH pattern of the chain the lean kernels may assume (DhPattern; 0: none) -- only with `plain`
};

// The per-handle device state a LEAN launch reads lives in ONE allocation with offsets that follow from (io type, joints, Bpad):
//   [goal: 4 quad planes | aux (funnel 3, hemisphere 3): 6 quad planes | kconst: KCONST_SLOT(nj) bytes | lastvec: (nj + 4) / 4 planes of 16 B | slots_fast ... | slots_uni ... | slots ...]
// so that such a launch's kernarg is one base pointer + the io pointers (KLean, 56 bytes) instead of the 340-byte KArgs: what a
// launch costs the HOST grows with its kernarg (tools/ubench_launch: 32-64 B 2.8 us, 336 B 3.9 us on a slow host), and at a
// 5-us launch period the enqueue loop is never far from being the bottleneck.
#define VFIK_KCONST_SLOT(kconst_bytes) ((((kconst_bytes) + 2048) + 255) / 256 * 256)
struct KLean {
    const void* base;            // the arena
    const void* q;
    void* qdot_out;
    int* status;
    int B, Bpad, slots_used, fast_order;   // (slots_used: of the compact image)
    unsigned flags;
    int block;
};

// size of KConst<nj> for the host (0 if nj is not built); kconst_fill returns the largest
// recomposition error of the DH conversion (the caller refuses a chain above 1e-9)
size_t kconst_bytes(int nj);
// fill a host image of KConst<nj> at dst
double kconst_fill(int nj, void* dst, const vfik_chain& chain, const vfik_params& p, const double* tool12, int* plain, int* dhp = nullptr);

// Type-erased launchers (implemented in vfik_kernel.hip).  kargs points to a KArgs<nj>.
uint32_t supported_joints_mask();
// *sub8 (may be NULL) is set to 1 when the launch took the eight-lanes-per-arm kernel
hipError_t launch_cycle(int io_dtype, int nj, const KArgs& kargs, int block, hipStream_t stream, int* sub8 = nullptr);
hipError_t launch_probe(int io_dtype, const void* pose, const void* goal, const void* slots, int B, long Bp, int slots_used,
                        double rot_slow, double cos_slow, void* out, hipStream_t stream);
hipError_t launch_monitor(int io_dtype, const void* pose, const void* frames, int O, long count, void* out, hipStream_t stream, const int* active = nullptr);
hipError_t launch_track(int io_dtype, const void* pose, const void* v6, double* state, void* out, const int* active, int B, hipStream_t stream);
hipError_t launch_mix(int io_dtype, const void* cmds, const double* w_dev, int K, long count, long chan_stride,
                      void* out, hipStream_t stream);

}  // namespace vfik
