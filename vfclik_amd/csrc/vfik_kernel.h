// vfik_kernel.h -- kernel argument block shared by vfik_kernel.hip (device) and vfik_abi.cpp (host).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vfik_types.h"

// joint counts the library is built for (one fully unrolled kernel each)
#ifndef VFIK_NJ_LIST
#define VFIK_NJ_LIST X(6) X(7) X(10) X(14)
#endif

namespace vfik {

// Device data layout (DESIGN.md "Data layout in HBM").  T = io dtype (float | double), Bp = batch.
//   q, qdot_*, qdist      [Bp][n]    batch-major (the reference's bottles: n doubles per arm)
//   pose, pose_nt         [Bp][16]
//   goal                  [18][Bp]   SoA: frame16 (element 15 = "goal present" flag), slow-down, force
//   slots                 [S][8][Bp] SoA: p0..p5, force, type   (type -1 = continuation of the
//                                    previous slot: p6..p11 / p12..p16; type 0 = empty)
//   tool                  [12] shared or [12][Bp]
//   lastvec               [n][Bp] double, sig [Bp] int   (nullspace:91-92 for the unique basis vector)
//   ext                   [4][Bp][n] last commands of mixer channels 2..5
template <int NJ>
struct KArgs {
    int B;
    int slots_used;
    unsigned flags;
    int tool_per_arm;
    const void* q;
    const void* goal;
    const void* slots;
    const void* tool;
    const void* null_control;
    const void* ext;
    double* lastvec;
    int* sig;
    void* qdot_vf;
    void* qdot_null;
    void* qdot_out;
    void* pose;
    void* pose_nt;
    void* v6;
    void* qdist;
    int* status;
    // chain (z-normal form) and limits
    double CB[NJ + 1][12];
    double q_lo[NJ];
    double q_hi[NJ];
    unsigned prismatic_mask;
    unsigned pad0;
    // parameters
    double speed, lambda2, rot_slow, null_gain, lookahead, jl_gain, max_vel;
    double wy[6];
    double wq[NJ];
    double mix_w[VFIK_MIX_CHANNELS];
};

// Type-erased launchers (implemented in vfik_kernel.hip).  kargs points to a KArgs<nj>.
uint32_t supported_joints_mask();
hipError_t launch_cycle(int io_dtype, int nj, const void* kargs, int B, int block, hipStream_t stream);
hipError_t launch_mix(int io_dtype, const void* cmds, const double* w_dev, int K, long count, long chan_stride,
                      void* out, hipStream_t stream);

}  // namespace vfik
