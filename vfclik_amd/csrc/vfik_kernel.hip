// vfik_kernel.hip -- the fused control-cycle kernel for gfx950 (MI355X), one LANE per arm.
//
// One launch = one control cycle of B arms = the loop bodies of the reference's per-arm processes:
//   scripts/vf:311-347,455-466        q -> FK -> tool -> field -> twist -> RefPoint -> getIKV -> qdot
//   scripts/nullspace:162-184         J -> nullspace vector (sign memory) -> control -> check_limits
//   scripts/debug_jointlimits:61-73   distToCenter
//   src/command_mixer.py:78-82        weighted sum of the command channels (+ bridge:188-195 limiter)
//
// Mapping: every arm is an independent ~2 kflop float64 problem whose largest
// matrix is 6 x n (n <= 16); a wavefront evaluates 64 arms, one per lane, entirely in registers,
// with every input of the cycle staged into the wave's LDS region by direct global -> LDS loads.  No
// cross-lane traffic is needed, no lane idles, and the per-arm field list is read as a structure of
// arrays so every wave-level load is one contiguous 1-KiB row.  No MFMA: there is no contraction to feed it.
// Small batches (a handful of arms with vfclik's default process set, up to 4 096 arms when the per-cycle rows are published or
// no module runs) take cycle_sub8_kernel instead: eight lanes per arm, adopted where the same-box A/B wins (launch_v).
// Kernels of this file: cycle_kernel (variants by template: io type, joints, nullspace module, PLAIN, rollout, field path, LEAN,
// compile-time flags, persistent), cycle_sub8_kernel, mix_kernel, track_kernel, monitor_kernel, probe_kernel.
//
// Arithmetic is float64 whatever the io dtype (DESIGN.md "Precision").
#include "vfik_kernel.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

namespace vfik {
#define VFIK_CAT2(a, b) a##b
#define VFIK_CAT(a, b) VFIK_CAT2(a, b)
// Chains of VFIK_HEAVY_MIN_NJ joints or more: the non-lean single-cycle variants are an object of their own (-DVFIK_HEAVY_PART; launch_v)
#ifndef VFIK_HEAVY_MIN_NJ
#define VFIK_HEAVY_MIN_NJ 12
#endif
#if defined(VFIK_ONLY_NJ) && VFIK_ONLY_NJ >= VFIK_HEAVY_MIN_NJ
void VFIK_CAT(launch_heavy_nj, VFIK_ONLY_NJ)(int io_dtype, bool ns, bool plain, bool fastf, const KArgs& a, dim3 grid, dim3 blk, size_t lds, hipStream_t stream);
#endif
namespace {

constexpr double EPS_LEN = 1e-12;  // lengths below this are zero (unit vector := 0)
constexpr double D_FLOOR = 1e-9;   // distance floor inside decay laws
constexpr double MAG_CAP = 1e6;    // cap of a repeller's magnitude

// ------------------------------------------------------------------------------------------------
// float64 building blocks.  On gfx950 one wave per SIMD issues a float64 FMA about every 5 cycles
// (8 when dependent), v_rcp/v_rsq_f64 take ~17-20, and the IEEE sequences the compiler emits for
// `a / b`, sqrt() and sincos() cost ~65, ~90-110 and ~370 cycles (tools/ubench_fp64.hip).  The
// control cycle needs none of their corner-case handling, so it uses Newton-refined reciprocals and
// a branch-free sincos that the scheduler can interleave across the independent joints.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_nr(double x) {  // 1/x, ~1 ulp, x normal and non-zero
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// 1/sqrt(x) from ONE Newton step on v_rsq_f64 (5 instructions, relative error ~1e-14): enough for the decay repellers,
// whose magnitude ((radius + safe) / D)^order feeds a vector that is normalised afterwards; sqrt_rsqrt's two
// Goldschmidt steps (10 instructions, ~1 ulp) stay where the value itself is published.  x = 0 gives a large finite value.
__device__ __forceinline__ double rsqrt_1nr(double x) {
    const double y = __builtin_amdgcn_rsq(fmax(x, 1e-300));
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

__device__ __forceinline__ double rsqrt_1nr_pos(double x) {   // ... for x > 0 known to the caller (no clamp)
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

// 1/x from ONE Newton step on v_rcp_f64 (relative error ~1e-14, like rsqrt_1nr): the LDL^T pivots of the IK's normal matrix, whose
// condition number is bounded by (sigma_max^2 + lambda^2) / lambda^2 ~ 1e3 -- 1e-11 in the joint velocities against a 1e-9 test bar.
__device__ __forceinline__ double rcp_1nr(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}

// sqrt(x) and 1/sqrt(x) together (Goldschmidt from v_rsq_f64).  x = 0 gives (0, large finite).
__device__ __forceinline__ void sqrt_rsqrt(double x, double& root, double& inv) {
    const double y = __builtin_amdgcn_rsq(fmax(x, 1e-300));
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    root = g;
    inv = h + h;
}

__device__ __forceinline__ double norm3(double x, double y, double z) {
    double r, i;
    sqrt_rsqrt(x * x + y * y + z * z, r, i);
    return r;
}

// acc + x*w with the product and the sum rounded separately, as CPython evaluates
// `result[i] += v[i] * w` (command_mixer.py:81).  HIP's __dmul_rn/__dadd_rn are plain operators that
// the compiler may still fuse, so contraction is switched off for this statement block.
__device__ __forceinline__ double mac_unfused(double acc, double x, double w) {
#pragma clang fp contract(off)
    const double prod = x * w;
    return acc + prod;
}

__device__ __forceinline__ double mul_unfused(double x, double w) {
#pragma clang fp contract(off)
    return x * w;
}

// sin and cos: Cody-Waite reduction by pi/2 in three parts, then the classic minimax kernels on
// [-pi/4, pi/4] (coefficients of fdlibm's __kernel_sin / __kernel_cos), < 1 ulp for |x| up to ~1e5 rad.
// Joint angles live inside their limits (a few radians).
__device__ __forceinline__ void sincos_fast(double x, double& s, double& c) {
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-k, 6.07710050630396597660e-11, r);
    r = __builtin_fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double sr = __builtin_fma(z * r, ps, r);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int n = (int)k;
    const double sv = (n & 1) ? cr : sr, cv = (n & 1) ? sr : cr;
    s = (n & 2) ? -sv : sv;
    c = ((n + 1) & 2) ? -cv : cv;
}

// The same for N independent arguments, written operation by operation across the N (structure-of-
// arrays order): a lone wave issues a dependent float64 op every ~8 cycles but independent ones every
// ~5, and the scheduler mostly keeps source order, so the source is laid out interleaved.
template <int N>
__device__ __forceinline__ void sincos_fast_n(const double* x, double* s, double* c) {
    double k[N], r[N], z[N], ps[N], pc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = __builtin_rint(x[i] * 6.36619772367581382433e-01);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-k[i], 1.57079632673412561417e+00, x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-k[i], 6.07710050630396597660e-11, r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-k[i], 2.02226624879595063154e-21, r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = r[i] * r[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = __builtin_fma(z[i], 1.58969099521155010221e-10, -2.50507602534068634195e-08);
        pc[i] = __builtin_fma(z[i], -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = __builtin_fma(z[i], ps[i], 2.75573137070700676789e-06);
        pc[i] = __builtin_fma(z[i], pc[i], -2.75573143513906633035e-07);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = __builtin_fma(z[i], ps[i], -1.98412698298579493134e-04);
        pc[i] = __builtin_fma(z[i], pc[i], 2.48015872894767294178e-05);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = __builtin_fma(z[i], ps[i], 8.33333333332248946124e-03);
        pc[i] = __builtin_fma(z[i], pc[i], -1.38888888888741095749e-03);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = __builtin_fma(z[i], ps[i], -1.66666666666666324348e-01);
        pc[i] = __builtin_fma(z[i], pc[i], 4.16666666666666019037e-02);
    }
    double zr[N], zz[N], hh[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { zr[i] = z[i] * r[i]; zz[i] = z[i] * z[i]; hh[i] = __builtin_fma(z[i], -0.5, 1.0); }
#pragma unroll
    for (int i = 0; i < N; ++i) { ps[i] = __builtin_fma(zr[i], ps[i], r[i]); pc[i] = __builtin_fma(zz[i], pc[i], hh[i]); }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int n = (int)k[i];
        const double sv = (n & 1) ? pc[i] : ps[i], cv = (n & 1) ? ps[i] : pc[i];
        s[i] = (n & 2) ? -sv : sv;
        c[i] = ((n + 1) & 2) ? -cv : cv;
    }
}

// sin and cos of N independent arguments through a 64-entry table: x = k pi/32 + r, |r| <= pi/64, and
//   sin x = S_k cos r + C_k sin r,   cos x = C_k cos r - S_k sin r
// with (S_k, C_k) = (sin, cos)(k pi/32) read from the wave's LDS copy of the table (one 16-byte read per angle, index
// k mod 64) and Taylor polynomials that are exact to the last bit on so short an interval (r^9/9! < 1e-16 r,
// r^10/10! < 3e-20).  23 instructions an angle against 33 for the pi/2 reduction above, whose minimax kernels need
// six coefficients each and a sign / swap selection by quadrant; ~2 ulp (the table entries are rounded).  pi/32 is
// split into a 33-bit head, so that k * head is exact for |x| < 1e5 rad, and a tail.
template <int N>
__device__ __forceinline__ void sincos_tab_n(const double* x, const char* tab, double* s, double* c) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    double k[N], r[N], z[N], ps[N], pc[N], S[N], C[N];
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = __builtin_rint(x[i] * 10.185916357881302);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const d2 t = *reinterpret_cast<const d2*>(tab + (((int)k[i]) & 63) * 16);
        S[i] = t.x; C[i] = t.y;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-k[i], 0.09817477042088285, x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-k[i], 3.79818781656637e-12, r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = r[i] * r[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = __builtin_fma(z[i], -1.98412698412698412698e-04, 8.33333333333333333333e-03);
        pc[i] = __builtin_fma(z[i], 2.48015873015873015873e-05, -1.38888888888888888889e-03);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = __builtin_fma(z[i], ps[i], -1.66666666666666666667e-01);
        pc[i] = __builtin_fma(z[i], pc[i], 4.16666666666666666667e-02);
    }
    double zr[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { zr[i] = z[i] * r[i]; pc[i] = __builtin_fma(z[i], pc[i], -0.5); }
#pragma unroll
    for (int i = 0; i < N; ++i) { ps[i] = __builtin_fma(zr[i], ps[i], r[i]); pc[i] = __builtin_fma(z[i], pc[i], 1.0); }  // sin r, cos r
#pragma unroll
    for (int i = 0; i < N; ++i) { s[i] = S[i] * pc[i]; c[i] = C[i] * pc[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { s[i] = __builtin_fma(C[i], ps[i], s[i]); c[i] = __builtin_fma(-S[i], ps[i], c[i]); }
}

// N independent 1/sqrt(x), operation by operation (x = 0 gives a large finite value)
template <int N>
__device__ __forceinline__ void rsqrt_n(const double* x, double* inv) {
    double y[N], g[N], h[N], r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = __builtin_amdgcn_rsq(fmax(x[i], 1e-300));
#pragma unroll
    for (int i = 0; i < N; ++i) { g[i] = x[i] * y[i]; h[i] = 0.5 * y[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-g[i], h[i], 0.5);
#pragma unroll
    for (int i = 0; i < N; ++i) { g[i] = __builtin_fma(g[i], r[i], g[i]); h[i] = __builtin_fma(h[i], r[i], h[i]); }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(-g[i], h[i], 0.5);
#pragma unroll
    for (int i = 0; i < N; ++i) inv[i] = 2.0 * __builtin_fma(h[i], r[i], h[i]);
}

// x^n, n a wave-uniform small non-negative integer: square-and-multiply with scalar control flow
__device__ __forceinline__ double powi_uniform(double x, int n) {
    double r = 1.0, b = x;
    while (n) {
        if (n & 1) r *= b;
        n >>= 1;
        if (n) b *= b;
    }
    return r;
}

// x^order for x > 0.  Decay orders are small integers in every message the reference sends
// (object_feeder:277,279,302; README.old:75 uses 20): multiply chain when we can, pow() otherwise.
__device__ __forceinline__ double pow_order(double x, double order) {
    const int n = (int)order;
    const bool isint = (double)n == order && n >= 0 && n < 128;
    if (__all(isint)) {
        const int n0 = __builtin_amdgcn_readfirstlane(n);
        if (__all(n == n0)) return powi_uniform(x, n0);
        double r = 1.0, b = x;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            r = (n & (1 << k)) ? r * b : r;
            b = b * b;
        }
        return r;
    }
    return isint ? powi_uniform(x, n) : pow(x, order);
}

// atan2(y, x) for y >= 0 (an angle in [0, pi]), branch-free: two reductions bring the argument of the
// arctangent below tan(pi/8), where fdlibm's 11-coefficient kernel (s_atan.c, |x| < 7/16) is good to < 1 ulp.
// ocml's atan2 costs a lone wave ~2 000 cycles (tools/stamps.py: the waves that needed it ended the kernel
// 0.4 us after the rest); this is ~45 instructions.
__device__ __forceinline__ double atan2_pos(double y, double x) {
    const double ax = fabs(x);
    const bool steep = y > ax;                         // angle from the nearer axis: t in [0, 1]
    const double num = steep ? ax : y, den = steep ? y : ax;
    double t = den > 0.0 ? num * rcp_nr(den) : 0.0;    // atan2(0, 0) = 0
    const bool upper = t > 0.41421356237309503;        // tan(pi/8): atan(t) = pi/4 + atan((t - 1) / (t + 1))
    const double tr = (t - 1.0) * rcp_nr(t + 1.0);
    t = upper ? tr : t;
    const double z = t * t, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                      w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                      w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    double a = t - t * (s1 + s2);
    a = upper ? 0.78539816339744830962 + a : a;        // angle from the nearer axis, in [0, pi/4]
    a = steep ? 1.57079632679489661923 - a : a;        // angle from the x axis of (|x|, y)
    return x < 0.0 ? 3.14159265358979323846 - a : a;
}

// Unit rotation axis (base frame) and angle of the rotation taking R to G, i.e. of log(G R^T) =
// KDL diff(R, G).rot.  Branch-free main line; the half-turn neighbourhood (sin(theta) < 1e-4, cos < 0),
// where the antisymmetric part vanishes, is fixed up from the symmetric part under a wave-uniform
// test.  The angle itself is only needed below the slow-down angle: `need_theta` (wave-uniform) says
// whether any lane can be that close; otherwise theta is reported as pi (any value >= rot_slow does).
__device__ __forceinline__ void rot_axis_angle(const double* R, const double* G, double cos_slow, double* axis, double& theta,
                                               bool& has_axis) {
    double E[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E[3 * i + j] = G[3 * i] * R[3 * j] + G[3 * i + 1] * R[3 * j + 1] + G[3 * i + 2] * R[3 * j + 2];
    const double a0 = 0.5 * (E[7] - E[5]), a1 = 0.5 * (E[2] - E[6]), a2 = 0.5 * (E[3] - E[1]);
    const double c = 0.5 * (E[0] + E[4] + E[8] - 1.0);
    double s, sinv;
    sqrt_rsqrt(a0 * a0 + a1 * a1 + a2 * a2, s, sinv);
    axis[0] = a0 * sinv; axis[1] = a1 * sinv; axis[2] = a2 * sinv;
    has_axis = s >= EPS_LEN;
    theta = 3.14159265358979323846;
    if (__any(c > cos_slow)) theta = atan2_pos(s, c);  // some lane is within the slow-down angle
    const bool half_turn = s < 1e-4 && c < 0.0;
    if (__any(half_turn)) {
        // theta near pi (4 arms in 65 536 random goals): the axis comes from the symmetric part,
        // (E + E^T)/2 - c I = (1 - c) a a^T, whose row with the largest diagonal element is parallel to a;
        // the sign from the (small) antisymmetric part.  Selects, one rsqrt: the waves that get here
        // used to end the kernel 0.3 us after the others through three divergent branches with divisions.
        const bool k0 = E[0] >= E[4] && E[0] >= E[8], k1 = !k0 && E[4] >= E[8];
        const double s01 = 0.5 * (E[1] + E[3]), s02 = 0.5 * (E[2] + E[6]), s12 = 0.5 * (E[5] + E[7]);
        double x = k0 ? E[0] - c : (k1 ? s01 : s02);
        double y = k0 ? s01 : (k1 ? E[4] - c : s12);
        double z = k0 ? s02 : (k1 ? s12 : E[8] - c);
        const double flip = (x * a0 + y * a1 + z * a2 < 0.0) ? -1.0 : 1.0;
        double nn, k;
        sqrt_rsqrt(x * x + y * y + z * z, nn, k);
        k *= flip;
        if (half_turn) {
            axis[0] = x * k; axis[1] = y * k; axis[2] = z * k;
            has_axis = true;
            theta = atan2_pos(s, c);
        }
    }
}

// type 1, point attractor: G = goal rotation (9) + position (3); adds force*vector to tot, scales sc.
// `on` masks the whole contribution (goal block absent): selects instead of a branch.
__device__ __forceinline__ void attractor(const double* R, const double* p, const double* GR, const double* Gp,
                                          double slow, double force, double rot_slow, double cos_slow, bool on,
                                          double* tot, double* sc, double* dist = nullptr) {
    const double dx = Gp[0] - p[0], dy = Gp[1] - p[1], dz = Gp[2] - p[2];
    double D, Dinv;
    sqrt_rsqrt(dx * dx + dy * dy + dz * dz, D, Dinv);
    const double kt = (on && D > EPS_LEN) ? force * Dinv : 0.0;
    tot[0] += dx * kt; tot[1] += dy * kt; tot[2] += dz * kt;
    double ax[3], th;
    bool has_axis;
    rot_axis_angle(R, GR, cos_slow, ax, th, has_axis);
    const bool rot_on = on && has_axis && th > EPS_LEN;  // selects, not a multiply by 0: an absent goal's axis may be NaN
    tot[3] += rot_on ? ax[0] * force : 0.0; tot[4] += rot_on ? ax[1] * force : 0.0; tot[5] += rot_on ? ax[2] * force : 0.0;
    const double s0 = slow > 0.0 ? fmin(1.0, D * rcp_nr(slow)) : 1.0;
    const double s1 = rot_slow > 0.0 ? fmin(1.0, th * rcp_nr(rot_slow)) : 1.0;
    sc[0] *= on ? s0 : 1.0;
    sc[1] *= on ? s1 : 1.0;
    if (dist) { dist[0] = D; dist[1] = th; }
}

// Element e (0..7) of slot m of this lane's arm in the quad-plane layout (vfik_kernel.h): plane
// 2m + e/4, component e%4.  sq points at plane 0, component 0 of the lane's arm; Q = plane pitch in
// elements (4 * Bpad).
template <typename T>
__device__ __forceinline__ double slot_elem(const T* sq, long Q, int m, int e) {
    return (double)sq[(long)(2 * m + (e >> 2)) * Q + (e & 3)];
}

// One field slot of any type, read from memory (the general path: mixed primitive types in a wave,
// fractional decay orders, more slots than the prefetch window).
// `rd(m, e)` = element e of slot m: SlotGlobal reads the quad planes in memory, SlotLds the rows staged in LDS.
template <typename RD>
__device__ void eval_slot(const RD& rd, int m, const double* Rt, const double* pt, double rot_slow, double cos_slow,
                          double* tot, double* sc) {
    const int type = (int)rd(m, 7);
    if (type <= 0) return;
    const double p0 = rd(m, 0), p1 = rd(m, 1), p2 = rd(m, 2),
                 p3 = rd(m, 3), p4 = rd(m, 4), p5 = rd(m, 5),
                 force = rd(m, 6);
    if (type == VFIK_FIELD_REPELLER) {  // x y z radius safeDist order
        const double dx = p0 - pt[0], dy = p1 - pt[1], dz = p2 - pt[2];
        double D, Dinv;
        sqrt_rsqrt(dx * dx + dy * dy + dz * dz, D, Dinv);
        Dinv = D < D_FLOOR ? 1.0 / D_FLOOR : Dinv;
        const double mag = fmin(pow_order((p3 + p4) * Dinv, p5), MAG_CAP);
        const double k = force * mag * Dinv;
        tot[0] += dx * k; tot[1] += dy * k; tot[2] += dz * k;
    } else if (type == VFIK_FIELD_HEMISPHERE) {  // x y z nx ny nz | safeDist order
        const double safe = rd(m + 1, 0), order = rd(m + 1, 1);
        double nn, ninv;
        sqrt_rsqrt(p3 * p3 + p4 * p4 + p5 * p5, nn, ninv);
        if (nn > EPS_LEN) {
            const double h = ((pt[0] - p0) * p3 + (pt[1] - p1) * p4 + (pt[2] - p2) * p5) * ninv;
            const double mag = fmin(pow_order(safe * rcp_nr(fmax(h, D_FLOOR)), order), MAG_CAP);
            const double k = -force * mag * ninv;
            tot[0] += p3 * k; tot[1] += p4 * k; tot[2] += p5 * k;
        }
    } else if (type == VFIK_FIELD_FUNNEL) {  // x y z ax ay az | cutAngle angleOrder cutDist distOrder
        const double cutA = rd(m + 1, 0), ordA = rd(m + 1, 1),
                     cutD = rd(m + 1, 2), ordD = rd(m + 1, 3);
        double an, ainv;
        sqrt_rsqrt(p3 * p3 + p4 * p4 + p5 * p5, an, ainv);
        if (an > EPS_LEN) {
            const double ax = p3 * ainv, ay = p4 * ainv, az = p5 * ainv;
            const double wx = pt[0] - p0, wy = pt[1] - p1, wz = pt[2] - p2;
            const double along = wx * ax + wy * ay + wz * az;
            const double ex = wx - along * ax, ey = wy - along * ay, ez = wz - along * az;
            double P, Pinv, dist, dinv;
            sqrt_rsqrt(ex * ex + ey * ey + ez * ez, P, Pinv);
            sqrt_rsqrt(wx * wx + wy * wy + wz * wz, dist, dinv);
            Pinv = P < D_FLOOR ? 1.0 / D_FLOOR : Pinv;
            dinv = dist < D_FLOOR ? 1.0 / D_FLOOR : dinv;
            const double phi = atan2_pos(P, along);
            const double ga = cutA > 0.0 ? fmin(1.0, pow_order(phi * rcp_nr(cutA), ordA)) : 1.0;
            const double gd = fmin(1.0, pow_order(cutD * dinv, ordD));
            const double k = -force * ga * gd * Pinv;
            tot[0] += ex * k; tot[1] += ey * k; tot[2] += ez * k;
        }
    } else if (type == VFIK_FIELD_ATTRACTOR) {  // a second attractor: frame16 + slow over 3 slots
        double GR[9], Gp[3];
        GR[0] = p0; GR[1] = p1; GR[2] = p2; Gp[0] = p3; GR[3] = p4; GR[4] = p5;
        GR[5] = rd(m + 1, 0); Gp[1] = rd(m + 1, 1);
        GR[6] = rd(m + 1, 2); GR[7] = rd(m + 1, 3); GR[8] = rd(m + 1, 4);
        Gp[2] = rd(m + 1, 5);
        attractor(Rt, pt, GR, Gp, rd(m + 2, 4), force, rot_slow, cos_slow, true, tot, sc);
    }
}


// ------------------------------------------------------------------------------------------------
// Nullspace module, chains of up to 7 joints (scripts/nullspace:75-117), shared by cycle_kernel and cycle_sub8_kernel.
// In: Jm = the Jacobian by columns (destroyed: its rows are orthonormalised in place), the arm's sign memory
// (lv_r = lastvec, sig_r = sig, has_vec: lastvec holds a vector), c0 = /control[0], jl_task + descent(z) = the
// joint-limit task's direction.  Out: qn = move_in_nullspace (+ the projected joint-limit task), before check_limits and
// the gain; status bits; returns true when the sign memory was advanced (a unique nullspace direction exists).
// Every decision is taken PER LANE (per arm): which path an arm takes never depends on the other arms of its wave, so
// its result does not depend on where in a batch it sits (wave-wide votes only decide whether a path is executed at all).
// ------------------------------------------------------------------------------------------------
// ZFIRST / ZLAST: the Jacobian's first column is (x, y, 0, 0, 0, 1) / its last column (0, 0, 0, z) -- structural zeros of the chain's DH
// pattern (cycle_body: jzero).  The Gram-Schmidt pass tracks which entries are still exactly zero (a compile-time table once the loops are
// unrolled) and skips their terms: the last column's linear part stays zero to the end, the first column's zeros fill in at the first
// two steps -- 44 of the module's ~550 operations.
template <int NJ, bool ZFIRST = false, bool ZLAST = false, typename ZF>
__device__ __forceinline__ bool nullspace_core(double (&Jm)[NJ][6], double* lv_r, int& sig_r, bool& has_vec, double c0, bool jl_task,
                                               ZF&& descent, double* qn, int& status) {
    // Orthonormal basis (rows) of the row space of J by modified Gram-Schmidt, in place in Jm:
    // afterwards I - Q^T Q is restrict(I6, J) = I - pinv(J) J (nullspace:75-79).
    // One pass: loss of orthogonality ~ eps * cond(J), far below the 1e-6 bar wherever J is usable.
    // Right-looking order: once row s is normalised, the projections of ALL later rows onto it are
    // independent chains (5, 4, ... of them) that a lone wave can interleave; row by row (left-looking) the
    // same operations are one serial chain of 7-term dot products (dependent float64 ops issue every ~8
    // cycles, independent ones every ~5).  Same arithmetic, same order per row: same results.
    bool Z[NJ][6];   // entry (column i, row r) of J is still structurally zero
#pragma unroll
    for (int i = 0; i < NJ; ++i)
#pragma unroll
        for (int r = 0; r < 6; ++r) Z[i][r] = (ZFIRST && i == 0 && (r == 2 || r == 3 || r == 4)) || (ZLAST && i == NJ - 1 && r < 3);
    int rank = 0;
    double n0[6];  // squared norms of the rows of J: the rank test's yardstick
#pragma unroll
    for (int r = 0; r < 6; ++r) n0[r] = 0.0;
#pragma unroll
    for (int i = 0; i < NJ; ++i)
#pragma unroll
        for (int r = 0; r < 6; ++r)
            if (!Z[i][r]) n0[r] = __builtin_fma(Jm[i][r], Jm[i][r], n0[r]);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        double n1 = 0.0;
        if (s == 0) {
            n1 = n0[0];  // nothing has been projected out of the first row
        } else {
#pragma unroll
            for (int i = 0; i < NJ; ++i)
                if (!Z[i][s]) n1 += Jm[i][s] * Jm[i][s];
        }
        const bool keep = n1 > 1e-24 * n0[s] && n0[s] > 0.0;
        double n1r, n1i;
        sqrt_rsqrt(n1, n1r, n1i);
        const double inv = keep ? n1i : 0.0;
        rank += keep ? 1 : 0;
#pragma unroll
        for (int i = 0; i < NJ; ++i)
            if (!Z[i][s]) Jm[i][s] *= inv;
        double c[6];
#pragma unroll
        for (int r = s + 1; r < 6; ++r) c[r] = 0.0;
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int r = s + 1; r < 6; ++r)
                if (!Z[i][s] && !Z[i][r]) c[r] += Jm[i][s] * Jm[i][r];
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int r = s + 1; r < 6; ++r) {
                if (Z[i][s]) continue;                 // nothing of row s in this column
                if (Z[i][r]) { Jm[i][r] = -(c[r] * Jm[i][s]); Z[i][r] = false; }
                else Jm[i][r] -= c[r] * Jm[i][s];
            }
    }
    const int nullity = NJ - rank;
    bool advanced = false;
    if (nullity == 1) {
        double u[NJ];
        // u <- (I - Q^T Q) u: all six coefficients first (independent dot products), then the update (classical
        // Gram-Schmidt against an orthonormal Q); returns the largest |coefficient| = how much of u was row space
        auto project = [&](double* x) {
            double c[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) c[r] = 0.0;
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (!Z[i][r]) c[r] += Jm[i][r] * x[i];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    if (!Z[i][r]) x[i] -= c[r] * Jm[i][r];
            double cm = fabs(c[0]);
#pragma unroll
            for (int r = 1; r < 6; ++r) cm = fmax(cm, fabs(c[r]));
            return cm;
        };
        auto norm2 = [&](const double* x) {
            double n = 0.0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) n += x[i] * x[i];
            return n;
        };
        double nn = 0.0;
        // Warm start: the nullspace direction turns little between two control cycles, so last cycle's vector
        // (the sign memory, nullspace:92,104) is projected instead of a unit vector: no search for the best
        // unit vector, and ONE projection suffices when it removes little (its residual row-space part is
        // ~ eps cond(J) times what was removed).  An arm without a usable previous vector (first cycle, a
        // jump that leaves less than half of it) takes the cold path below.
        bool warm = false;
        if (__any(has_vec)) {  // (a stored vector is a unit vector)
#pragma unroll
            for (int i = 0; i < NJ; ++i) u[i] = lv_r[i];
            const double cm = project(u);
            nn = norm2(u);
            warm = has_vec && nn > 0.25;
            const bool again = warm && cm > 1e-2;  // a real move: project once more, as the cold path does
            if (__any(again)) {
                double u2[NJ];
#pragma unroll
                for (int i = 0; i < NJ; ++i) u2[i] = u[i];
                project(u2);
                const double nn2 = norm2(u2);
#pragma unroll
                for (int i = 0; i < NJ; ++i) u[i] = again ? u2[i] : u[i];
                nn = again ? nn2 : nn;
            }
        }
        if (__any(!warm)) {
            // cold: the normalised column of the projector with the largest diagonal, projected twice
            double dg[NJ], uc[NJ];
            double best = -1.0;
            int ib = 0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) dg[i] = 1.0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < NJ; ++i) dg[i] -= Jm[i][r] * Jm[i][r];
#pragma unroll
            for (int i = 0; i < NJ; ++i)
                if (dg[i] > best) { best = dg[i]; ib = i; }
#pragma unroll
            for (int i = 0; i < NJ; ++i) uc[i] = (i == ib) ? 1.0 : 0.0;
            project(uc);  // twice: the second pass squares the residual
            project(uc);
            const double nnc = norm2(uc);
#pragma unroll
            for (int i = 0; i < NJ; ++i) u[i] = warm ? u[i] : uc[i];
            nn = warm ? nn : nnc;
        }
        double nrm, ninv;
        sqrt_rsqrt(nn, nrm, ninv);
        // All three sign decisions are taken on the un-normalised u and applied with the normalisation, in one
        // multiplication per joint.  (1) The raw vector v as LAPACK's SVD leaves it: the first component that is
        // not negligible (|v_i| > 1e-9 of the unit vector) is negative (oracle + golden).
        const double thr = 1e-9 * nrm;
        bool found = false, negate = false;  // negate: v = -u / |u|
#pragma unroll
        for (int i = 0; i < NJ; ++i)
            if (!found && fabs(u[i]) > thr) { found = true; negate = u[i] > 0.0; }
        // (2) sign continuity against the previous cycle (nullspace:101-105): sig flips when sig v is farther from
        // lastvec than -sig v; |sig v - l|^2 - |sig v + l|^2 = -4 sig (v . l), so only the sign of v . l counts
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) dot += u[i] * lv_r[i];
        int sig = sig_r;
        const double vl = negate ? -dot : dot;
        if ((sig < 0 ? -vl : vl) < 0.0) sig = -sig;
        sig_r = sig;
        has_vec = true;
        const double k = (negate != (sig < 0)) ? -ninv : ninv;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            u[i] *= k;
            lv_r[i] = u[i];
            qn[i] = u[i] * c0;  // move_in_nullspace (nullspace:113-117): min(n, 4, 1) = 1 row
        }
        advanced = true;
    } else if (nullity >= 2) {
        status |= VFIK_ST_NULL_AMBIGUOUS;  // SVD basis not unique: /control cannot be honoured
    }
    if (jl_task) {
        double z[NJ];
        descent(z);
        double c[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) c[r] = 0.0;
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int r = 0; r < 6; ++r) c[r] += Jm[i][r] * z[i];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int i = 0; i < NJ; ++i) z[i] -= c[r] * Jm[i][r];
#pragma unroll
        for (int i = 0; i < NJ; ++i) qn[i] += z[i];
    }
    return advanced;
}

// ------------------------------------------------------------------------------------------------
// Staging through LDS (direct global -> LDS loads, no VGPR destination).  Every lane's inputs of
// one cycle are requested at wave start and land in the wave's private LDS region while the
// kinematics run; the arithmetic waits with counted s_waitcnt vmcnt at the three points where a
// group of inputs is first needed.  Rows are lane-linear (lane * 16 B or lane * 4 B): conflict-free.
//   region layout per wave, Q16 = 16-byte sub-planes per quad of T (1 for float, 2 for double):
//     quad rows  [tool 3 | goal 4 | slots 2*PRE | mixer weights 2] x Q16 x 1 KiB, then q (16-byte and 4-byte pieces),
//     then the kinematics block of KConst (1-2 KiB)
// ------------------------------------------------------------------------------------------------
template <typename T> struct Stage {
    // Slots staged at a time (prefetch window / chunk size).  float64 I/O stages 4: with 8 the region is 56 KB
    // and only three of a CU's four SIMDs get a wave (160 KB LDS) -- the launch then runs in two rounds
    // (measured 15.3 us instead of ~8 for the C3 batch).
    static constexpr int PRE = sizeof(T) == 8 ? 4 : 8;
    static constexpr int Q16 = (int)sizeof(T) / 4;           // 16-B pieces per quad
    static constexpr int QBYTES = 4 * (int)sizeof(T);        // bytes of one quad
    static constexpr int QSTEP = Q16 * 1024;                 // LDS bytes of one staged quad (a 1-KiB row per 16 bytes per lane)
    static constexpr int QPCF = 3 * PRE / 2;                 // quads of one chunk of the compact repeller image
    // Region of one wave, in the order [goal 4 | slot quads 0 .. QPCF-1 | q | kinematics | table] -- that much is all a LEAN
    // launch on the straight-line path touches (`lean_bytes`: 19.75 KB for 7 joints with float I/O, eight waves per CU) --
    // then [slot quads QPCF .. 2 PRE - 1 | tool 3 | mixer weights 2] for the general path and the optional per-arm inputs.
    static constexpr int GOAL_OFF = 0, SLOT_OFF = 4 * QSTEP, Q_OFF = (4 + QPCF) * QSTEP;
    // q is batch-major ([B][n]): a lane's n values are contiguous and travel as 16-byte pieces plus a
    // remainder of one to three 4-byte pieces (a 12-byte LDS-DMA did not land lane-linear on gfx950)
    __host__ __device__ static constexpr int qbytes(int nj) { return nj * (int)sizeof(T); }
    __host__ __device__ static constexpr int q16(int nj) { return qbytes(nj) / 16; }
    __host__ __device__ static constexpr int qrem(int nj) { return qbytes(nj) % 16; }
    __host__ __device__ static constexpr int qregion(int nj) { return q16(nj) * 1024 + qrem(nj) * 64; }
    __host__ __device__ static constexpr int kin_rows(int nj) { return ((12 + 10 * nj + 4 + 10 + VFIK_MIX_CHANNELS + 12 + 6 + nj) * 8 + 1023) / 1024; }  // = KConst<nj>::KIN_ROWS
    __host__ __device__ static constexpr int kin_off(int nj) { return Q_OFF + qregion(nj); }
    __host__ __device__ static constexpr int tab_off(int nj) { return kin_off(nj) + kin_rows(nj) * 1024; }  // sin / cos table, 1 KiB
    __host__ __device__ static constexpr int lean_bytes(int nj) { return tab_off(nj) + 1024; }
    __host__ __device__ static constexpr int slot_off(int idx, int nj) {  // slot quad idx of the staged chunk
        return idx < QPCF ? SLOT_OFF + idx * QSTEP : lean_bytes(nj) + (idx - QPCF) * QSTEP;
    }
    __host__ __device__ static constexpr int tool_off(int nj) { return lean_bytes(nj) + (2 * PRE - QPCF) * QSTEP; }
    __host__ __device__ static constexpr int mixw_off(int nj) { return tool_off(nj) + 3 * QSTEP; }
    __host__ __device__ static constexpr int bytes(int nj) { return mixw_off(nj) + 2 * QSTEP; }
};

// LDS byte address (relative to the q area) of byte b of this lane's q vector
template <typename T, int NJ>
__device__ __forceinline__ int q_lds_off(int b, int lane) {
    constexpr int n16 = Stage<T>::q16(NJ), rem = Stage<T>::qrem(NJ);
    if (b < n16 * 16) return (b >> 4) * 1024 + lane * 16 + (b & 15);
    const int rb = b - n16 * 16;
    (void)rem;
    return n16 * 1024 + (rb >> 2) * 256 + lane * 4 + (rb & 3);  // remainder: 4-byte pieces
}

typedef __attribute__((address_space(1))) const void* GPtr;
typedef __attribute__((address_space(3))) void* LPtr;

// one quad plane: this lane's quad (QBYTES at gsrc) -> the Q16 1-KiB rows at byte offset `off` of the region.
// NT: non-temporal cache policy (aux = 2) for bytes that one wave reads once per launch.  Round 2 adopted it for the long
// chains only (C5 -4.2 %; C3 and C3N +-0.6 %), from A/Bs in which back-to-back launches re-read one input set out of the
// Infinity Cache.  With the inputs coming from HBM (rotating input sets, bench.py --state cold) it is worth -6.8 % on C3
// (6.62 -> 6.16 us) and -5.8 % on C5, at +0.8 % / -4.3 % in the cache-resident state: every chain uses it since round 3
// (profiles/r03_nt_ab.txt).
#ifndef VFIK_SADDR_REQUESTS
#define VFIK_SADDR_REQUESTS 0     // 1: requests in the SGPR-base form (A/B builds: C5 +2.6 %, C3 +-0.5 % -- profiles/r03_ab_experiments.md 21)
#endif
// The address is (wave-uniform plane base) + (this lane's 32-bit byte offset).  Round 3 tried the SGPR-base form of the instruction
// (`global_load_lds v_off, s[base]`: the offset unsigned and opaque, so that the backend reads base + zext(offset)): 30 of the 34 requests
// take it, one 64-bit vector addition a request becomes two to three scalar instructions -- and a lone wave pays an issue slot for
// either: C3 +-0.5 %, C5 +2.6 % warm / +1.8 % cold.  Off (VFIK_SADDR_REQUESTS); voff < 4 GiB either way.
template <typename T, bool NT = false>
__device__ __forceinline__ void stage_quad(const char* gbase, unsigned voff, char* region, int off) {
#if VFIK_SADDR_REQUESTS
    asm("" : "+v"(voff));   // (opaque: or the optimiser widens the offset's multiplication and the addition no longer reads base + zext(offset))
    __builtin_amdgcn_global_load_lds((GPtr)(gbase + voff), (LPtr)(region + off), 16, 0, NT ? 2 : 0);
    if (Stage<T>::Q16 == 2) __builtin_amdgcn_global_load_lds((GPtr)(gbase + 16 + voff), (LPtr)(region + off + 1024), 16, 0, NT ? 2 : 0);
#else
    const char* gsrc = gbase + (long)(int)voff;
    __builtin_amdgcn_global_load_lds((GPtr)gsrc, (LPtr)(region + off), 16, 0, NT ? 2 : 0);
    if (Stage<T>::Q16 == 2) __builtin_amdgcn_global_load_lds((GPtr)(gsrc + 16), (LPtr)(region + off + 1024), 16, 0, NT ? 2 : 0);
#endif
}

template <typename T>
__device__ __forceinline__ void read_quad(const char* region, int off, int lane, double* out) {
    if (Stage<T>::Q16 == 1) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = *reinterpret_cast<const f4*>(region + off + lane * 16);
        out[0] = (double)v.x; out[1] = (double)v.y; out[2] = (double)v.z; out[3] = (double)v.w;
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 lo = *reinterpret_cast<const d2*>(region + off + lane * 16);
        const d2 hi = *reinterpret_cast<const d2*>(region + off + 1024 + lane * 16);
        out[0] = lo.x; out[1] = lo.y; out[2] = hi.x; out[3] = hi.y;
    }
}

// slot readers for eval_slot
template <typename T> struct SlotGlobal {
    const T* sq; long Q;  // plane 0, component 0 of this lane's arm; plane pitch in elements
    __device__ __forceinline__ double operator()(int m, int e) const { return slot_elem(sq, Q, m, e); }
};
template <typename T> struct SlotLds {
    const char* region; int lane; int nj;  // the wave's staging region: slot m sits in slot quads 2m, 2m + 1 (m < PRE)
    __device__ __forceinline__ double operator()(int m, int e) const {
        const int quad = 2 * m + (e >> 2), c = e & 3;
        const int off = Stage<T>::slot_off(quad, nj);
        if (Stage<T>::Q16 == 1)
            return (double)*reinterpret_cast<const float*>(region + off + lane * 16 + c * 4);
        return *reinterpret_cast<const double*>(region + off + (c >> 1) * 1024 + lane * 16 + (c & 1) * 8);
    }
};

// Chains of at least this many joints request their per-arm input planes with the non-temporal cache policy (stage_quad).
// A build knob for the A/B of that policy in the warm (cache-resident) and the cold (HBM-sourced) state: tools/ab_compare.py --state.
#ifndef VFIK_NT_MIN_NJ
#define VFIK_NT_MIN_NJ 0
#endif


// A scalar of the lean paths (lambda2, rot_slow, mix_w ...): they sit behind the kinematics block in KConst and travel to LDS with it.
// Long chains read them from that LDS copy (C5 -1.5 % in both cache states: no scalar-load round trip in front of the IK and the
// attractor); chains of up to 7 joints keep the scalar loads (from the LDS copy: C3 +0.2-0.3 %, C3N +0.6-1.2 %).
#define HOTK(member) (NJ >= 10 ? klc->member : kc->member)

#define VFIK_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// In-kernel section stamps for the diagnostic build only (make stamps; never in libvfik_hip.so).
#ifdef VFIK_STAMPS
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if ((threadIdx.x & 63) == 0) {                                                            \
            a.stamps[(long)(arm >> 6) * 10 + (i)] = t_;                                            \
            if ((i) == 0 || (i) == 7) a.stamps[(long)(arm >> 6) * 10 + 8 + ((i) != 0)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                          \
    } while (0)
#define PIN(x) asm volatile("" : "+v"(x))
#define PIN_ARR(arr, n)                              \
    do {                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) PIN((arr)[i_]); \
    } while (0)
#else
#define STAMP(i)
#define PIN(x)
#define PIN_ARR(arr, n)
#endif

// PLAIN = the common configuration, decided on the host: all joints revolute, last fixed transform a
// pure z-screw already absorbed, one identity tool for the batch, unit IK weights.  The general variant
// carries the joint-type blends, the tool product, RefPoint and the weight scaling.
// ROLL = closed-loop rollout: a.n_cycles control cycles in one launch, the joint angles
// integrated in registers (q += dt * qdot_out, the role of the external joint_sim, vfclik:99-103), the
// field set read from LDS every cycle; one launch boundary and one set of loads per n_cycles cycles.
// LEAN = nothing but q -> qdot_out (and status, which is free): no other optional input or output (and, without the
// nullspace module, no feature flag):
// the BASELINE C3 and C5 launches.  The arguments that select those options are compile-time nulls, so their code is
// not in the kernel at all -- present but never executed, it cost the C3 launch 4.5 % (6.47 -> 6.18 us, same box).
// amdgpu_waves_per_eu(1, 1): one wave per SIMD is what the launch gets anyway (registers, LDS); telling the backend
// lets its scheduler stop trading instruction order for a register count it cannot use (C3 -2.4 %, same-box A/B).
// CF = the feature flags as a compile-time constant, or -1: read from the launch.  The two flag sets the reference's
// default process set produces with the nullspace module (vfclik:95-97: nullspace + mixer; C5 adds the joint-limit
// task) get their own LEAN kernels, so that the branches of the other options are not in the code at all.
// PERS = persistent launch for batches beyond one wave per SIMD (lean straight-line launches, float I/O, chains of up to 7
// joints): the grid is one wave per SIMD, every wave strides over the 64-arm chunks of the batch, and while it computes
// chunk k the requests of chunk k + 1 -- q, goal block, slot quads -- are in flight into a SECOND per-arm area of its LDS
// region (they are issued between the joints of the kinematics, where the first chunk's slot requests go).  Launched in
// rounds of one wave per SIMD instead, 131 072 arms cost 2.33 x the 65 536-arm launch: every round pays its own request
// phase, q round trip and tail.
// Lean single-cycle launches on the straight-line path take the 56-byte KLean instead of the 340-byte KArgs (vfik_kernel.h): the
// handle's state is one arena whose layout follows from (io type, joints, Bpad).  (The diagnostic stamps build keeps KArgs.)
#ifndef VFIK_Q_FIRST
#define VFIK_Q_FIRST 0            // 1: q's pieces requested in front of the constants (A/B builds: no gain, profiles/r03_ab_experiments.md 18)
#endif
#ifndef VFIK_NT_STORES
#define VFIK_NT_STORES 0          // 1: the lane-by-lane output stores non-temporal (A/B builds)
#endif
#ifndef VFIK_SCALAR_KERNARG
#define VFIK_SCALAR_KERNARG 1     // 0: every kernel takes its argument block by value, as until round 3 (A/B builds)
#endif
#ifdef VFIK_STAMPS
template <int LEAN, bool ROLL, bool FASTF, bool MIXO = false> struct SmallArgs { static constexpr bool value = false; };
#else
// (MIXO variants keep the argument block: the order planes' address is one more pointer than KLean's fourteen dwords hold)
template <int LEAN, bool ROLL, bool FASTF, bool MIXO = false> struct SmallArgs { static constexpr bool value = LEAN == 1 && !ROLL && FASTF && !MIXO; };
#endif
template <typename T, int NJ> struct ArenaLayout {
    __host__ __device__ static long funnel_off(long Bpad) { return 4 * Bpad * 4 * (long)sizeof(T); }
    __host__ __device__ static long kconst_off(long Bpad) { return (4 + 6) * Bpad * 4 * (long)sizeof(T); }
    __host__ __device__ static long lastvec_off(long Bpad) { return kconst_off(Bpad) + VFIK_KCONST_SLOT(KTab<NJ>::OFFSET + 1024); }
    __host__ __device__ static long slots_fast_off(long Bpad) { return lastvec_off(Bpad) + (long)((NJ + 4) / 4) * Bpad * 16; }
};

// FUN = the straight-line field path with an AUX BLOCK: besides goal + decay repellers an arm may carry one funnel attractor and one
// hemisphere repeller (integer decay orders) -- the goalAndNormal scene of the object feeder (object_feeder:248-303: attractor + approach funnel +
// near-goal repeller + obstacles), which handlers.go_cart with a normal produces, and a surface (ObstacleH, object_feeder:344-353).
// Their 12 + 10 scalars travel like the goal block (6 quad planes, requested right behind it) and are evaluated straight-line
// behind the attractor; on the general path the same
// scene cost the C3 batch 8.8 instead of 5.5 us (the funnel is a ~200-instruction dependent chain evaluated entry by entry).
// WAVES = 2: the same lean launch compiled for TWO waves per SIMD (at most 256 registers a lane; with float I/O two blocks' lean
// regions, 2 x 79 KB, fit a CU's LDS), for batches beyond one wave per SIMD: the second wave issues into the first one's dependency
// stalls -- 131 072 arms 10.3 -> 9.0 us, 524 288 arms 38.3 -> 34.6 us, same box (profiles/r03_batch_scaling.txt).
// UNI = the straight-line path reading the UNIFORM repeller image: every decay repeller of the batch has the same safe distance and
// force (object_feeder sends 0.001 and -10 for every point obstacle and for the near-goal repeller: object_feeder:301-302,323,331), so a
// slot is ONE quad (x y z radius) and the pair travels in the constants -- 16 instead of 24 bytes and two thirds of the requests.
// MIXO = the straight-line path for decay repellers whose INTEGER orders differ, between slots or between arms (old/README.old:75
// documents `ObstacleP ... 0.05 20` while the feeder's near-goal repeller has order 5, object_feeder:302).  Every arm carries one byte
// per compact-image slot in the order planes (vfik_kernel.h), requested behind the goal block; a wave whose 64 arms agree slot by
// slot -- every scene in which the orders differ by obstacle, not by arm -- raises each slot to ITS power under scalar control
// flow, the slots in lock step; a wave with an odd arm falls back to per-lane selects over the bits of its largest order.  Until
// round 4 two different orders anywhere in the batch sent the whole batch to the general path (C3-sized: 7.1 instead of 5.3 us).
// DHP = the chain's DH pattern as compile-time masks (vfik_kernel.h: DhPattern), decided on the host like PLAIN: the KUKA LWR -- vfclik's
// default robot (vfclik:42) -- has a = 0 on every link and alpha = +-pi/2 on six of seven; such a link's rotation about x is a renaming
// with a sign instead of 12 operations, its a-term vanishes, and links with d = 0 skip the offset: 114 of the lean C3 kernel's 1 269
// vector instructions (C3 -5.3 % warm / -3.7 % cold, C3N -3.1 %, C3F -2.7 %: profiles/r04_ab_dhp.txt, before the swap lost its last
// multiplications).  Built for the lean and the publishing-lean straight-line float variants (dhp_of); every other variant, and every
// chain that does not match a built pattern, runs the general DH form.
template <typename T, int NJ, bool NULLSP, bool PLAIN, bool ROLL, bool FASTF, int LEAN, int CF, bool PERS, bool FUN, int WAVES, bool UNI, bool MIXO, int DHP>
__device__ __forceinline__ void
cycle_body(const typename std::conditional<SmallArgs<LEAN, ROLL, FASTF, MIXO>::value, KLean, KArgs>::type& a_in) {
    static_assert(DHP == 0 || (PLAIN && FASTF && (!ROLL || LEAN == 1) && !PERS && LEAN != 0 && (sizeof(T) == 4 || DHP == 1)), "DHP: the lean straight-line variants (single cycle, a stepped rollout's cycle, and the lean rollout); the option bits: float32 I/O");
    static_assert(!(DHP & 2) || LEAN != 2, "TOOLC: not in a stepped rollout's cycle");
    static_assert(!(DHP & 2) || !ROLL, "TOOLC: single-cycle variants only (a rollout with a tool is stepped)");
    // DHP bit 0: the chain's DH pattern (DhPattern<NJ, 1>); bit 1 (TOOLC): ONE tool for the whole batch, applied by the PLAIN kernel --
    // vfclik's normal state is an arm with a hand on it (`set tool`, old/README.old:84; vf:321-332), and until round 4 any tool sent the
    // launch to the general variants: C3 4.5 -> 8.0 us, C3N 6.6 -> 10.6 (profiles/r04_tool_cost.txt).  A compile-time property like the
    // pattern: as a run-time (wave-uniform) branch of the PLAIN kernels it cost every launch WITHOUT a tool 1-2 % (two more basic blocks in the
    // straight-line code; profiles/r04_ab_tool.txt).
    constexpr int DHPAT = DHP & 1;
    constexpr bool TOOLC = (DHP & 2) != 0;
    // DHP bit 2 (WTSC): the batch's IK weights (/weight, vf:295-309: Wy = diag of the 't' weights, Wq of the 'j' weights) other than one, on the
    // PLAIN kernel of a chain of up to 7 joints -- the weighted normal matrix and the two scalings, ~80 instructions, instead of the general
    // variants' run-time everything (C3N 6.7 -> 10.6 us).  WEIGHTED = what the general variants always do.
    constexpr bool WTSC = (DHP & 4) != 0;
    static_assert(!WTSC || (NJ <= 7 && !ROLL && LEAN != 2), "WTSC: single-cycle variants of chains of up to 7 joints");
    constexpr bool WEIGHTED = !PLAIN || WTSC;
    static_assert(!MIXO || (FASTF && PLAIN && !ROLL && !PERS && !UNI && WAVES == 1 && (LEAN == 1 || LEAN == 3)), "MIXO: the lean single-cycle straight-line variants");
    static_assert(!PERS || (LEAN == 1 && FASTF && PLAIN && !ROLL && sizeof(T) == 4 && NJ <= 7 && !FUN), "PERS: lean straight-line float launches only");
    static_assert(!FUN || (FASTF && PLAIN && !ROLL && (LEAN == 1 || LEAN == 3)), "FUN: the lean single-cycle straight-line variants");
    static_assert(WAVES == 1 || (WAVES == 2 && LEAN == 1 && FASTF && PLAIN && !ROLL && !PERS && !FUN && sizeof(T) == 4 && NJ <= 7), "WAVES 2: lean straight-line float launches only");
    static_assert(!UNI || (FASTF && PLAIN && !ROLL && !PERS && !FUN && (LEAN == 1 || LEAN == 3)), "UNI: the lean single-cycle straight-line variants");
    KArgs a;
    if constexpr (SmallArgs<LEAN, ROLL, FASTF, MIXO>::value) {
        a = KArgs{};
        a.B = a_in.B; a.Bpad = a_in.Bpad; a.slots_used = a_in.slots_used; a.fast_order = a_in.fast_order; a.flags = a_in.flags; a.block = a_in.block;
        a.q = a_in.q; a.qdot_out = a_in.qdot_out; a.status = a_in.status;
        const char* const base = static_cast<const char*>(a_in.base);
        a.goal = base;
        a.funnel = base + ArenaLayout<T, NJ>::funnel_off(a_in.Bpad);
        a.kc = base + ArenaLayout<T, NJ>::kconst_off(a_in.Bpad);
        a.lastvec = reinterpret_cast<float*>(const_cast<char*>(base) + ArenaLayout<T, NJ>::lastvec_off(a_in.Bpad));
        a.slots_fast = base + ArenaLayout<T, NJ>::slots_fast_off(a_in.Bpad);
        a.slots = a.slots_fast;  // (never read on the straight-line path)
    } else {
        a = a_in;
    }
    if constexpr (UNI) {  // the uniform image sits behind the compact one: its offset in quad planes rides in fast_order's upper bits
        a.slots_fast = static_cast<const char*>(a.slots_fast) + (long)(a.fast_order >> 8) * a.Bpad * (4 * (long)sizeof(T));
        a.fast_order &= 255;
    }
    if constexpr (CF >= 0) a.flags = (unsigned)CF;
    // LEAN: 1 = lean, 2 = lean with q_out kept (one cycle of a stepped rollout, long chains), 3 = PUBLISHING lean: no per-arm
    // option either, but the rows the per-arm processes publish (qdot_vf, qdot_null, pose, pose_nt, v6, qdist, goal_dist),
    // /control and the fresh-q gate stay run-time -- what ControlCycleBatch's default cycle and an array caller that wants
    // the poses ask for.  The fully general variant (LEAN 0) costs such a launch ~1 700 cycles a wave in options it does not use.
    if constexpr (LEAN != 0) {
        if constexpr (!NULLSP) a.flags = 0;  // (with the nullspace module the flags are run-time unless CF fixes them)
        a.tool_stride = 0; a.mixw = nullptr; a.wts = nullptr; a.ext = nullptr;
        a.q_ref = nullptr; a.q_cmded = nullptr; a.q_lo = nullptr; a.q_hi = nullptr; a.q_ref_out = nullptr;
        if constexpr (LEAN != 3) {
            a.null_control = nullptr; a.qdot_vf = nullptr; a.qdot_null = nullptr; a.pose = nullptr; a.pose_nt = nullptr;
            a.v6 = nullptr; a.qdist = nullptr; a.goal_dist = nullptr; a.active = nullptr;
        } else {
            a.status_or = 0; a.q_out = nullptr;
        }
        if constexpr (LEAN == 1) a.status_or = 0;  // (a stepped rollout accumulates the status bits of its cycles)
        if constexpr (!ROLL && LEAN == 1) a.q_out = nullptr;  // (a rollout's q_out is its result)
    }
    // Fetch the kernel arguments the prologue needs with one batch of scalar loads: left to itself the
    // compiler loads them one by one, each time waiting out a full scalar-load latency.
    // (Entered through cycle_kernel_s / cycle_kernel_x they arrive preloaded in SGPRs, and the rest of the block is best left where
    // the compiler first needs it.)
    if constexpr (!VFIK_SCALAR_KERNARG)
        asm volatile("" ::"s"(a.B), "s"(a.block), "s"(a.Bpad), "s"(a.slots_used), "s"(a.tool_stride), "s"(a.q), "s"(a.goal), "s"(a.slots), "s"(a.slots_fast),
                     "s"(a.tool), "s"(a.mixw), "s"(a.kc));
    // PERS: one wave per block; chunk = 64 consecutive arms; lanes past the end of the batch compute arm B - 1 again and store nothing
    const int nchunks = (a.B + 63) >> 6;
    int chunk = PERS ? (int)blockIdx.x : 0;
    int arm = PERS ? chunk * 64 + (int)threadIdx.x : (int)(blockIdx.x * a.block + threadIdx.x);
    const long Bs = a.B;
    constexpr unsigned DHP_SWAP = DhPattern<NJ, DHPAT>::SWAP, DHP_NONE = DhPattern<NJ, DHPAT>::NONE, DHP_D0 = DhPattern<NJ, DHPAT>::D0;
    constexpr bool TABSC = NJ <= 8;  // sin / cos through the LDS table (sincos_tab_n)
    constexpr bool NTL = NJ >= VFIK_NT_MIN_NJ;   // non-temporal policy for the per-arm input planes (stage_quad)
    // batch constants through the constant address space: always scalar loads
    typedef const KConst<NJ> __attribute__((address_space(4))) * KcPtr;
    const KcPtr kc_launch = (KcPtr)(unsigned long long)a.kc;
    int status = 0;
    STAMP(0);

    // ---------------- loads: request everything the cycle needs, straight into LDS ------------
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction; say so, or every LDS destination goes through a VGPR + readfirstlane
    // (LEAN launches run the straight-line path and touch only the head of the region: their waves are packed closer)
    constexpr int FUN_OFF = Stage<T>::lean_bytes(NJ);       // FUN: the aux block's rows sit behind the lean region
    constexpr int NFUN = FUN ? 6 : 0;                       // (funnel 3 + hemisphere 3)
    constexpr int NORD = MIXO ? 1 : 0;                      // MIXO: one 1-KiB row for the chunk's 16 order bytes per arm, behind the aux rows
    constexpr int ORD_OFF = FUN_OFF + NFUN * Stage<T>::QSTEP;
    constexpr int REGION_BYTES = ((LEAN != 0 && FASTF) ? Stage<T>::lean_bytes(NJ) : Stage<T>::bytes(NJ)) + NFUN * Stage<T>::QSTEP + NORD * 1024;
    char* const region = lds_all + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * REGION_BYTES;
    // per-arm inputs (goal block, first-chunk slot quads, q) of the chunk being computed: the head of the region, or
    // (PERS, odd chunks of the wave) the second per-arm area behind the constants and the table
    char* dreg = region;
    constexpr int DAREA2 = Stage<T>::lean_bytes(NJ);      // offset of the second per-arm area; it is kin_off(NJ) bytes long
    const long Bp = a.Bpad;
    constexpr int QB = Stage<T>::QBYTES, Q16 = Stage<T>::Q16;
    constexpr int PRE = Stage<T>::PRE;
    const long planeB = Bp * QB;  // bytes of one quad plane
    constexpr int NQREQ = Stage<T>::q16(NJ) + Stage<T>::qrem(NJ) / 4;   // requests that bring one q vector
    auto issue_q_piece = [&](int r, int armx, char* dr) {  // piece r of arm armx's q into the per-arm area dr
#if VFIK_SADDR_REQUESTS
        const char* const q0 = static_cast<const char*>(a.q);            // (uniform base + unsigned lane offset: stage_quad)
        unsigned qv = (unsigned)armx * (unsigned)(NJ * sizeof(T));
        asm("" : "+v"(qv));
        char* qrow = dr + Stage<T>::Q_OFF;
        constexpr int n16 = Stage<T>::q16(NJ);
        if (r < n16) __builtin_amdgcn_global_load_lds((GPtr)(q0 + r * 16 + qv), (LPtr)(qrow + r * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((GPtr)(q0 + (n16 * 16 + (r - n16) * 4) + qv), (LPtr)(qrow + n16 * 1024 + (r - n16) * 256), 4, 0, 0);
#else
        const char* qg = static_cast<const char*>(a.q) + (long)armx * NJ * sizeof(T);
        char* qrow = dr + Stage<T>::Q_OFF;
        constexpr int n16 = Stage<T>::q16(NJ);
#ifndef VFIK_Q_NT
#define VFIK_Q_NT 0               // 1: q's pieces with the non-temporal policy too (A/B builds)
#endif
        if (r < n16) __builtin_amdgcn_global_load_lds((GPtr)(qg + r * 16), (LPtr)(qrow + r * 1024), 16, 0, VFIK_Q_NT ? 2 : 0);
        else __builtin_amdgcn_global_load_lds((GPtr)(qg + n16 * 16 + (r - n16) * 4), (LPtr)(qrow + n16 * 1024 + (r - n16) * 256), 4, 0, VFIK_Q_NT ? 2 : 0);
#endif
    };
    // (Round 3 tried q's pieces FIRST, in front of the constants -- q is the one input the wave cannot start without, and with the
    // inputs in HBM the one it waits for longest: C3 / C3N +-0.1 % cold and warm, C5 +0.9 % cold.  Two requests earlier is nothing
    // against a round trip.)
    constexpr bool QFIRST = VFIK_Q_FIRST && !PERS;
    if constexpr (QFIRST) {
        if (arm < a.B) {
#pragma unroll
            for (int r = 0; r < NQREQ; ++r) issue_q_piece(r, arm, dreg);
        }
    }
    {   // kinematics constants (oldest request: covered by the first wait).  All 64 lanes copy
        // 16 bytes each, so this comes before the lanes past the end of the batch retire.
        const char* kg = static_cast<const char*>(a.kc) + lane * 16;
#pragma unroll
        for (int r = 0; r < KConst<NJ>::KIN_ROWS; ++r)
            __builtin_amdgcn_global_load_lds((GPtr)(kg + r * 1024), (LPtr)(region + Stage<T>::kin_off(NJ) + r * 1024), 16, 0, 0);
        // the sin / cos table (64 x 16 bytes) sits behind the constants, on the next 1-KiB boundary
        if constexpr (TABSC) __builtin_amdgcn_global_load_lds((GPtr)(kg + KTab<NJ>::OFFSET), (LPtr)(region + Stage<T>::tab_off(NJ)), 16, 0, 0);
    }
    // q as ONE BLOCK per wave (round 4).  q is batch-major ([B][n]): the 64 arms of a wave own 64 n sizeof(T) CONTIGUOUS bytes -- 1 792 for
    // seven float joints.  Requested arm by arm that was four requests (one 16-byte piece and three 4-byte pieces per lane, a lane's row not being
    // 16-byte sized); as a block it is ceil(n sizeof(T) / 16) requests of 16 bytes per lane that land in the q area as they lie in memory, and
    // every lane reads ITS row from there (row stride n sizeof(T): 7 dwords, odd, conflict-free).  Two requests instead of four at ~70 cycles of
    // the issuing wave each.  The lanes fetch pieces, not rows, so this happens before the lanes past the end of the batch retire; pieces past
    // the batch's last row are not requested.
#ifndef VFIK_Q_BLOCK
#define VFIK_Q_BLOCK 1            // 0: q arm by arm, as until round 3 (A/B builds)
#endif
    // (the lean single-cycle variants at one wave per SIMD: elsewhere the few registers of the block form tipped variants that sit at their
    // register limit into scratch -- 12 to 130 B per lane in seven of them)
    constexpr bool QBLK = VFIK_Q_BLOCK && !PERS && !QFIRST && !ROLL && LEAN != 0 && WAVES == 1;
    if constexpr (QBLK) {
        constexpr int BQ = NJ * (int)sizeof(T);              // bytes of one row
        constexpr int NBLK = (BQ + 15) / 16;                 // requests: 64 BQ bytes in pieces of 64 x 16
        const int w0 = arm - lane;                           // the wave's first arm
        const int rows = (a.B - w0) < 64 ? (a.B - w0) : 64;
        const char* qb = static_cast<const char*>(a.q) + (long)w0 * BQ + lane * 16;
        char* qrow = dreg + Stage<T>::Q_OFF;
#pragma unroll
        for (int r = 0; r < NBLK; ++r)
            if ((r * 64 + lane) * 16 < rows * BQ)
                __builtin_amdgcn_global_load_lds((GPtr)(qb + r * 1024), (LPtr)(qrow + r * 1024), 16, 0, 0);
    }
    if constexpr (!PERS) {
        if (arm >= a.B) return;
    }
    // Fresh-q gate (vf:312-313, nullspace:162-163): an arm whose joint angles did not arrive this cycle stores
    // nothing.  Requested first, consumed at the stores: every counted wait below covers this oldest request.
    int act = 1;
    if constexpr (PERS) {
        act = arm < a.B;
        arm = arm < a.B ? arm : a.B - 1;
    }
    if (a.active) act = a.active[arm];
    if (a.tool_stride) {          // per-arm tools ([3][Bpad] quads); a shared tool sits in KConst
        const char* tg = static_cast<const char*>(a.tool);
#pragma unroll
        for (int k = 0; k < 3; ++k) stage_quad<T, NTL>(tg + k * planeB, (unsigned)arm * (unsigned)QB, region, Stage<T>::tool_off(NJ) + k * Stage<T>::QSTEP);
    }
    if (a.mixw) {  // per-arm mixer weights ([2][Bpad] quads: w0..w3 | w4 w5 - -); else KConst::mix_w
        const char* mg = static_cast<const char*>(a.mixw);
#pragma unroll
        for (int k = 0; k < 2; ++k) stage_quad<T, NTL>(mg + k * planeB, (unsigned)arm * (unsigned)QB, region, Stage<T>::mixw_off(NJ) + k * Stage<T>::QSTEP);
    }
    if constexpr (!QFIRST && !QBLK) {
#pragma unroll
        for (int r = 0; r < NQREQ; ++r) issue_q_piece(r, arm, dreg);
    }
    // The goal and slot requests are issued later, between the joints of the kinematics: a load costs
    // the issuing wave ~50 cycles while the CU's four waves queue on the one address unit
    // (tools/ubench_loads.hip), and spreading them out lets that queue drain under arithmetic.
    const int npre = a.slots_used < PRE ? a.slots_used : PRE;
    // The straight-line path reads the COMPACT repeller image (two slots in three quads, vfik_kernel.h): a chunk of PRE
    // slots is QPC = 3 PRE / 2 quads instead of 2 PRE -- a quarter fewer bytes and requests for what is, at these
    // batches, the longest wait of the wave (the slots' data is the last to arrive).
    constexpr int QPC = UNI ? PRE : (FASTF ? 3 * PRE / 2 : 2 * PRE);       // slot quads per chunk
    const char* const goal0 = static_cast<const char*>(a.goal);
    const char* const slots0 = static_cast<const char*>(FASTF ? a.slots_fast : a.slots);
    const char* sg = slots0 + (long)arm * QB;  // this arm's quad of slot plane 0
    auto issue_goal_quad = [&](int k, int armx, char* dr) {
        stage_quad<T, NTL>(goal0 + k * planeB, (unsigned)armx * (unsigned)QB, dr, Stage<T>::GOAL_OFF + k * Stage<T>::QSTEP);
    };
    auto issue_slot_quad_of = [&](int idx, int armx, char* dr) {  // idx in [0, QPC): quad idx of the first chunk of slots
        // quads past the slots in use re-request plane 0 (cache hit) and are masked below, so the
        // number of outstanding requests is a compile-time constant for the counted waits
        // (Round 3 tried a wave-uniform fast path without the per-quad compare / select when every slot of the window is in use --
        // ~60 fewer scalar instructions a wave: C3 +1.2 % warm, +0.3 % cold, C5 +-0.1 %; scalar work is not what a lone wave waits for.)
        // (UNI: plane 0 of the uniform image is an EMPTY slot for every arm -- radius -inf --, slot m sits in plane m + 1: a quad past the
        // slots in use is then harmless by itself and the chunk needs no mask)
        const bool in = UNI ? idx < npre : (FASTF ? 2 * (idx / 3) < npre : (idx >> 1) < npre);
        stage_quad<T, NTL>(slots0 + (in ? (long)(UNI ? idx + 1 : idx) * planeB : 0), (unsigned)armx * (unsigned)QB, dr, Stage<T>::slot_off(idx, NJ));
    };
    auto issue_slot_quad = [&](int idx) { issue_slot_quad_of(idx, arm, dreg); };
    constexpr int N_SLOT = QPC * Q16;                       // requests issued after the goal
    // The wave has to sit out q's round trip (~500 cycles) anyway: the goal block and the first EARLY_Q
    // slot quads are requested into that wait, the remaining slot quads between the joints.
    // (long chains have two rows of constants and five q pieces in front already: nothing early there, C5 -1.3 %)
    // PERS: the first chunk's requests all go out here; the slots between the joints are the NEXT chunk's requests.
    // Round 4: NO slot quad goes out early any more (until then six of them were requested into q's round trip).  With the inputs in HBM every
    // wave's ~20 requests of the launch's first half microsecond compete for the same bandwidth -- 16.5 MB in all, 2-3 us of HBM time -- and
    // what a wave needs FIRST (q: 1.8 MB over the batch, then the goal block) queued behind slot data it needs last: slot quads requested
    // between the joints instead, C3 cold 5.57 -> 5.21 us (-6.5 %), three early 5.44; warm +-0 (profiles/r04_ab_early_q.txt).
#ifndef VFIK_EARLY_Q
#define VFIK_EARLY_Q 0            // slot quads requested into q's round trip (A/B builds)
#endif
    constexpr int EARLY_Q = PERS ? QPC : (NJ >= 10 ? 0 : (VFIK_EARLY_Q < QPC ? VFIK_EARLY_Q : QPC));
#ifndef VFIK_GOAL_LATE
#define VFIK_GOAL_LATE 0          // 1: the goal block requested behind the first joint's transform instead of in front of the kinematics (A/B builds)
#endif
#ifndef VFIK_SLOT_JOINT0
#define VFIK_SLOT_JOINT0 0        // first joint behind which slot quads are requested (A/B builds)
#endif
    constexpr bool GOAL_LATE = VFIK_GOAL_LATE && !PERS;
    constexpr int SJ0 = (VFIK_SLOT_JOINT0 < NJ - 1 && !PERS) ? VFIK_SLOT_JOINT0 : 0;
    constexpr int SLOTQ_PER_JOINT = (QPC - EARLY_Q + (NJ - SJ0) - 1) / (NJ - SJ0);  // slot quads requested after each joint
    // PERS: request r of a chunk's NPF = q pieces, goal quads, slot quads, in that order (float I/O: one request a quad)
    constexpr int NPF = PERS ? NQREQ + 4 + QPC : 0;
    constexpr int PF_PER_JOINT = (NPF + NJ - 1) / NJ;
    auto issue_prefetch = [&](int r, int armx, char* dr) {
        if (r < NQREQ) issue_q_piece(r, armx, dr);
        else if (r < NQREQ + 4) issue_goal_quad(r - NQREQ, armx, dr);
        else issue_slot_quad_of(r - NQREQ - 4, armx, dr);
    };
    if constexpr (!GOAL_LATE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) issue_goal_quad(k, arm, dreg);
    }
    if constexpr (FUN) {  // the funnel block, right behind the goal block: the goal's wait covers it
        const char* fg = static_cast<const char*>(a.funnel);
#pragma unroll
        for (int k = 0; k < NFUN; ++k) stage_quad<T, NTL>(fg + k * planeB, (unsigned)arm * (unsigned)QB, region, FUN_OFF + k * Stage<T>::QSTEP);
    }
    // MIXO: the 16 order bytes that cover slots [c0 & ~15, +16) of this arm -> the order row (one request whatever T)
    auto issue_orders = [&](int c0) {
        const char* og = static_cast<const char*>(a.orders) + (long)(c0 >> 4) * Bp * 16 + (long)arm * 16;
        __builtin_amdgcn_global_load_lds((GPtr)og, (LPtr)(region + ORD_OFF), 16, 0, NTL ? 2 : 0);
    };
    if constexpr (MIXO) issue_orders(0);  // (behind the goal / aux blocks: the goal's wait covers it)
#pragma unroll
    for (int idx = 0; idx < EARLY_Q; ++idx) issue_slot_quad(idx);

    // The members of the argument block the rest of the cycle needs, in one batch of scalar loads behind the requests (cycle_kernel_x:
    // left alone, the compiler fetches each where it is first used -- a round trip to the kernarg segment every time)
    // (not in the non-lean variants of the long chains: 34 scalar registers pinned from here to the epilogue spill to vector lanes in kernels
    // that have no vector register to spare -- part of what took their scratch away, profiles/r04_ab_experiments.md C)
    if constexpr (VFIK_SCALAR_KERNARG && !SmallArgs<LEAN, ROLL, FASTF, MIXO>::value && !(LEAN == 0 && NJ >= 12))
        asm volatile("" ::"s"(a.null_control), "s"(a.qdot_vf), "s"(a.qdot_null), "s"(a.pose), "s"(a.pose_nt), "s"(a.v6), "s"(a.qdist), "s"(a.goal_dist),
                     "s"(a.status), "s"(a.q_out), "s"(a.ext), "s"(a.q_ref), "s"(a.q_cmded), "s"(a.q_lo), "s"(a.q_hi), "s"(a.q_ref_out), "s"(a.wts));
    STAMP(1);
    // ---------------- A3: forward kinematics (vf:316-318) -------------------------------------
    double q[NJ], sn[NJ], cs[NJ];
    const KConst<NJ>* const kl = reinterpret_cast<const KConst<NJ>*>(region + Stage<T>::kin_off(NJ));  // kinematics block only
    bool has_next = false;    // PERS: this wave has another chunk of arms after the current one (wave-uniform)
    int arm_next = 0;         // PERS: this lane's arm of that chunk (clamped to the batch)
    char* dreg_next = region;
    auto read_q = [&]() {
        const char* qrow = dreg + Stage<T>::Q_OFF;
        if constexpr (QBLK) {   // the wave's rows as they lie in memory
            const T* mine = reinterpret_cast<const T*>(qrow + lane * (NJ * (int)sizeof(T)));
#pragma unroll
            for (int i = 0; i < NJ; ++i) q[i] = (double)mine[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            if (Q16 == 1) {
                q[i] = (double)*reinterpret_cast<const float*>(qrow + q_lds_off<T, NJ>(i * 4, lane));
            } else {
                const int lo = *reinterpret_cast<const int*>(qrow + q_lds_off<T, NJ>(i * 8, lane));
                const int hi = *reinterpret_cast<const int*>(qrow + q_lds_off<T, NJ>(i * 8 + 4, lane));
                q[i] = __hiloint2double(hi, lo);
            }
        }
    };
    if constexpr (!PERS) {
        VFIK_WAIT_VM(((GOAL_LATE ? 0 : 4) + NFUN + EARLY_Q) * Q16 + NORD);  // constants, table, tool and q have landed (the goal, funnel, order and early slot requests may still be out)
        STAMP(2);
        read_q();
    }
    // nullspace sign memory (nullspace:91-92) lives in registers across the cycles of a launch
    int sig_r = 1;
    bool has_vec = false;  // lastvec holds a vector (false until the first cycle with a unique nullspace direction)
    double lv_r[NJ];
    // Nullspace state of an arm (nullspace:91-92), chains of up to 7 joints: lastvec (n values) and sig, kept as
    // FLOATs in (n + 4) / 4 planes of 16 bytes per arm -- element n holds sig as +-1 (no vector stored yet) or +-2
    // (lastvec holds one).  The state only steers decisions (the sign continuity test) and seeds the projection
    // below; the vector that is published is recomputed in float64 every cycle.  As doubles the state was 4.2 MB
    // written per C3 launch: 0.3 us of the launch period (measured by leaving the store out).
    constexpr int NS_PLANES = (NJ + 4) / 4;
    typedef float f4s __attribute__((ext_vector_type(4)));
    auto store_null_state = [&]() {
        f4s* sp = reinterpret_cast<f4s*>(a.lastvec) + arm;
        float sv[NS_PLANES * 4];
#pragma unroll
        for (int i = 0; i < NS_PLANES * 4; ++i) sv[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < NJ; ++i) sv[i] = (float)lv_r[i];
        sv[NJ] = (float)sig_r * (has_vec ? 2.0f : 1.0f);
#pragma unroll
        for (int k = 0; k < NS_PLANES; ++k) {
            f4s v;
            v.x = sv[4 * k]; v.y = sv[4 * k + 1]; v.z = sv[4 * k + 2]; v.w = sv[4 * k + 3];
            sp[(long)k * a.Bpad] = v;
        }
    };
    // The state is requested a phase ahead of its use (read where it is used, the round trip stood in the wave's
    // way, ~1 500 cycles): a rollout loads it here, once; a single cycle requests it behind the field evaluation, in
    // front of the IK's solve, which covers the latency without the registers being held through kinematics and field.
    // Chains of 8+ joints have nullity >= 2 and never use the sign memory.
    auto load_null_state = [&]() {
        const f4s* sp = reinterpret_cast<const f4s*>(a.lastvec) + arm;
        float sv[NS_PLANES * 4];
#pragma unroll
        for (int k = 0; k < NS_PLANES; ++k) {
            const f4s v = sp[(long)k * a.Bpad];
            sv[4 * k] = v.x; sv[4 * k + 1] = v.y; sv[4 * k + 2] = v.z; sv[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) lv_r[i] = (double)sv[i];
        sig_r = sv[NJ] < 0.0f ? -1 : 1;
        has_vec = fabsf(sv[NJ]) > 1.5f;
    };
    if constexpr (NULLSP && NJ <= 7 && ROLL) load_null_state();
    const int ncyc = ROLL ? a.n_cycles : 1;
    for (;;) {  // PERS: the wave's chunks of arms; otherwise one pass
    if constexpr (PERS) {
        // Top of a chunk.  First chunk: constants, table and q have landed (goal and slots may still be out).  Later chunks:
        // everything landed before the previous chunk's stores (the wait in front of them), and those few stores are
        // younger than anything this wait looks at.
        VFIK_WAIT_VM((4 + QPC) * Q16);
        STAMP(2);
        read_q();
        status = 0;
        const int nxt = chunk + (int)gridDim.x;
        has_next = nxt < nchunks;
        const int an = nxt * 64 + (int)threadIdx.x;
        arm_next = an < a.B ? an : a.B - 1;
        dreg_next = dreg == region ? region + DAREA2 : region;
    }
    for (int cyc = 0; cyc < ncyc; ++cyc) {
    // The LDS addresses are made opaque once per cycle for long chains: otherwise the compiler hoists the
    // cycle-invariant reads (constants, goal, slots: ~200 doubles) out of the cycle loop and spills.
    const KConst<NJ>* klc = kl;
    int lanec = lane;
    KcPtr kc = kc_launch;
    if (ROLL && (NJ >= 10 || !PLAIN)) {  // long chains / the general variant have no registers to spare for cycle-invariant copies of the inputs
        // (an opaque ZERO added to the LDS address and the lane: the reads stay LDS reads -- an opaque pointer would turn them
        // into flat loads -- but the compiler can no longer prove them cycle-invariant)
        int zero_v = 0, zero_s = 0;
        asm volatile("" : "+v"(zero_v));
        asm volatile("" : "+s"(zero_s));
        klc = reinterpret_cast<const KConst<NJ>*>(reinterpret_cast<const char*>(kl) + zero_s);
        lanec = lane + zero_v;
        asm volatile("" : "+s"(kc));  // nor SGPRs for ~70 hoisted scalar constants (they would spill through VGPR lanes)
    }
    // Joint limits of this cycle: the arm's own (io.q_lo / q_hi; nullspace:167 and joint_p_controller:80 re-read
    // them every cycle) or the chain's.  Fetched where they are used, all joints in one go.
    auto limits_of = [&](double* lo, double* hi) {
        if (a.q_lo) {
            const T* l = static_cast<const T*>(a.q_lo) + (long)arm * NJ;
            const T* h = static_cast<const T*>(a.q_hi) + (long)arm * NJ;
#pragma unroll
            for (int i = 0; i < NJ; ++i) { lo[i] = (double)l[i]; hi[i] = (double)h[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < NJ; ++i) { lo[i] = kc->q_lo[i]; hi[i] = kc->q_hi[i]; }
        }
    };
    // z = -jl_gain (q - mid) / half^2, the descent direction of the joint-limit potential
    auto jl_descent = [&](double* z) {
        if (a.q_lo) {
            double lo[NJ], hi[NJ];
            limits_of(lo, hi);
            const double g = kc->jl_gain;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const double ih = rcp_nr(0.5 * (hi[i] - lo[i]));
                z[i] = -g * (q[i] - 0.5 * (lo[i] + hi[i])) * ih * ih;
            }
        } else {  // constants fetched together under the one uniform branch
            double jk[NJ], qm[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) { jk[i] = kc->jl_k[i]; qm[i] = kc->q_mid[i]; }
#pragma unroll
            for (int i = 0; i < NJ; ++i) z[i] = -jk[i] * (q[i] - qm[i]);
        }
    };
    const bool first = !ROLL || cyc == 0;
    if (ROLL && !first && a.slots_used > PRE) {  // the rows hold the last chunk of the previous cycle
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int idx = 0; idx < QPC; ++idx) issue_slot_quad(idx);
    }
    {
        // No libm fallback: the three-part reduction keeps full accuracy to |angle| ~ 1e5 rad and degrades
        // smoothly beyond (error ~ |angle| * 1e-21); NaN / Inf propagate and are flagged VFIK_ST_NAN.
        // (DH pattern: an offset that is an even multiple of pi is no offset, an odd one negates sine and cosine below)
        constexpr unsigned DHP_OFFK = DhPattern<NJ, DHPAT>::OFF0 | DhPattern<NJ, DHPAT>::OFFPI;
        double ang[NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i) ang[i] = (PLAIN && ((DHP_OFFK >> i) & 1u)) ? q[i] : q[i] + klc->dh[i].off;
        // Chains of up to 8 joints go through the LDS table (C3 -2 %, C3N -3.5 %, C2 -2.5 % in same-box A/Bs).  Long
        // chains keep the table-free form: the 14-joint kernel has no registers for 28 table values on top of its
        // Jacobian and was 2-4 % SLOWER with the table, all angles at once, in two halves or inside the kinematics.
        if constexpr (TABSC) sincos_tab_n<NJ>(ang, reinterpret_cast<const char*>(klc) + (Stage<T>::tab_off(NJ) - Stage<T>::kin_off(NJ)), sn, cs);
        else sincos_fast_n<NJ>(ang, sn, cs);
#pragma unroll
        for (int i = 0; i < NJ; ++i)
            if (PLAIN && ((DhPattern<NJ, DHPAT>::OFFPI >> i) & 1u)) { sn[i] = -sn[i]; cs[i] = -cs[i]; }
    }
    double R[9], p[3];
    constexpr bool BASE_I = PLAIN && DhPattern<NJ, DHPAT>::BASE_I;   // the base frame is the identity: joint 1 starts from unit vectors
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[3 * r + c] = BASE_I ? (r == c ? 1.0 : 0.0) : klc->base[4 * r + c];
        p[r] = BASE_I ? 0.0 : klc->base[4 * r + 3];
    }
    // Jm[i] = column i of the 6 x n Jacobian (rows 0..2 linear, 3..5 angular); during the kinematics
    // it first holds the joint origin and axis.  The nullspace module later orthonormalises its ROWS in
    // place (the columns are no longer needed once the IK has used them).
    double Jm[NJ][6];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        // joint i in DH form: Screw_z(angle, disp) Tx(a) Rx(alpha); ~33 flops, 9 constants from LDS
        Jm[i][3] = R[2]; Jm[i][4] = R[5]; Jm[i][5] = R[8];
        Jm[i][0] = p[0]; Jm[i][1] = p[1]; Jm[i][2] = p[2];
        const double ci = PLAIN ? cs[i] : __builtin_fma(klc->dh[i].crev, cs[i], klc->dh[i].cprs);
        const double si = PLAIN ? sn[i] : __builtin_fma(klc->dh[i].crev, sn[i], klc->dh[i].sprs);
        const double di = PLAIN ? klc->dh[i].d : __builtin_fma(klc->dh[i].qd, q[i], klc->dh[i].d);
        const double ai = klc->dh[i].a, ca = klc->dh[i].ca, sa = klc->dh[i].sa;
        // DH pattern of the chain, joint by joint, as compile-time masks (DHP): joints whose link has a = 0 and alpha = +-pi/2 (the
        // rotation about x is a swap with signs: 6 instead of 12 operations, no a-term), a = 0 and alpha = 0 (no rotation at all), d = 0
        // (i is a constant once the loop is unrolled: the tests fold away)
        const bool J_SWAP = PLAIN && ((DHP_SWAP >> i) & 1u), J_NONE = PLAIN && ((DHP_NONE >> i) & 1u), J_D0 = PLAIN && ((DHP_D0 >> i) & 1u);
        double xn[3], ym[3];
        if (BASE_I && i == 0) {   // R = I, p = 0: x' = (c, s, 0), y' = (-s, c, 0), p' = (0, 0, d) -- no arithmetic
            xn[0] = ci; xn[1] = si; xn[2] = 0.0;
            ym[0] = -si; ym[1] = ci; ym[2] = 0.0;
            if (!J_D0) p[2] = di;
        } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) { xn[r] = si * R[3 * r + 1]; ym[r] = si * R[3 * r]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { xn[r] = __builtin_fma(ci, R[3 * r], xn[r]); ym[r] = __builtin_fma(ci, R[3 * r + 1], -ym[r]); }
        if (!J_D0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) p[r] = __builtin_fma(di, R[3 * r + 2], p[r]);
        }
        }
        if (J_SWAP) {          // alpha = +pi/2 (the DH factorisation keeps sin alpha >= 0): y' = z, z' = -y -- no arithmetic at all
#pragma unroll
            for (int r = 0; r < 3; ++r) { R[3 * r] = xn[r]; R[3 * r + 1] = R[3 * r + 2]; R[3 * r + 2] = -ym[r]; }
        } else if (J_NONE) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { R[3 * r] = xn[r]; R[3 * r + 1] = ym[r]; }
        } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) { p[r] = __builtin_fma(ai, xn[r], p[r]); R[3 * r] = xn[r]; }
        double t1[3], t2[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { t1[r] = sa * R[3 * r + 2]; t2[r] = sa * ym[r]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { R[3 * r + 1] = __builtin_fma(ca, ym[r], t1[r]); R[3 * r + 2] = __builtin_fma(ca, R[3 * r + 2], -t2[r]); }
        }
        if (GOAL_LATE && first && i == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) issue_goal_quad(k, arm, dreg);
        }
#pragma unroll
        for (int k = 0; k < SLOTQ_PER_JOINT; ++k)
            if (first && i >= SJ0 && EARLY_Q + (i - SJ0) * SLOTQ_PER_JOINT + k < QPC) issue_slot_quad(EARLY_Q + (i - SJ0) * SLOTQ_PER_JOINT + k);
        if constexpr (PERS) {  // the next chunk's requests, into the other per-arm area, a few after every joint
            if (has_next) {
#pragma unroll
                for (int k = 0; k < PF_PER_JOINT; ++k)
                    if (i * PF_PER_JOINT + k < NPF) issue_prefetch(i * PF_PER_JOINT + k, arm_next, dreg_next);
            }
        }
    }
    if (!PLAIN) {   // trailing z-screw of the last fixed transform
        const double tc = klc->tail_c, ts = klc->tail_s, te = klc->tail_e;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double x = R[3 * r], y = R[3 * r + 1];
            p[r] += te * R[3 * r + 2];
            R[3 * r] = tc * x + ts * y;
            R[3 * r + 1] = tc * y - ts * x;
        }
    }
    // Chains of 8+ joints with the nullspace module: the projector of the joint-limit task,
    // z - J^T (J J^T)^-1 J z (see A10-A13 below), shares the IK's passes over the Jacobian.
    constexpr bool FUSEP = NULLSP && NJ >= 8;
    // ... except with IK weights or a tool (general variant), or on the general field path: the undamped Gram matrix G = J J^T and J z are then accumulated in a
    // pass of their own AFTER the IK's solve, when the weighted normal matrix is dead -- both matrices live at once spill
    // (float64 I/O, PLAIN, general path: 120 B of scratch without it, 108 with -- and slower; the non-lean float variant of 12+ joints on
    // the general path: 60 B with it, none without, under the flags that object is built with -- Makefile, HEAVY)
    constexpr bool GLATE = FUSEP && (!PLAIN || (!FASTF && sizeof(T) == 4 && !(LEAN == 0 && NJ >= 12)));
    double zp[FUSEP ? NJ : 1];
    // ZLATE (float64 I/O, GLATE): the task's direction is formed where it is used, behind the IK's solve -- 2 n registers less through
    // field and IK (152 -> 56, 128 -> 32 B of scratch; the float variants got WORSE with it, 76 -> 128, and keep the early form)
    constexpr bool ZLATE = GLATE && sizeof(T) == 8;
    if constexpr (FUSEP && !ZLATE) {
        const bool jlt = a.flags & VFIK_F_JOINT_LIMIT_TASK;
#pragma unroll
        for (int i = 0; i < NJ; ++i) zp[i] = 0.0;
        if (jlt) jl_descent(zp);
    }
    // ACCJ (long chains, unit weights): J J^T and J z are accumulated here, while each Jacobian column is still in
    // VGPRs on its way into the AGPRs the Jacobian lives in at these sizes -- one pass of 168 register moves less.
    constexpr bool ACCJ = NJ >= 8 && PLAIN;
    double Ae[ACCJ ? 6 : 1][6], wne[6];
    if constexpr (ACCJ) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            wne[r] = 0.0;
#pragma unroll
            for (int c = 0; c <= r; ++c) Ae[r][c] = 0.0;
        }
    }
    // Columns of the Jacobian that are structurally sparse under the DH pattern (jzero(i, r): entry r of column i is EXACTLY zero):
    //   * an identity base puts joint 1's axis on the base z axis through the origin: column (-p_y, p_x, 0, 0, 0, 1);
    //   * a last link with a = 0 and alpha = 0 leaves the flange ON the last joint's axis: column (0, 0, 0, z) -- computed, its linear part
    //     is rounding noise of size 1e-17.
    // Every accumulation over the columns below (J J^T, J z, J^T y) skips the zero entries: 2 x 27 operations for the LWR.
    constexpr bool J0_UNIT = BASE_I;
    constexpr bool JL_AXIAL = PLAIN && ((DhPattern<NJ, DHPAT>::NONE >> (NJ - 1)) & 1u);
    auto jzero = [&](int i, int r) { return (J0_UNIT && i == 0 && (r == 2 || r == 3 || r == 4)) || (JL_AXIAL && i == NJ - 1 && r < 3); };
    // geometric Jacobian at the flange, base frame
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        {
            double cx, cy, cz;
            if (J0_UNIT && i == 0) {
                cx = -p[1]; cy = p[0]; cz = 0.0;
                Jm[i][3] = 0.0; Jm[i][4] = 0.0; Jm[i][5] = 1.0;
            } else if (JL_AXIAL && i == NJ - 1) {
                cx = 0.0; cy = 0.0; cz = 0.0;
            } else {
                const double dx = p[0] - Jm[i][0], dy = p[1] - Jm[i][1], dz = p[2] - Jm[i][2];
                cx = Jm[i][4] * dz - Jm[i][5] * dy; cy = Jm[i][5] * dx - Jm[i][3] * dz;
                cz = Jm[i][3] * dy - Jm[i][4] * dx;
            }
            if constexpr (ACCJ) {
                const double col[6] = {cx, cy, cz, Jm[i][3], Jm[i][4], Jm[i][5]};
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    if (jzero(i, r)) continue;
#pragma unroll
                    for (int c = 0; c <= r; ++c)
                        if (!jzero(i, c)) Ae[ACCJ ? r : 0][c] = __builtin_fma(col[r], col[c], Ae[ACCJ ? r : 0][c]);
                    if constexpr (FUSEP && !GLATE) wne[r] = __builtin_fma(col[r], zp[FUSEP ? i : 0], wne[r]);
                }
            }
            if (PLAIN) {
                Jm[i][0] = cx; Jm[i][1] = cy; Jm[i][2] = cz;
            } else {
                const bool pris = (kc->prismatic_mask >> i) & 1u;
                Jm[i][0] = pris ? Jm[i][3] : cx; Jm[i][1] = pris ? Jm[i][4] : cy; Jm[i][2] = pris ? Jm[i][5] : cz;
                Jm[i][3] = pris ? 0.0 : Jm[i][3]; Jm[i][4] = pris ? 0.0 : Jm[i][4]; Jm[i][5] = pris ? 0.0 : Jm[i][5];
            }
        }
    }

    PIN_ARR(R, 9); PIN_ARR(p, 3);
#pragma unroll
    for (int i = 0; i < NJ; ++i) { PIN_ARR(Jm[i], 6); }
    STAMP(3);
    // ---------------- A4: tool offset (vf:321-332) --------------------------------------------
    double Rt[9], pt[3], rr[3];
    // (TOOLC: the batch's shared tool on the PLAIN kernel.  The flange frame dies here either way: the point shift of the twist (below)
    // and /pose_no_tool (epilogue) are formed from the tool pose and the tool's constants.)
    if (PLAIN) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rt[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { pt[k] = p[k]; rr[k] = 0.0; }
        if constexpr (TOOLC) {
            double tl[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) tl[k] = HOTK(tool[k]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double x = Rt[3 * r], y = Rt[3 * r + 1], z = Rt[3 * r + 2];
#pragma unroll
                for (int c = 0; c < 3; ++c) Rt[3 * r + c] = x * tl[c] + y * tl[4 + c] + z * tl[8 + c];
                pt[r] += x * tl[3] + y * tl[7] + z * tl[11];
            }
        }
    } else {
        double tl[12];
        if (a.tool_stride) {
#pragma unroll
            for (int k = 0; k < 3; ++k) read_quad<T>(region, Stage<T>::tool_off(NJ) + k * Stage<T>::QSTEP, lanec, tl + 4 * k);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) tl[k] = kc->tool[k];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[3 * r] * tl[c] + R[3 * r + 1] * tl[4 + c] + R[3 * r + 2] * tl[8 + c];
            rr[r] = -(R[3 * r] * tl[3] + R[3 * r + 1] * tl[7] + R[3 * r + 2] * tl[11]);  // p_ee - p_tip
            pt[r] = p[r] - rr[r];
        }
    }

    // ---------------- A7 (first half): the IK's normal matrix and its factorisation -------------
    // A = Jw Jw^T + lambda^2 I depends on q alone, not on the field: for chains of up to 7 joints it is
    // factorised BEFORE the field is evaluated (registers permitting), which frees the Jacobian-sized
    // working set early and leaves only the two triangular solves after the field.  Measured neutral on
    // C3 (7.05-7.14 us per launch either way): the pivot chain is not what the wave waits for.
    constexpr bool EARLY_FACTOR = NJ <= 7;
    // (FUSEP: G = J J^T and J z are accumulated with A, and J^T w is subtracted while J^T y is formed -- for n = 14
    // the Jacobian lives in AGPRs and every further pass costs 168 register moves.)
    const double* wts = (!PLAIN && a.wts) ? a.wts + arm : nullptr;
    const long wpitch = a.Bpad;
    double A[6][6], dinv[6];
    double G[FUSEP ? 6 : 1][6], wn[6];  // FUSEP: undamped Gram matrix and J z of the projector
    // Weighted form (vf:295-309: Wy = diag of the 't' weights, Wq of the 'j' weights), without a weighted copy of the
    // Jacobian: A = Wy (J Wq^2 J^T) Wy + lambda^2 I and qdot = Wq^2 J^T (Wy y).  (Until round 3 the general variants kept
    // Sw = Wy J Wq beside J: 12 n more registers, which the 10- and 14-joint kernels and the rollouts spilled to scratch.)
    double wyv[WEIGHTED ? 6 : 1];
    auto wq2_of = [&](int i) {  // wq_i^2 of this arm
        const double wqi = wts ? wts[(long)(6 + i) * wpitch] : kc->wq[i];
        return wqi * wqi;
    };
    auto ik_factor = [&]() {
        if constexpr (WEIGHTED) {
#pragma unroll
            for (int r = 0; r < 6; ++r) wyv[WEIGHTED ? r : 0] = wts ? wts[(long)r * wpitch] : kc->wy[r];
        }
        constexpr bool GFROMA = FUSEP && PLAIN && !GLATE;  // unit weights: G is A before the damping is added
        if constexpr (ACCJ) {  // accumulated with the Jacobian columns above
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                if constexpr (!GLATE) wn[r] = wne[r];
#pragma unroll
                for (int c = 0; c <= r; ++c) A[r][c] = Ae[ACCJ ? r : 0][c];
                if constexpr (!GFROMA) A[r][r] += HOTK(lambda2);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) A[r][c] = (r == c && !GFROMA && !WEIGHTED) ? HOTK(lambda2) : 0.0;
            if constexpr (FUSEP && !GLATE) {
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    wn[r] = 0.0;
#pragma unroll
                    for (int c = 0; c <= r; ++c) G[r][c] = 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < NJ; ++i) {  // joint by joint: 21 independent accumulators per step
                double t[6];
                if constexpr (WEIGHTED) {
                    const double w2 = wq2_of(i);
#pragma unroll
                    for (int r = 0; r < 6; ++r) t[r] = w2 * Jm[i][r];
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    if (jzero(i, r)) continue;
#pragma unroll
                    for (int c = 0; c <= r; ++c)
                        if (!jzero(i, c)) A[r][c] = __builtin_fma(WEIGHTED ? t[r] : Jm[i][r], Jm[i][c], A[r][c]);
                    if constexpr (FUSEP && !GLATE) {
                        wn[r] = __builtin_fma(Jm[i][r], zp[i], wn[r]);
                        if constexpr (!PLAIN) {
#pragma unroll
                            for (int c = 0; c <= r; ++c) G[FUSEP ? r : 0][c] = __builtin_fma(Jm[i][r], Jm[i][c], G[FUSEP ? r : 0][c]);
                        }
                    }
                }
            }
            if constexpr (WEIGHTED) {  // A <- Wy A Wy + lambda^2 I
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c) A[r][c] = __builtin_fma(wyv[WEIGHTED ? r : 0] * wyv[WEIGHTED ? c : 0], A[r][c], r == c ? HOTK(lambda2) : 0.0);
            }
        }
        if constexpr (GFROMA) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c <= r; ++c) G[FUSEP ? r : 0][c] = A[r][c];
                A[r][r] += HOTK(lambda2);
            }
        }
        // LDL^T (unit lower L stored in A's strict lower part, d on the diagonal)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double v[6];  // v_k = L_jk d_k
#pragma unroll
            for (int k = 0; k < j; ++k) v[k] = A[j][k] * A[k][k];
            double dj = A[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) dj = __builtin_fma(-A[j][k], v[k], dj);
            A[j][j] = dj;
            dinv[j] = rcp_1nr(dj);
#pragma unroll
            for (int k = 0; k < j; ++k)  // rows below j, one k at a time: the rows are independent chains
#pragma unroll
                for (int i = j + 1; i < 6; ++i) A[i][j] = __builtin_fma(-A[i][k], v[k], A[i][j]);
#pragma unroll
            for (int i = j + 1; i < 6; ++i) A[i][j] *= dinv[j];
        }
    };
    if constexpr (EARLY_FACTOR) ik_factor();

    // ---------------- A5: vector field at the tool pose (vf:276-293,344-347) -------------------
    double tot[6] = {0, 0, 0, 0, 0, 0}, sc[2] = {1.0, 1.0};
    double gdist[2] = {0.0, 0.0};  // distance and rotation angle to the goal (monitor_distance:161-167)
    double speed;  // this arm's speedScale (vf:134-137,197-207), 4th component of the goal block's last quad
    // (Round 3 tried pinning the factorisation in front of this wait by tying the wait to its results: neutral in the cold
    // state, noisy-to-slower in the warm one; the compiler's own placement stays.)
    if (PERS && has_next) VFIK_WAIT_VM(N_SLOT + NPF);
    else
    VFIK_WAIT_VM(N_SLOT);  // goal block has landed
    {
        double gq[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) read_quad<T>(dreg, Stage<T>::GOAL_OFF + k * Stage<T>::QSTEP, lanec, gq + 4 * k);
        speed = gq[15];
        {   // goal block = the arm's lowest-id attractor: [frame rows 0..2 | present, slow, force, speedScale]
            double GR[9], Gp[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) GR[3 * r + c] = gq[4 * r + c];
                Gp[r] = gq[4 * r + 3];
            }
            // a.goal_dist requested: the angle is needed whatever its size (cos_slow = -2 forces atan2)
            attractor(Rt, pt, GR, Gp, gq[13], gq[14], HOTK(rot_slow), a.goal_dist ? -2.0 : HOTK(cos_slow), gq[12] != 0.0, tot, sc, gdist);
        }
    }
    if constexpr (FUN) {
        // type 5, funnel attractor (object_feeder:262-279), straight-line from its own block -- the arithmetic of eval_slot's funnel
        // branch, an absent funnel masked by selects: w = p - o, perp = w - (w . a^) a^, phi = atan2(|perp|, w . a^),
        // vector -perp^ min(1, (phi / cutAngle)^angleOrder) min(1, (cutDist / |w|)^distOrder)
        double f[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) read_quad<T>(region, FUN_OFF + k * Stage<T>::QSTEP, lanec, f + 4 * k);
        double an, ainv;
        sqrt_rsqrt(f[3] * f[3] + f[4] * f[4] + f[5] * f[5], an, ainv);
        const bool on = f[11] != 0.0 && an > EPS_LEN;
        const double ax = f[3] * ainv, ay = f[4] * ainv, az = f[5] * ainv;
        const double wx = pt[0] - f[0], wy = pt[1] - f[1], wz = pt[2] - f[2];
        const double along = wx * ax + wy * ay + wz * az;
        const double ex = wx - along * ax, ey = wy - along * ay, ez = wz - along * az;
        double P, Pinv, dist, dinv;
        sqrt_rsqrt(ex * ex + ey * ey + ez * ez, P, Pinv);
        sqrt_rsqrt(wx * wx + wy * wy + wz * wz, dist, dinv);
        Pinv = P < D_FLOOR ? 1.0 / D_FLOOR : Pinv;
        dinv = dist < D_FLOOR ? 1.0 / D_FLOOR : dinv;
        const double phi = atan2_pos(P, on ? along : 1.0);
        const double cutA = f[6];
        const double ga = cutA > 0.0 ? fmin(1.0, pow_order(phi * rcp_nr(cutA > 0.0 ? cutA : 1.0), on ? f[7] : 1.0)) : 1.0;
        const double gd = fmin(1.0, pow_order(f[8] * dinv, on ? f[9] : 1.0));
        const double kf = on ? -f[10] * ga * gd * Pinv : 0.0;  // (a select: an absent funnel's block holds zeros, its terms may be NaN)
        tot[0] += on ? ex * kf : 0.0; tot[1] += on ? ey * kf : 0.0; tot[2] += on ? ez * kf : 0.0;
        // type 4, hemisphere repeller (object_feeder:344-353: a surface with its normal), from the block behind the funnel's:
        // h = (p - o) . n^, vector -n^ min((safe / max(h, floor))^order, cap) -- eval_slot's hemisphere branch, masked likewise
        double g[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) read_quad<T>(region, FUN_OFF + (3 + k) * Stage<T>::QSTEP, lanec, g + 4 * k);
        if (__any(g[9] != 0.0)) {  // (a wave without a table skips the chain: wave-uniform)
            double nn, ninv;
            sqrt_rsqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5], nn, ninv);
            const bool hon = g[9] != 0.0 && nn > EPS_LEN;
            const double hh = ((pt[0] - g[0]) * g[3] + (pt[1] - g[1]) * g[4] + (pt[2] - g[2]) * g[5]) * ninv;
            const double hmag = fmin(pow_order(g[6] * rcp_nr(fmax(hon ? hh : 1.0, D_FLOOR)), hon ? g[7] : 1.0), MAG_CAP);
            const double kh = hon ? -g[8] * hmag * ninv : 0.0;
            tot[0] += hon ? g[3] * kh : 0.0; tot[1] += hon ? g[4] * kh : 0.0; tot[2] += hon ? g[5] * kh : 0.0;
        }
    }
    PIN_ARR(tot, 6); PIN_ARR(sc, 2);
    STAMP(4);
    {
        const T* sq = static_cast<const T*>(a.slots) + (long)arm * 4;
        const long Qp = Bp * 4;
        if constexpr (FASTF) {  // (the host launches this variant when a.fast_order >= 0)
            // Fast path, decided by the host when the field sets were packed (vfik_set_fields): every
            // used slot of every arm is a decay repeller with the same integer decay order -- what
            // object_feeder produces for point obstacles (object_feeder:317-334).  Empty slots carry
            // force 0.  Straight-line code per chunk of PRE slots: the slots interleave in the schedule.
            const int n0 = a.fast_order;
            // One chunk = PRE slots: read them out of LDS, then -- before the arithmetic -- request the
            // next chunk into the same rows, so that its latency hides behind this chunk's math.
            // One chunk = PRE slots: read them out of LDS, then -- before the arithmetic -- request the
            // next chunk into the same rows, so that its latency hides behind this chunk's math.
            // (Round 3 tried the first chunk in two halves, each behind its own counted wait, so that the second half's
            // arrival would hide behind the first half's arithmetic when the inputs come from HBM: +2.4 % cold, +1 % warm
            // -- the eight slots in lock step are worth more than the overlap; profiles/r03_ab_experiments.md.)
            auto chunk = [&](int c0) {
                const int ncur = a.slots_used - c0;  // slots of this chunk that are in use (may exceed PRE)
                double dx[PRE], dy[PRE], dz[PRE], rs[PRE], fk[PRE];
                unsigned ow[MIXO ? PRE / 4 : 1];     // MIXO: this chunk's decay orders, one byte a slot
                if constexpr (MIXO) {
                    const char* orow = region + ORD_OFF + lanec * 16 + (c0 & 15);
#pragma unroll
                    for (int u = 0; u < PRE / 4; ++u) ow[u] = *reinterpret_cast<const unsigned*>(orow + 4 * u);
                }
                if constexpr (UNI) {  // a slot = one quad (x y z radius | radius = -inf: unused); safe distance and force are the batch's (KConst)
                    // An unused slot needs no test: max(-inf + safe, 0) = 0 makes its magnitude 0 (the host admits the uniform image only
                    // when radius + safe >= 0 for every repeller), and quads past the slots in use were requested from the image's empty
                    // plane -- 2 operations a slot where the masked form had 6 (round 4: C3 -2 %).
                    const double usafe = kl->rep_safe, uforce = kl->dh[0].pad;
#pragma unroll
                    for (int m = 0; m < PRE; ++m) {
                        double v[4];
                        read_quad<T>(dreg, Stage<T>::slot_off(m, NJ), lanec, v);
                        dx[m] = v[0] - pt[0];
                        dy[m] = v[1] - pt[1];
                        dz[m] = v[2] - pt[2];
                        rs[m] = fmax(v[3] + usafe, 0.0);
                        fk[m] = uforce;
                    }
                } else
#pragma unroll
                for (int k = 0; k < PRE / 2; ++k) {  // a pair of slots = three quads: (x0 y0 z0 r0 | s0 f0 x1 y1 | z1 r1 s1 f1)
                    double v[12];
#pragma unroll
                    for (int u = 0; u < 3; ++u) read_quad<T>(dreg, Stage<T>::slot_off(3 * k + u, NJ), lanec, v + 4 * u);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int m = 2 * k + hf;
                        dx[m] = v[6 * hf] - pt[0];
                        dy[m] = v[6 * hf + 1] - pt[1];
                        dz[m] = v[6 * hf + 2] - pt[2];
                        rs[m] = v[6 * hf + 3] + v[6 * hf + 4];
                        fk[m] = m < ncur ? v[6 * hf + 5] : 0.0;
                    }
                }
                if (c0 + PRE < a.slots_used) {  // wave-uniform
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the rows have returned
#pragma unroll
                    for (int idx = 0; idx < QPC; ++idx) {
                        const int m = UNI ? c0 + PRE + idx : c0 + PRE + 2 * (idx / 3);  // (first) slot of the quad
                        const char* sm = slots0 + (m < a.slots_used ? (long)(UNI ? c0 + PRE + idx + 1 : (c0 + PRE) / 2 * 3 + idx) * planeB : 0);
                        stage_quad<T, NTL>(sm, (unsigned)arm * (unsigned)QB, dreg, Stage<T>::slot_off(idx, NJ));
                    }
                    if constexpr (MIXO) issue_orders(c0 + PRE);   // (the row's bytes of this chunk are in registers)
                }
                double di[PRE], rb[PRE], rp[PRE];
#pragma unroll
                for (int m = 0; m < PRE; ++m) {
                    // (the sum starts from 1e-300 instead of being clamped there afterwards: one operation less a slot, the same guard
                    // against 0 * inf when the tool sits exactly on an obstacle's centre)
                    di[m] = fmin(rsqrt_1nr_pos(__builtin_fma(dx[m], dx[m], __builtin_fma(dy[m], dy[m], __builtin_fma(dz[m], dz[m], 1e-300)))), 1.0 / D_FLOOR);
                    rb[m] = rs[m] * di[m];
                    rp[m] = 1.0;
                }
                if constexpr (MIXO) {
                    // do the wave's 64 arms agree on every order of the chunk?  (the words compared whole: two or one)
                    unsigned su[PRE / 4];
                    bool agree = true;
#pragma unroll
                    for (int u = 0; u < PRE / 4; ++u) {
                        su[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)ow[MIXO ? u : 0]);
                        agree = agree && __all(ow[MIXO ? u : 0] == su[u]);
                    }
                    // Wave-uniform orders, at most TWO distinct ones in the chunk (the README's scene: order-20 obstacles beside the feeder's
                    // order-5 near-goal repeller): one pass per distinct order under scalar control flow -- rb^o for all slots in lock step,
                    // then the slots with that order take their value by a select on a scalar condition.  Three or more distinct orders,
                    // or an arm with an order of its own: per-lane selects over the bits of the largest order (a pass costs ~0.3 us of a
                    // 5.3-us launch, the bit-serial form ~0.9 us whatever the number of orders: profiles/r04_ab_experiments.md; a first
                    // version that tested every (slot, bit) pair with a scalar branch cost 1.0 us for two orders).
                    bool lanes = !agree;
                    if (agree) {
                        bool all5 = true;
#pragma unroll
                        for (int u = 0; u < PRE / 4; ++u) all5 = all5 && su[u] == 0x05050505u;
                        auto order_of = [&](int m) { return ((m < 4 ? su[0] : su[PRE / 4 - 1]) >> (8 * (m & 3))) & 127u; };
                        auto fifth = [&](double* dst) {
#pragma unroll
                            for (int m = 0; m < PRE; ++m) { const double b2 = rb[m] * rb[m]; dst[m] = b2 * b2 * rb[m]; }
                        };
                        const unsigned o1 = order_of(0);
                        unsigned todo = 0, rest = 0, o2 = o1;
                        if (!all5) {
#pragma unroll
                            for (int m = 1; m < PRE; ++m) todo |= (order_of(m) != o1 ? 1u : 0u) << m;
                            if (todo) {
                                o2 = order_of(__builtin_ctz(todo));
#pragma unroll
                                for (int m = 1; m < PRE; ++m) rest |= ((todo >> m) & 1u && order_of(m) != o2 ? 1u : 0u) << m;
                            }
                        }
                        lanes = rest != 0;
                        if (all5) {
                            fifth(rp);
                        } else if (!lanes) {
                            auto raise_all = [&](unsigned o, double* dst) {   // left-to-right binary: the control flow depends on o alone
                                if (o <= 1) {
#pragma unroll
                                    for (int m = 0; m < PRE; ++m) dst[m] = o ? rb[m] : 1.0;
                                    return;
                                }
                                int k = 30 - __builtin_clz(o);        // the bit below the leading one (o >= 2: k >= 0)
#pragma unroll
                                for (int m = 0; m < PRE; ++m) dst[m] = rb[m] * rb[m];
                                for (;;) {
                                    if ((o >> k) & 1u) {
#pragma unroll
                                        for (int m = 0; m < PRE; ++m) dst[m] *= rb[m];
                                    }
                                    if (--k < 0) break;
#pragma unroll
                                    for (int m = 0; m < PRE; ++m) dst[m] *= dst[m];
                                }
                            };
                            if (o1 == 5) fifth(rp);
                            else raise_all(o1, rp);          // the first order: every slot takes it, no select
                            if (todo) {
                                double pw[PRE];
                                // an order that is 2^j times the first (20 after 5): j squarings of the first pass's powers
                                const unsigned ratio = o1 ? o2 / o1 : 0u;
                                if (ratio > 1 && ratio * o1 == o2 && (ratio & (ratio - 1)) == 0) {
#pragma unroll
                                    for (int m = 0; m < PRE; ++m) pw[m] = rp[m] * rp[m];
                                    for (unsigned r = ratio >> 1; r > 1; r >>= 1) {
#pragma unroll
                                        for (int m = 0; m < PRE; ++m) pw[m] *= pw[m];
                                    }
                                } else {
                                    raise_all(o2, pw);
                                }
#pragma unroll
                                for (int m = 1; m < PRE; ++m) rp[m] = (todo >> m) & 1u ? pw[m] : rp[m];
                            }
                        }
                    }
                    if (lanes) {
                        int nn[PRE], nmax = 0;
#pragma unroll
                        for (int m = 0; m < PRE; ++m) { nn[m] = (int)((ow[MIXO ? (m >> 2) : 0] >> (8 * (m & 3))) & 127u); nmax |= nn[m]; }
                        int top = 0;
#pragma unroll
                        for (int k = 0; k < 7; ++k)
                            if (__any((nmax >> k) != 0)) top = k + 1;
                        for (int k = 0; k < top; ++k) {
#pragma unroll
                            for (int m = 0; m < PRE; ++m) rp[m] = (nn[m] >> k) & 1 ? rp[m] * rb[m] : rp[m];
                            if (k + 1 < top) {
#pragma unroll
                                for (int m = 0; m < PRE; ++m) rb[m] *= rb[m];
                            }
                        }
                    }
                } else
                if (n0 == 5) {  // what object_feeder sends (object_feeder:302,333): no loop, no branches
#pragma unroll
                    for (int m = 0; m < PRE; ++m) { const double b2 = rb[m] * rb[m]; rp[m] = b2 * b2 * rb[m]; }
                } else {
                for (int e = n0; e;) {  // square-and-multiply, all slots in lock step
                    if (e & 1) {
#pragma unroll
                        for (int m = 0; m < PRE; ++m) rp[m] *= rb[m];
                    }
                    e >>= 1;
                    if (e) {
#pragma unroll
                        for (int m = 0; m < PRE; ++m) rb[m] *= rb[m];
                    }
                }
                }
#pragma unroll
                for (int m = 0; m < PRE; ++m) {
                    const double k = fk[m] * fmin(rp[m], MAG_CAP) * di[m];
                    tot[0] += dx[m] * k; tot[1] += dy[m] * k; tot[2] += dz[m] * k;
                }
            };
            // the first chunk (requested during the kinematics, or -- PERS -- at the start of the launch / under the previous
            // chunk of arms) has landed; PERS: the requests of the wave's NEXT chunk of arms are younger and stay out
            if (PERS && has_next) VFIK_WAIT_VM(NPF);
            else VFIK_WAIT_VM(0);
            chunk(0);
            // further chunks: kept out of line of the first so that the common (<= PRE slots) case is
            // straight-line code with no loop-carried register shuffling
            for (int c0 = PRE; c0 < a.slots_used; c0 += PRE) {
                VFIK_WAIT_VM(0);
                chunk(c0);
            }
        } else {
            // General path (mixed primitive types, fractional or differing orders): slot by slot.  The first
            // PRE slots are read from the rows staged in LDS during the kinematics -- read from memory, every
            // slot cost a round trip of its own (C3 with one odd arm: 10.9 us per launch) -- the rest, and
            // entries that would straddle the staged window, from the quad planes.
            const SlotLds<T> rl{dreg, lanec, NJ};
            const double rs = HOTK(rot_slow), csl = HOTK(cos_slow);
            for (int c0 = 0; c0 < a.slots_used; c0 += PRE) {  // chunks of PRE slots through the staged rows
                if (c0 > 0) {  // (no overlap with the previous chunk's arithmetic here: its entries read the rows lazily)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int idx = 0; idx < 2 * PRE; ++idx) {
                        const int m = c0 + (idx >> 1);
                        const char* sm = slots0 + (m < a.slots_used ? (long)m * 2 * planeB : 0) + (idx & 1) * planeB;
                        stage_quad<T, NTL>(sm, (unsigned)arm * (unsigned)QB, dreg, Stage<T>::slot_off(idx, NJ));
                    }
                }
                VFIK_WAIT_VM(0);
                const int ncur = a.slots_used - c0 < PRE ? a.slots_used - c0 : PRE;  // slots of this chunk in use
                const SlotGlobal<T> rg{sq + (long)c0 * 2 * Qp, Qp};  // the same slots in memory: local index m = slot c0 + m
                // Stage A: the decay repellers with an integer order among the staged slots, all together as in
                // the straight-line path (slot by slot the dependent chains of sqrt and power cost ~1 600 cycles a
                // slot); the order may differ from lane to lane and slot to slot: square-and-multiply over the
                // bits of the largest order in the wave, each lane selecting by its own bits.
                unsigned done = 0;  // bit m: slot m of the chunk was handled here (per lane)
                {
                    double dx[PRE], dy[PRE], dz[PRE], rsum[PRE], fk[PRE], di[PRE], rb[PRE], rp[PRE];
                    int nn[PRE];
                    int nmax = 0;
#pragma unroll
                    for (int m = 0; m < PRE; ++m) {
                        double s0[4], s1[4];
                        read_quad<T>(dreg, Stage<T>::slot_off(2 * m, NJ), lanec, s0);
                        read_quad<T>(dreg, Stage<T>::slot_off(2 * m + 1, NJ), lanec, s1);
                        const int n = (int)s1[1];
                        const bool ok = m < ncur && (int)s1[3] == VFIK_FIELD_REPELLER && (double)n == s1[1] && n >= 0 && n < 128;
                        dx[m] = s0[0] - pt[0]; dy[m] = s0[1] - pt[1]; dz[m] = s0[2] - pt[2];
                        rsum[m] = s0[3] + s1[0];
                        fk[m] = ok ? s1[2] : 0.0;
                        nn[m] = ok ? n : 0;
                        nmax |= nn[m];
                        done |= ok ? (1u << m) : 0u;
                    }
#pragma unroll
                    for (int m = 0; m < PRE; ++m) {
                        di[m] = fmin(rsqrt_1nr(dx[m] * dx[m] + dy[m] * dy[m] + dz[m] * dz[m]), 1.0 / D_FLOOR);
                        rb[m] = rsum[m] * di[m];
                        rp[m] = 1.0;
                    }
                    int top = 0;  // number of order bits in use anywhere in the wave (uniform)
#pragma unroll
                    for (int k = 0; k < 7; ++k)
                        if (__any((nmax >> k) != 0)) top = k + 1;
                    for (int k = 0; k < top; ++k) {
#pragma unroll
                        for (int m = 0; m < PRE; ++m) rp[m] = (nn[m] >> k) & 1 ? rp[m] * rb[m] : rp[m];
                        if (k + 1 < top) {
#pragma unroll
                            for (int m = 0; m < PRE; ++m) rb[m] *= rb[m];
                        }
                    }
#pragma unroll
                    for (int m = 0; m < PRE; ++m) {
                        const double k = (done >> m) & 1 ? fk[m] * fmin(rp[m], MAG_CAP) * di[m] : 0.0;  // a select: the unused slots may hold anything
                        tot[0] += dx[m] * k; tot[1] += dy[m] * k; tot[2] += dz[m] * k;
                    }
                }
                // Stage B: every other entry, one at a time: from the staged rows when the whole entry lies in
                // them, else (it runs into the next chunk) from the quad planes in memory.
                for (int m = 0; m < ncur; ++m) {
                    if (__all((done >> m) & 1)) continue;
                    const int t = (int)rl(m, 7);  // an entry spans 1 (repeller), 2 (hemisphere, funnel) or 3 (attractor) slots
                    const int span = t == VFIK_FIELD_ATTRACTOR ? 3 : (t == VFIK_FIELD_HEMISPHERE || t == VFIK_FIELD_FUNNEL) ? 2 : 1;
                    if ((done >> m) & 1) continue;
                    if (m + span <= ncur) eval_slot(rl, m, Rt, pt, rs, csl, tot, sc);
                    else eval_slot(rg, m, Rt, pt, rs, csl, tot, sc);
                }
            }
        }
    }
    PIN_ARR(tot, 6);
    STAMP(5);
    // The nullspace state is requested HERE, a phase ahead of the module that uses it (the IK's solve covers the round trip;
    // requested in front of the Gram-Schmidt block, as in round 2, the launch is 2.0 % slower with the inputs in HBM, 0.3 %
    // with them in the Infinity Cache: profiles/r03_ab_experiments.md).  A rollout loads it once, before its cycles.
    // (the two-waves-per-SIMD build has 256 registers a lane: it requests the state where the module starts -- its co-resident wave
    // covers the round trip -- instead of holding 8 more registers through the solve: 28 B of scratch per lane otherwise)
    if constexpr (NULLSP && NJ <= 7 && !ROLL && WAVES == 1) load_null_state();
    // normCart + speedScale * scalars (vf:292,346-347)
    double v[3], w[3];
    {
        double nt, nti, nr, nri;
        sqrt_rsqrt(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2], nt, nti);
        sqrt_rsqrt(tot[3] * tot[3] + tot[4] * tot[4] + tot[5] * tot[5], nr, nri);
        const double kt = nt > EPS_LEN ? speed * sc[0] * nti : 0.0;
        const double kr = nr > EPS_LEN ? speed * sc[1] * nri : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k] = tot[k] * kt; w[k] = tot[3 + k] * kr; }
    }

    // ---------------- A6: Twist.RefPoint(p_ee - p_tip) (vf:456-459) ----------------------------
    double tw[6];
    tw[0] = PLAIN ? v[0] : v[0] + (w[1] * rr[2] - w[2] * rr[1]);
    tw[1] = PLAIN ? v[1] : v[1] + (w[2] * rr[0] - w[0] * rr[2]);
    tw[2] = PLAIN ? v[2] : v[2] + (w[0] * rr[1] - w[1] * rr[0]);
    if constexpr (PLAIN) {
        if constexpr (TOOLC) {  // p_ee - p_tip = -R t = -Rt (Rtool^T t): from the tool pose, so that nothing of the flange frame lives through the field
            double c3[3], r3[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) c3[j] = HOTK(tool[j]) * HOTK(tool[3]) + HOTK(tool[4 + j]) * HOTK(tool[7]) + HOTK(tool[8 + j]) * HOTK(tool[11]);
#pragma unroll
            for (int r = 0; r < 3; ++r) r3[r] = -(Rt[3 * r] * c3[0] + Rt[3 * r + 1] * c3[1] + Rt[3 * r + 2] * c3[2]);
            tw[0] += w[1] * r3[2] - w[2] * r3[1];
            tw[1] += w[2] * r3[0] - w[0] * r3[2];
            tw[2] += w[0] * r3[1] - w[1] * r3[0];
        }
    }
    tw[3] = w[0]; tw[4] = w[1]; tw[5] = w[2];

    // ---------------- A7: weighted damped least squares (vf:461) -------------------------------
    double qv[NJ];
    {
        if constexpr (!EARLY_FACTOR) ik_factor();
        double y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double t = WEIGHTED ? wyv[WEIGHTED ? i : 0] * tw[i] : tw[i];
#pragma unroll
            for (int k = 0; k < i; ++k) t -= A[i][k] * y[k];
            y[i] = t;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) y[i] *= dinv[i];
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            double t = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) t -= A[k][i] * y[k];
            y[i] = t;
        }
        if constexpr (GLATE) {
            asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]));  // (behind the solve)
            if constexpr (ZLATE) {   // the joint-limit task's direction, from the joint angles in the wave's LDS rows
                const bool jlt = a.flags & VFIK_F_JOINT_LIMIT_TASK;
#pragma unroll
                for (int i = 0; i < NJ; ++i) zp[FUSEP ? i : 0] = 0.0;
                if (jlt) {
                    if constexpr (NJ >= 10 && !ROLL) read_q();
                    jl_descent(zp);
                }
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                wn[r] = 0.0;
#pragma unroll
                for (int c = 0; c <= r; ++c) G[FUSEP ? r : 0][c] = 0.0;
            }
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    wn[r] = __builtin_fma(Jm[i][r], zp[FUSEP ? i : 0], wn[r]);
#pragma unroll
                    for (int c = 0; c <= r; ++c) G[FUSEP ? r : 0][c] = __builtin_fma(Jm[i][r], Jm[i][c], G[FUSEP ? r : 0][c]);
                }
        }
        if constexpr (FUSEP) {
            // LDL^T of the undamped Gram matrix; a vanishing pivot (singular pose) drops that direction
            // instead of dividing by it.  Then wn <- (J J^T)^-1 J z.
            double (*const Gm)[6] = G;
            double gmax = Gm[0][0];
#pragma unroll
            for (int r = 1; r < 6; ++r) gmax = fmax(gmax, Gm[FUSEP ? r : 0][r]);
            double gi[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int jj = FUSEP ? j : 0;
                double v[6];
#pragma unroll
                for (int k = 0; k < j; ++k) v[k] = Gm[jj][k] * Gm[FUSEP ? k : 0][k];
                double dj = Gm[jj][j];
#pragma unroll
                for (int k = 0; k < j; ++k) dj = __builtin_fma(-Gm[jj][k], v[k], dj);
                const bool okp = dj > 1e-12 * gmax;
                Gm[jj][j] = okp ? dj : 0.0;
                gi[j] = okp ? rcp_nr(dj) : 0.0;
#pragma unroll
                for (int k = 0; k < j; ++k)
#pragma unroll
                    for (int i = j + 1; i < 6; ++i) Gm[FUSEP ? i : 0][j] = __builtin_fma(-Gm[FUSEP ? i : 0][k], v[k], Gm[FUSEP ? i : 0][j]);
#pragma unroll
                for (int i = j + 1; i < 6; ++i) Gm[FUSEP ? i : 0][j] *= gi[j];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int k = 0; k < i; ++k) wn[i] = __builtin_fma(-Gm[FUSEP ? i : 0][k], wn[k], wn[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) wn[i] *= gi[i];
#pragma unroll
            for (int i = 5; i >= 0; --i)
#pragma unroll
                for (int k = i + 1; k < 6; ++k) wn[i] = __builtin_fma(-Gm[FUSEP ? k : 0][i], wn[k], wn[i]);
        }
        if constexpr (WEIGHTED) {  // qdot = Wq^2 J^T (Wy y)
#pragma unroll
            for (int r = 0; r < 6; ++r) y[r] *= wyv[WEIGHTED ? r : 0];
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            qv[i] = jzero(i, 0) ? 0.0 : Jm[i][0] * y[0];
            if constexpr (FUSEP) { if (!jzero(i, 0)) zp[i] = __builtin_fma(-Jm[i][0], wn[0], zp[i]); }
        }
#pragma unroll
        for (int r = 1; r < 6; ++r)
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                if (jzero(i, r)) continue;
                qv[i] = __builtin_fma(Jm[i][r], y[r], qv[i]);
                if constexpr (FUSEP) zp[i] = __builtin_fma(-Jm[i][r], wn[r], zp[i]);
            }
        if (WEIGHTED) {
#pragma unroll
            for (int i = 0; i < NJ; ++i) qv[i] *= wq2_of(i);
        }
    }

    PIN_ARR(qv, NJ);
    STAMP(6);
    if constexpr (PERS) {
        // The next chunk's inputs have landed (requested a field evaluation and an IK ago): from here on the only
        // requests this wave leaves outstanding are its own stores, and the waits at the top of the next chunk pass.
        if (has_next) VFIK_WAIT_VM(0);
    }
    if constexpr (NJ >= 10 && !ROLL) {
        // Long chains: the joint angles are read from the wave's LDS rows AGAIN for the nullspace module, the mixer and the
        // outputs, instead of being held in 2 n registers through the field and the IK (the 14-joint kernels spill otherwise).
        asm volatile("" ::: "memory");
        read_q();
    }
    // ---------------- A10-A13: nullspace module (nullspace:95-131,162-184) ----------------------
    double qn[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) qn[i] = 0.0;
    if constexpr (NULLSP) {
        // Keep the compiler from starting this module before the IK has finished with Jm: interleaved,
        // the two keep two copies of the Jacobian alive and (n = 14) spill to scratch.  No instruction
        // is emitted: the empty asm only ties every Jm element to the IK result.
        if constexpr (NJ <= 7) {
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int r = 0; r < 6; ++r) asm volatile("" : "+v"(Jm[i][r]) : "v"(qv[0]));
            if constexpr (!ROLL && WAVES == 2) load_null_state();
            double c0 = 0.0;
            if (a.null_control) c0 = (double)static_cast<const T*>(a.null_control)[(long)arm * VFIK_NULL_CONTROLS];
            const bool advanced = nullspace_core<NJ, J0_UNIT, JL_AXIAL>(Jm, lv_r, sig_r, has_vec, c0, (a.flags & VFIK_F_JOINT_LIMIT_TASK) != 0, jl_descent, qn, status);
            if constexpr (!ROLL) {
                if (advanced && act) store_null_state();
            }
        } else {
            // n >= 8: the nullspace of a 6 x n Jacobian has dimension >= 2, so the reference's SVD basis is
            // never unique and /control is never honoured; only the projector is needed (joint-limit task):
            // (I - J^+ J) z = z - J^T (J J^T)^-1 J z, computed with the IK above (FUSEP).
            status |= VFIK_ST_NULL_AMBIGUOUS;
#pragma unroll
            for (int i = 0; i < NJ; ++i) qn[i] += zp[FUSEP ? i : 0];
        }
        // check_limits (nullspace:120-131) then gain (nullspace:183).  The limits are fetched in one go and
        // the test is bit arithmetic: written with `||` it became a chain of branches, each with its own
        // scalar load and wait (~200 cycles a joint).
        double lo[NJ], hi[NJ];
        limits_of(lo, hi);
        const double look = HOTK(lookahead), ngain = HOTK(null_gain);
        int bad = 0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const double d = q[i] + look * qn[i];
            bad |= (int)(d < lo[i]) | (int)(d > hi[i]);
        }
        const bool stop = bad != 0;
        if (stop) status |= VFIK_ST_LIMIT_STOP;
#pragma unroll
        for (int i = 0; i < NJ; ++i) qn[i] = stop ? 0.0 : qn[i] * ngain;
    }

    // ---------------- A15: command mixer (command_mixer.py:78-82) + limiter (bridge:188-195) ----
    double qo[NJ];
    bool direct = false;
    double mw[8];  // this arm's bridge state: mixer weights 0..5, limiter max_vel 6 (per-arm quads, else the batch's)
    if (a.flags & (VFIK_F_MIXER | VFIK_F_LIMITER)) {
        if (a.mixw) {
            read_quad<T>(region, Stage<T>::mixw_off(NJ), lanec, mw);
            read_quad<T>(region, Stage<T>::mixw_off(NJ) + Stage<T>::QSTEP, lanec, mw + 4);
        } else {
#pragma unroll
            for (int k = 0; k < VFIK_MIX_CHANNELS; ++k) mw[k] = HOTK(mix_w[k]);
            mw[6] = HOTK(max_vel);
        }
    }
    if (a.flags & VFIK_F_MIXER) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) qo[i] = mac_unfused(mul_unfused(qv[i], mw[0]), qn[i], mw[1]);  // (0.0 + x w is x w up to the sign of zero)
        if (a.q_ref) {  // joint P controller -> /bridge/jointcmd = channel 2 (joint_p_controller:78,89-99,124-128)
            const T* rf = static_cast<const T*>(a.q_ref) + (long)arm * NJ;
            double rlo[NJ], rhi[NJ];
            limits_of(rlo, rhi);
            const double delta = kc->jp_delta, kp = kc->jp_kp;
            // a NaN in the row's first element: this arm has no joint controller, its channel 2 is the external command
            const double rv0 = (double)rf[0];
            const bool ctl = rv0 == rv0;
            const T* e2 = a.ext ? static_cast<const T*>(a.ext) + (long)arm * NJ : nullptr;
            T* ro = (a.q_ref_out && act && (!ROLL || cyc == ncyc - 1)) ? static_cast<T*>(a.q_ref_out) + (long)arm * NJ : nullptr;
            int far = 0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const double rv = (double)rf[i];  // check_limits, joint_p_controller:89-99
                const double ref = rv < rlo[i] ? rlo[i] : (rv > rhi[i] ? rhi[i] : rv);
                const double err = ref - q[i];
                far |= (int)!(err < delta);  // signed, as joint_p_controller:135 compares it
                const double other = e2 ? (double)e2[i] : 0.0;
                qo[i] = mac_unfused(qo[i], ctl ? err * kp : other, mw[2]);
                if (ro) ro[i] = (T)(ctl ? ref : rv);  // the controller keeps the clamped reference (joint_p_controller:121)
            }
            if (!far && ctl) status |= VFIK_ST_JOINT_AT_GOAL;
        }
        if (a.ext) {
            const T* e = static_cast<const T*>(a.ext);
#pragma unroll
            for (int ch = 0; ch < VFIK_MIX_CHANNELS - 2; ++ch) {
                if (ch == 0 && a.q_ref) continue;
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    qo[i] = mac_unfused(qo[i], (double)e[((long)ch * Bs + arm) * NJ + i], mw[2 + ch]);
            }
        }
        direct = true;  // bridge:604: no controller has a weight -> the velocity goes out as it is
#pragma unroll
        for (int k = 0; k < VFIK_MIX_CHANNELS; ++k) direct = direct && mw[k] == 0.0;
    } else {
#pragma unroll
        for (int i = 0; i < NJ; ++i) qo[i] = qv[i];
    }
    if (a.flags & VFIK_F_LIMITER) {
        double lead = 0.0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) lead = fmax(lead, fabs(qo[i]));
        if (lead > mw[6]) {
            const double ratio = mw[6] * rcp_nr(lead);
#pragma unroll
            for (int i = 0; i < NJ; ++i) qo[i] *= ratio;
            status |= VFIK_ST_LIMITED;
        }
    }
    int nan = 0;
#pragma unroll
    for (int i = 0; i < NJ; ++i) nan |= (int)(qo[i] != qo[i]);
    if (nan) status |= VFIK_ST_NAN;

    if ((!ROLL || cyc == ncyc - 1) && act) {  // the outputs are those of the last evaluated cycle
        // ---------------- outputs (vf:341-342,462-466; nullspace:180-184; debug_jointlimits:69-73) --
        // Every output is batch-major ([B][K]: the reference publishes a bottle of K numbers per arm), so a lane's K values
        // are contiguous and the 64 lanes of a store instruction hit 64 different places K elements apart: 61 such
        // instructions for everything vf / nullspace / debug publish cost the C3 batch 3.5 us (7 700 cycles per wave,
        // tools/stamps.py C3F).  The wave's 64 rows are ONE contiguous tile of 64 K elements: it is assembled in LDS (the
        // goal block's rows, dead by now) and written with 16 bytes per lane, 1 KiB per instruction.  Not for LEAN launches of short chains
        // (seven stores in all), gated launches (a silent arm's row must stay), the batch's last partial wave, or output
        // pointers that are not 16-byte aligned: those store lane by lane.
        // (lean launches -- one row set -- take the tile for long chains only: 14 columns are 4 tile stores instead of 14 strided ones, C5 -2.1 %;
        // 7 columns are 2 instead of 7 and the LDS round trip costs more than they save, C3 +2.0 %, C3N +1.7 %: profiles/r04_ab_lean_tiles.txt)
        const bool tiles = (LEAN == 0 || LEAN == 3 || (LEAN == 1 && NJ >= 10)) && !a.active && (arm - lanec) + 64 <= a.B;
        auto put_rows = [&](void* out, auto kconst, auto&& val) {   // out[arm][i] = val(i), i < K
            constexpr int K = decltype(kconst)::value;
            T* const o = static_cast<T*>(out);
            if (tiles && (reinterpret_cast<unsigned long long>(out) & 15ull) == 0) {  // wave-uniform
                char* const tile = dreg + Stage<T>::GOAL_OFF;   // 64 x 16 elements at most: 4 QSTEP
                // A frame's row is 64 / 128 bytes: written element by element, the lanes of a store sit 16 / 32 banks apart and
                // land on two banks -- 16-way conflicts, 512 / 1 024 LDS cycles a tile for each of the CU's four waves (the
                // "1 100 cycles a 16-column tile" of round 3).  Such rows go in as 16-byte quads, the quad index XORed with
                // the row's number (bits 1-2 for 4 quads a row, bits 0-2 for 8) so that the 8 lanes of a store group cover
                // all 32 banks; the 16-byte reads below undo the permutation and stay conflict-free (it permutes within a
                // row).  Other widths (2, 6, 7, 10, 14 columns) are at most 2-way: element by element.
                constexpr int QPR = K * (int)sizeof(T) / 16;   // quads of a row
                constexpr bool SWZ = K == 16 && !ROLL;   // (a rollout publishes once per launch, and its variants have no register to spare: 12-228 B of scratch with the quads)
                if constexpr (SWZ) {
                    const int sw = sizeof(T) == 4 ? (lanec >> 1) & 3 : lanec & 7;
                    char* const row = tile + lanec * (K * (int)sizeof(T));
    #pragma unroll
                    for (int k = 0; k < QPR; ++k) {
                        if constexpr (sizeof(T) == 4) {
                            f4s x;
                            x.x = (float)val(4 * k); x.y = (float)val(4 * k + 1); x.z = (float)val(4 * k + 2); x.w = (float)val(4 * k + 3);
                            *reinterpret_cast<f4s*>(row + ((k ^ sw) * 16)) = x;
                        } else {
                            typedef double d2s __attribute__((ext_vector_type(2)));
                            d2s x;
                            x.x = (double)val(2 * k); x.y = (double)val(2 * k + 1);
                            *reinterpret_cast<d2s*>(row + ((k ^ sw) * 16)) = x;
                        }
                    }
                } else {
                    T* const mine = reinterpret_cast<T*>(tile) + lanec * K;
    #pragma unroll
                    for (int i = 0; i < K; ++i) mine[i] = (T)val(i);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's rows are in LDS (in-order LDS queue; nothing may move across)
                constexpr int PIECES = 64 * K * (int)sizeof(T) / 16;
                char* const g = reinterpret_cast<char*>(o + (long)(arm - lanec) * K);
    #pragma unroll
                for (int it = 0; it * 64 < PIECES; ++it) {
                    const int pc = it * 64 + lanec;
                    int src = pc;
                    if constexpr (SWZ) {
                        const int r = pc / QPR;
                        src = pc ^ (sizeof(T) == 4 ? (r >> 1) & 3 : r & 7);   // (the XOR touches the quad-in-row bits only)
                    }
                    if (PIECES % 64 == 0 || it * 64 + 64 <= PIECES || pc < PIECES)
                        *reinterpret_cast<f4s*>(g + (long)pc * 16) = *reinterpret_cast<const f4s*>(tile + src * 16);
                }
                asm volatile("" ::: "memory");
            } else {
    #pragma unroll
                for (int i = 0; i < K; ++i) {
#if VFIK_NT_STORES
                    __builtin_nontemporal_store((T)val(i), &o[(long)arm * K + i]);
#else
                    o[(long)arm * K + i] = (T)val(i);
#endif
                }
            }
        };
        typedef std::integral_constant<int, NJ> KNJ;
        // (every put_rows call sits in wave-uniform control flow: with `tiles` all 64 lanes assemble the tile together)
        if (a.qdot_out) {
            if (!ROLL && a.q_cmded) {  // LWR command form (bridge:199-203): -last_qcmded + last_q + qdot_lim, unless direct_control
                const T* qc = static_cast<const T*>(a.q_cmded) + (long)arm * NJ;
                double cmd[NJ];
    #pragma unroll
                for (int i = 0; i < NJ; ++i) cmd[i] = direct ? qo[i] : (-(double)qc[i] + q[i]) + qo[i];
                put_rows(a.qdot_out, KNJ(), [&](int i) { return cmd[i]; });
            } else {
                put_rows(a.qdot_out, KNJ(), [&](int i) { return qo[i]; });
            }
        }
        if (a.qdot_vf) put_rows(a.qdot_vf, KNJ(), [&](int i) { return qv[i]; });
        if (a.qdot_null) put_rows(a.qdot_null, KNJ(), [&](int i) { return qn[i]; });
        // (Round 3 tried pose / pose_no_tool / qdist right behind the read of the goal block, so that their writes would drain
        // under the field and the IK: C3F +-0 -- the wave pays the same ~1 100 cycles a 16-column tile wherever they sit --
        // and C5F 16.7 -> 18.2 us: the 14-joint kernel then spills.  They stay in the epilogue.)
        if (a.pose)
            put_rows(a.pose, std::integral_constant<int, 16>(), [&](int i) {
                return i < 12 ? ((i & 3) == 3 ? pt[i >> 2] : Rt[3 * (i >> 2) + (i & 3)]) : (i == 15 ? 1.0 : 0.0);
            });
        if (a.pose_nt) {
            if constexpr (!PLAIN && NJ >= 10) {
                // Long chains: the flange frame is recomposed from the tool pose, R = Rt Rtool^T and p = pt + (p_ee - p_tip),
                // instead of living in 12 double registers through the field and the IK beside Rt / pt (scratch otherwise).
                double tl[12];
                if (a.tool_stride) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) read_quad<T>(region, Stage<T>::tool_off(NJ) + k * Stage<T>::QSTEP, lanec, tl + 4 * k);
                } else {
#pragma unroll
                    for (int k = 0; k < 12; ++k) tl[k] = kc->tool[k];
                }
                double Rf[9];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) Rf[3 * r + c] = Rt[3 * r] * tl[4 * c] + Rt[3 * r + 1] * tl[4 * c + 1] + Rt[3 * r + 2] * tl[4 * c + 2];
                put_rows(a.pose_nt, std::integral_constant<int, 16>(), [&](int i) {
                    return i < 12 ? ((i & 3) == 3 ? pt[i >> 2] + rr[i >> 2] : Rf[3 * (i >> 2) + (i & 3)]) : (i == 15 ? 1.0 : 0.0);
                });
            } else if constexpr (PLAIN) {
                // (the tool pose IS the flange pose without a tool; with the batch's shared tool -- TOOLC -- the flange frame is recomposed,
                // R = Rt Rtool^T and p = pt - R t)
                double Rf[9], pf[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) Rf[k] = Rt[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) pf[k] = pt[k];
                if constexpr (TOOLC) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Rf[3 * r + c] = Rt[3 * r] * HOTK(tool[4 * c]) + Rt[3 * r + 1] * HOTK(tool[4 * c + 1]) + Rt[3 * r + 2] * HOTK(tool[4 * c + 2]);
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) pf[r] = pt[r] - (Rf[3 * r] * HOTK(tool[3]) + Rf[3 * r + 1] * HOTK(tool[7]) + Rf[3 * r + 2] * HOTK(tool[11]));
                }
                put_rows(a.pose_nt, std::integral_constant<int, 16>(), [&](int i) {
                    return i < 12 ? ((i & 3) == 3 ? pf[i >> 2] : Rf[3 * (i >> 2) + (i & 3)]) : (i == 15 ? 1.0 : 0.0);
                });
            } else {
                put_rows(a.pose_nt, std::integral_constant<int, 16>(), [&](int i) {
                    return i < 12 ? ((i & 3) == 3 ? p[i >> 2] : R[3 * (i >> 2) + (i & 3)]) : (i == 15 ? 1.0 : 0.0);
                });
            }
        }
        if (a.v6) put_rows(a.v6, std::integral_constant<int, 6>(), [&](int i) { return i < 3 ? v[i] : w[i - 3]; });
        if (a.qdist) {
            double dc[NJ];
            if (a.q_lo) {
                double lo[NJ], hi[NJ];
                limits_of(lo, hi);
    #pragma unroll
                for (int i = 0; i < NJ; ++i) dc[i] = fabs(q[i] - 0.5 * (lo[i] + hi[i])) * rcp_nr(0.5 * (hi[i] - lo[i]));
            } else {
    #pragma unroll
                for (int i = 0; i < NJ; ++i) dc[i] = fabs(q[i] - kc->q_mid[i]) * kc->inv_half[i];
            }
            put_rows(a.qdist, KNJ(), [&](int i) { return dc[i]; });
        }
        if (a.goal_dist)  // /dmonitor/distOut entry of object 0: xyz distance, rotation angle in DEGREES (monitor_distance:76-84,161-172)
            put_rows(a.goal_dist, std::integral_constant<int, 2>(), [&](int i) { return i == 0 ? gdist[0] : gdist[1] * 57.295779513082320877; });
    }
    if (ROLL || a.q_out) {  // joint_sim: integrate the commanded velocity; optionally stay inside the joint limits
#pragma unroll
        for (int i = 0; i < NJ; ++i) q[i] = __builtin_fma(a.dt, qo[i], q[i]);
        if (a.clamp) {  // one uniform branch, the limits fetched together (not a load and a wait per joint)
            double lo[NJ], hi[NJ];
            limits_of(lo, hi);
#pragma unroll
            for (int i = 0; i < NJ; ++i) q[i] = fmin(fmax(q[i], lo[i]), hi[i]);
        }
    }
    }  // cycles of this launch
    if (act) {  // (else no store has been made for this arm)
        if constexpr (NULLSP && ROLL && NJ <= 7) store_null_state();
        if (a.q_out) {
            T* o = static_cast<T*>(a.q_out) + (long)arm * NJ;
#pragma unroll
            for (int i = 0; i < NJ; ++i) o[i] = (T)q[i];
        }
        if (a.status) a.status[arm] = a.status_or ? (a.status[arm] | status) : status;
    }
    STAMP(7);
    if constexpr (!PERS) {
        break;
    } else {
        if (!has_next) break;
        chunk += (int)gridDim.x;
        const int an = chunk * 64 + (int)threadIdx.x;
        act = an < a.B;
        arm = arm_next;
        sg = slots0 + (long)arm * QB;
        dreg = dreg_next;
    }
    }  // chunks of this wave (PERS)
}

// The kernel proper: the body above behind an argument block (KArgs, or KLean for the lean single-cycle straight-line variants) ...
template <typename T, int NJ, bool NULLSP, bool PLAIN, bool ROLL, bool FASTF, int LEAN, int CF = -1, bool PERS = false, bool FUN = false, int WAVES = 1, bool UNI = false, bool MIXO = false, int DHP = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
cycle_kernel(const typename std::conditional<SmallArgs<LEAN, ROLL, FASTF, MIXO>::value, KLean, KArgs>::type a_in) {
    cycle_body<T, NJ, NULLSP, PLAIN, ROLL, FASTF, LEAN, CF, PERS, FUN, WAVES, UNI, MIXO, DHP>(a_in);
}
// ... or, for the KLean variants, behind KLean's ten members as SCALAR kernel arguments: those the command processor can preload into
// the wave's SGPRs at dispatch (-amdgpu-kernarg-preload-count, Makefile; an argument block passed by value is never preloaded), which
// takes the scalar-load round trip of the kernarg out of every wave's prologue.
template <typename T, int NJ, bool NULLSP, bool PLAIN, bool ROLL, bool FASTF, int LEAN, int CF = -1, bool PERS = false, bool FUN = false, int WAVES = 1, bool UNI = false, int DHP = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
cycle_kernel_s(const void* base, const void* q, void* qdot_out, int* status, int B, int Bpad, int slots_used, int fast_order, unsigned flags, int block) {
    static_assert(SmallArgs<LEAN, ROLL, FASTF>::value, "scalar arguments: the KLean variants");
    KLean k;
    k.base = base; k.q = q; k.qdot_out = qdot_out; k.status = status;
    k.B = B; k.Bpad = Bpad; k.slots_used = slots_used; k.fast_order = fast_order; k.flags = flags; k.block = block;
    cycle_body<T, NJ, NULLSP, PLAIN, ROLL, FASTF, LEAN, CF, PERS, FUN, WAVES, UNI, false, DHP>(k);
}
// The members of the handle's state arena and the launch's first arguments, from scalars (the arena's layout: vfik_kernel.h)
template <typename T, int NJ>
__device__ __forceinline__ void args_from_scalars(KArgs& a, const void* base, const void* q, void* qdot_out, int* status, int B, int Bpad, int slots_used,
                                                  unsigned flags) {
    a.q = q; a.qdot_out = qdot_out; a.status = status;
    a.B = B; a.Bpad = Bpad; a.slots_used = slots_used; a.flags = flags;
    const char* const b = static_cast<const char*>(base);
    a.goal = b;
    a.funnel = b + ArenaLayout<T, NJ>::funnel_off(Bpad);
    a.kc = b + ArenaLayout<T, NJ>::kconst_off(Bpad);
    a.lastvec = reinterpret_cast<float*>(const_cast<char*>(b) + ArenaLayout<T, NJ>::lastvec_off(Bpad));
    a.slots_fast = b + ArenaLayout<T, NJ>::slots_fast_off(Bpad);
}
// Every other variant: the same ten scalars IN FRONT of the argument block -- what the prologue needs to issue its first requests (the
// arena's members, q, the sizes) arrives preloaded; the rest of the block is read by scalar loads that run under those requests.
template <typename T, int NJ, bool NULLSP, bool PLAIN, bool ROLL, bool FASTF, int LEAN, int CF = -1, bool PERS = false, bool FUN = false, int WAVES = 1, bool UNI = false, bool MIXO = false, int DHP = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
cycle_kernel_x(const void* base, const void* q, void* qdot_out, const int* active, int B, int Bpad, int slots_used, int fast_order, unsigned flags, int block,
               const KArgs a_in) {
    static_assert(!SmallArgs<LEAN, ROLL, FASTF, MIXO>::value, "the KLean variants take cycle_kernel_s");
    KArgs a = a_in;
    args_from_scalars<T, NJ>(a, base, q, qdot_out, a_in.status, B, Bpad, slots_used, flags);
    a.active = active;   // (the fresh-q gate is the first request of the prologue; status is stored last and stays in the block)
    a.fast_order = fast_order; a.block = block;
    cycle_body<T, NJ, NULLSP, PLAIN, ROLL, FASTF, LEAN, CF, PERS, FUN, WAVES, UNI, MIXO, DHP>(a);
}

// The MIXO variants: the order planes' address is needed in the REQUEST phase, so it travels among the preloaded scalars too (read from
// the argument block it cost the prologue the scalar-load round trip that the preload exists to avoid); the launch is made with
// 64-thread blocks, which frees the two dwords of `fast_order` (unused: the orders are per slot) and `block`.
template <typename T, int NJ, bool NULLSP, int LEAN, bool FUN, int DHP = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
cycle_kernel_m(const void* base, const void* q, void* qdot_out, const int* active, const void* orders, int B, int Bpad, int slots_used, unsigned flags,
               const KArgs a_in) {
    KArgs a = a_in;
    args_from_scalars<T, NJ>(a, base, q, qdot_out, a_in.status, B, Bpad, slots_used, flags);
    a.active = active;
    a.orders = orders;
    a.fast_order = 0; a.block = 64;
    cycle_body<T, NJ, NULLSP, true, false, true, LEAN, -1, false, FUN, 1, false, true, DHP>(a);
}

// CommandMixer.read's weighted sum alone (command_mixer.py:78-82): out = sum_k cmd[k] * w[k], left to
// right from 0.0, multiply and add rounded separately (what CPython does).
template <typename T>
__global__ void __launch_bounds__(256) mix_kernel(const T* cmds, const double* w, int K, long count, long chan_stride, T* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc = mac_unfused(acc, (double)cmds[k * chan_stride + i], w[k]);
    out[i] = (T)acc;
}

// ------------------------------------------------------------------------------------------------
// Tracking-error estimator of scripts/vf (vf:349-428), batched, as its own small kernel (a diagnostic:
// it never feeds qdot).  Per arm it keeps the previous tool frame, the last 4 commanded twists and a
// frame counter; from the 6th frame on (frame_list longer than 5, vf:354) it compares the twist
// MEASURED between the two latest frames, scaled by the assumed 150 Hz robot loop (vf:408-411), with
// the twist COMMANDED check_delay = 4 cycles earlier (vf:363): direction angles, corrected magnitudes,
// |difference| and arm_tracking = difference < 0.10.  state: [38][B] doubles (12 frame, 24 commands,
// count, -).  out: [B][8] = vel_diff_angle, rot_diff_angle, ext_vel_mag_corr, ext_rot_mag_corr,
// cmd_vel_mag_corr, cmd_rot_mag_corr, ext_int_diff, arm_tracking (vf:418-427); zeros until the 6th frame.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) track_kernel(const T* pose, const T* v6, double* st, T* out, const int* active, int B) {
    const int arm = blockIdx.x * blockDim.x + threadIdx.x;
    if (arm >= B) return;
    if (active && !active[arm]) return;  // inside vf's `if qInBottle` (vf:312-313,349): no fresh q, no new frame
    const long Bs = B;
    double F[12], P[12], cmd[4][6];
#pragma unroll
    for (int k = 0; k < 12; ++k) { F[k] = (double)pose[(long)arm * 16 + k]; P[k] = st[k * Bs + arm]; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 6; ++k) cmd[j][k] = st[(12 + 6 * j + k) * Bs + arm];
    const double cnt = st[36 * Bs + arm];  // frames seen before this one
    // cmd_buffer.append(...); keep the last 4 (vf:350-352): after the shift cmd[0] is the oldest
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 6; ++k) cmd[j][k] = cmd[j + 1][k];
#pragma unroll
    for (int k = 0; k < 6; ++k) cmd[3][k] = (double)v6[(long)arm * 6 + k];
    double res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cnt >= 5.0) {  // len(frame_list) > frame_list_size (vf:354)
        // ext_diff = PyKDL.diff(previous frame, this frame): vel = dp, rot = log(R_b R_a^T) in the base frame
        double ev[3] = {F[3] - P[3], F[7] - P[7], F[11] - P[11]};
        double Ra[9], Rb[9], ax[3], th;
        bool has_axis;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { Ra[3 * r + c] = P[4 * r + c]; Rb[3 * r + c] = F[4 * r + c]; }
        rot_axis_angle(Ra, Rb, -2.0, ax, th, has_axis);
        const double ext_vel_mag = sqrt(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
        const double ext_rot_mag = has_axis ? th : 0.0;
        // the command issued check_delay = 4 cycles ago is the oldest entry of the full buffer (vf:363)
        const double* c = cmd[0];
        const double cmd_vel_mag = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        const double cmd_rot_mag = sqrt(c[3] * c[3] + c[4] * c[4] + c[5] * c[5]);
        double eu[3] = {1, 0, 0}, cu[3] = {1, 0, 0}, er[3] = {1, 0, 0}, cr[3] = {1, 0, 0};  // vf:365-374,390-399
        if (ext_vel_mag > 0.0) for (int k = 0; k < 3; ++k) eu[k] = ev[k] / ext_vel_mag;
        if (cmd_vel_mag > 0.0) for (int k = 0; k < 3; ++k) cu[k] = c[k] / cmd_vel_mag;
        if (ext_rot_mag > 0.0) for (int k = 0; k < 3; ++k) er[k] = ax[k];
        if (cmd_rot_mag > 0.0) for (int k = 0; k < 3; ++k) cr[k] = c[3 + k] / cmd_rot_mag;
        const double vd = fmin(1.0, fmax(-1.0, cu[0] * eu[0] + cu[1] * eu[1] + cu[2] * eu[2]));
        const double rd = fmin(1.0, fmax(-1.0, cr[0] * er[0] + cr[1] * er[1] + cr[2] * er[2]));
        res[0] = fabs(acos(vd));
        res[1] = fabs(acos(rd));
        res[2] = ext_vel_mag * 150.0;   // loop_freq (vf:408)
        res[3] = ext_rot_mag * 150.0;
        res[4] = cmd_vel_mag;
        res[5] = cmd_rot_mag / 5.0;     // vf:412
        res[6] = fabs((res[4] + res[5]) - (res[2] + res[3]));
        res[7] = res[6] < 0.10 ? 1.0 : 0.0;  // tracking_th (vf:409,416)
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out[(long)arm * 8 + k] = (T)res[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) st[k * Bs + arm] = F[k];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 6; ++k) st[(12 + 6 * j + k) * Bs + arm] = cmd[j][k];
    st[36 * Bs + arm] = cnt + 1.0;
}

// ------------------------------------------------------------------------------------------------
// Distance monitor of scripts/monitor_distance (monitor_distance:76-84,148-167), batched: for every arm
// and every object frame it was told about (/dmonitor/objectsIn: the goal is object 0, obstacles
// follow), the xyz distance between the tool pose and the object and the rotation angle between their
// orientations, in DEGREES (orientLength = |diff(current, final).rot| * 180/pi).  One thread per
// (arm, object); pose [B][16], frames [B][O][16], out [B][O][2].  A diagnostic: it never feeds qdot.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) monitor_kernel(const T* pose, const T* frames, int O, long count, T* out, const int* active) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const long arm = t / O;
    if (active && !active[arm]) return;  // no fresh q, no new pose (monitor_distance:138-147 acts on a pose that ARRIVED): the rows stay
    double P[12], F[12];
    const T* pp = pose + arm * 16;
    const T* fp = frames + t * 16;
#pragma unroll
    for (int k = 0; k < 12; ++k) { P[k] = (double)pp[k]; F[k] = (double)fp[k]; }
    const double dx = P[3] - F[3], dy = P[7] - F[7], dz = P[11] - F[11];
    // relative rotation E = R_cur^T R_obj; its angle is what |diff().rot| measures
    double E[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E[3 * i + j] = P[i] * F[j] + P[4 + i] * F[4 + j] + P[8 + i] * F[8 + j];
    const double a0 = 0.5 * (E[7] - E[5]), a1 = 0.5 * (E[2] - E[6]), a2 = 0.5 * (E[3] - E[1]);
    const double c = 0.5 * (E[0] + E[4] + E[8] - 1.0);
    const double sn = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
    out[2 * t] = (T)sqrt(dx * dx + dy * dy + dz * dz);
    out[2 * t + 1] = (T)(atan2_pos(sn, c) * 57.295779513082320877);
}

// ------------------------------------------------------------------------------------------------
// Field probe of scripts/vf (vf:469-503, "for visualizing"): the arm's total field evaluated at a pose that
// is handed in (/pose_in) instead of the forward kinematics: v6 = speedScale * scalars * normCart(sum)
// (vf:491-494 = vf:344-347 at another frame).  One thread per arm; goal block and slots are read straight
// from the quad planes through the general per-slot path.  pose [B][16], out [B][6].
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) probe_kernel(const T* pose, const T* goal, const T* slots, int B, long Bp, int slots_used,
                                                    double rot_slow, double cos_slow, T* out) {
    const int arm = blockIdx.x * blockDim.x + threadIdx.x;
    if (arm >= B) return;
    double Rt[9], pt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Rt[3 * r + c] = (double)pose[(long)arm * 16 + 4 * r + c];
        pt[r] = (double)pose[(long)arm * 16 + 4 * r + 3];
    }
    const long Qp = Bp * 4;
    double gq[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) gq[k] = (double)goal[(long)(k >> 2) * Qp + (long)arm * 4 + (k & 3)];
    double tot[6] = {0, 0, 0, 0, 0, 0}, sc[2] = {1.0, 1.0};
    double GR[9], Gp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) GR[3 * r + c] = gq[4 * r + c];
        Gp[r] = gq[4 * r + 3];
    }
    attractor(Rt, pt, GR, Gp, gq[13], gq[14], rot_slow, cos_slow, gq[12] != 0.0, tot, sc, nullptr);
    const T* sq = slots + (long)arm * 4;
    const SlotGlobal<T> rg{sq, Qp};
    for (int m = 0; m < slots_used; ++m) eval_slot(rg, m, Rt, pt, rot_slow, cos_slow, tot, sc);
    double nt, nti, nr, nri;
    sqrt_rsqrt(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2], nt, nti);
    sqrt_rsqrt(tot[3] * tot[3] + tot[4] * tot[4] + tot[5] * tot[5], nr, nri);
    const double speed = gq[15];
    const double kt = nt > EPS_LEN ? speed * sc[0] * nti : 0.0;
    const double kr = nr > EPS_LEN ? speed * sc[1] * nri : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out[(long)arm * 6 + k] = (T)(tot[k] * kt);
        out[(long)arm * 6 + 3 + k] = (T)(tot[3 + k] * kr);
    }
}

// ------------------------------------------------------------------------------------------------
// EIGHT LANES PER ARM (small batches): the mapping BASELINE.json's north_star sketches -- an arm spread over the lanes of
// a (sub-)wave with cross-lane exchange -- for what vfclik itself runs: a handful of arms (scripts/vfclik:88-105), each
// with its vf, nullspace and debug process and the bridge's mixer.  Served: revolute chain of up to 7 joints, identity
// tool, unit weights, goal + integer-order decay repellers; NS = the nullspace module (+ joint-limit task), mixer and
// limiter by the launch's flags, /control; outputs qdot_out, qdotOut, qdotout, pose, pose_no_tool, qdist, status -- what
// the per-arm processes publish every cycle (vf:341-342,462-466; nullspace:180-184; debug_jointlimits:69-73).
// A wave holds 8 arms; lane j of an arm's group of 8
//   * computes sin / cos of joint j                                   (1 angle per lane instead of 7 in sequence),
//   * evaluates the repellers j, j + 8, ...                           (1 slot per lane per round instead of 8);
// everything else is replicated on the 8 lanes: the chained joint transforms, the attractor, the 6 x 6 LDL^T and its
// solves are serial, and what is parallel in form -- the 21 entries of J J^T, the rows of J^T y, the Gram-Schmidt
// projections -- would need every lane to pick ITS operands out of register arrays that all lanes hold alike (a chain of
// selects per operand), or the sums to cross lanes: measured on a lone wave (tools/ubench_xlane.hip) a 7-term float64 dot
// product costs 38-42 cycles replicated, 58-77 as a DPP reduction over 8 lanes (gfx950 has no DPP form of the 64-bit
// VALU ops: every double moves as two 32-bit DPP moves), 141-195 through LDS.  Exchange through 1 KiB of LDS per arm
// (sin / cos, the repellers' partial sums), without barriers: the wave is the workgroup and its LDS queue is in order.
// With 4 096 arms the launch has 512 waves instead of 64.  Plain loads: a batch this small has no bandwidth to stage
// for.  Same arithmetic as cycle_kernel up to the order of the field sum; lane 0 of each group stores the arm's rows.
// ------------------------------------------------------------------------------------------------
#define VFIK_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")  // one wave per block: in-order LDS queue, no s_barrier needed
template <typename T, int NJ, bool NS, int DHP>
__device__ __forceinline__ void cycle_sub8_body(const KArgs& a) {
    // DHP: bit 0 the chain's DH pattern, bit 1 (TOOLC) ONE tool for the batch -- as in cycle_body (what vfclik itself runs is one or two arms with a hand)
    constexpr int DHPAT = DHP & 1;
    constexpr bool TOOLC = (DHP & 2) != 0;
    constexpr bool WTSC = (DHP & 4) != 0;    // bit 2: the batch's IK weights other than one (cycle_body, WTSC)
    static_assert(NJ <= 8 && (!NS || NJ <= 7), "one lane per joint; the sign memory is for chains of up to 7 joints");
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, j = lane & 7;
    const int arm_raw = blockIdx.x * 8 + g;
    const bool live = arm_raw < a.B;
    const int arm = live ? arm_raw : a.B - 1;  // every lane stays active for the exchanges; stores are masked
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    double* const L = reinterpret_cast<double*>(lds_all) + g * 128;  // this arm's 1 KiB
    typedef const KConst<NJ> __attribute__((address_space(4))) * KcPtr;
    const KcPtr kc = (KcPtr)(unsigned long long)a.kc;
    const KConst<NJ>* const kv = static_cast<const KConst<NJ>*>(a.kc);  // the same block through vector loads (lane-indexed)
    const long Bp = a.Bpad;
    const unsigned flags = NS ? a.flags : 0u;
    int status = 0;

    // ---- loads: joint angle of this lane, goal block (every lane of the group), this lane's first repeller
    const int jq = j < NJ ? j : NJ - 1;
    const double qj = (double)static_cast<const T*>(a.q)[(long)arm * NJ + jq];
    const double offj = kv->dh[jq].off;
    const T* gg = static_cast<const T*>(a.goal) + (long)arm * 4;
    double gq[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 4; ++c) gq[4 * k + c] = (double)gg[(long)k * Bp * 4 + c];
    }
    const T* sl = static_cast<const T*>(a.slots) + (long)arm * 4;
    auto load_slot = [&](int m, double* s8) {  // slot m of this arm: (x y z radius | safe order force type); past the end: force 0
        const bool in = m < a.slots_used;
        const long mm = in ? m : 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s8[c] = (double)sl[(2 * mm) * Bp * 4 + c];
            s8[4 + c] = (double)sl[(2 * mm + 1) * Bp * 4 + c];
        }
        if (!in) s8[6] = 0.0;
    };
    double s8[8];
    load_slot(j, s8);
    // nullspace state (as cycle_kernel keeps it: floats, (n + 4) / 4 planes) and /control, requested up front
    typedef float f4s __attribute__((ext_vector_type(4)));
    constexpr int NS_PLANES = (NJ + 4) / 4;
    int sig_r = 1;
    bool has_vec = false;
    double lv_r[NJ], c0 = 0.0;
    if constexpr (NS) {
        const f4s* sp = reinterpret_cast<const f4s*>(a.lastvec) + arm;
        float sv[NS_PLANES * 4];
#pragma unroll
        for (int k = 0; k < NS_PLANES; ++k) {
            const f4s v = sp[(long)k * a.Bpad];
            sv[4 * k] = v.x; sv[4 * k + 1] = v.y; sv[4 * k + 2] = v.z; sv[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) lv_r[i] = (double)sv[i];
        sig_r = sv[NJ] < 0.0f ? -1 : 1;
        has_vec = fabsf(sv[NJ]) > 1.5f;
        if (a.null_control) c0 = (double)static_cast<const T*>(a.null_control)[(long)arm * VFIK_NULL_CONTROLS];
    }
    // The members of the argument block the rest of the kernel needs, in ONE batch of scalar loads behind the requests above (entered
    // through cycle_sub8_kernel_x, whose scalars are preloaded, the compiler otherwise fetches them one by one where each is first
    // used, every time a full round trip to the kernarg segment).
    if constexpr (VFIK_SCALAR_KERNARG)
        asm volatile("" ::"s"(a.qdot_vf), "s"(a.qdot_null), "s"(a.pose), "s"(a.pose_nt), "s"(a.qdist), "s"(a.status));

    // ---- A3: sin / cos of joint j on lane j, exchanged through LDS
    {
        double sj, cj;
        sincos_fast(qj + offj, sj, cj);
        L[2 * j] = sj;
        L[2 * j + 1] = cj;
        L[32 + j] = qj;  // ... and the joint angles themselves (check_limits, distToCenter)
    }
    VFIK_WAVE_LDS_SYNC();
    double sn[NJ], cs[NJ], q[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) { sn[i] = L[2 * i]; cs[i] = L[2 * i + 1]; q[i] = L[32 + i]; }
    // the chain, replicated (constants through the scalar cache)
    double R[9], p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[3 * r + c] = kc->base[4 * r + c];
        p[r] = kc->base[4 * r + 3];
    }
    double Jm[NJ][6];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        Jm[i][3] = R[2]; Jm[i][4] = R[5]; Jm[i][5] = R[8];
        Jm[i][0] = p[0]; Jm[i][1] = p[1]; Jm[i][2] = p[2];
        const double ci = cs[i], si = sn[i];
        // (the chain's DH pattern, as in cycle_body: links that are a renaming, links that keep their frame, links without offset)
        const bool J_SWAP = (DhPattern<NJ, DHPAT>::SWAP >> i) & 1u, J_NONE = (DhPattern<NJ, DHPAT>::NONE >> i) & 1u, J_D0 = (DhPattern<NJ, DHPAT>::D0 >> i) & 1u;
        double xn[3], ym[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { xn[r] = si * R[3 * r + 1]; ym[r] = si * R[3 * r]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { xn[r] = __builtin_fma(ci, R[3 * r], xn[r]); ym[r] = __builtin_fma(ci, R[3 * r + 1], -ym[r]); }
        if (!J_D0) {
            const double di = kc->dh[i].d;
#pragma unroll
            for (int r = 0; r < 3; ++r) p[r] = __builtin_fma(di, R[3 * r + 2], p[r]);
        }
        if (J_SWAP) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { R[3 * r] = xn[r]; R[3 * r + 1] = R[3 * r + 2]; R[3 * r + 2] = -ym[r]; }
        } else if (J_NONE) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { R[3 * r] = xn[r]; R[3 * r + 1] = ym[r]; }
        } else {
            const double ai = kc->dh[i].a, ca = kc->dh[i].ca, sa = kc->dh[i].sa;
#pragma unroll
            for (int r = 0; r < 3; ++r) { p[r] = __builtin_fma(ai, xn[r], p[r]); R[3 * r] = xn[r]; }
            double t1[3], t2[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { t1[r] = sa * R[3 * r + 2]; t2[r] = sa * ym[r]; }
#pragma unroll
            for (int r = 0; r < 3; ++r) { R[3 * r + 1] = __builtin_fma(ca, ym[r], t1[r]); R[3 * r + 2] = __builtin_fma(ca, R[3 * r + 2], -t2[r]); }
        }
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {  // geometric Jacobian at the flange
        const double dx = p[0] - Jm[i][0], dy = p[1] - Jm[i][1], dz = p[2] - Jm[i][2];
        const double cx = Jm[i][4] * dz - Jm[i][5] * dy, cy = Jm[i][5] * dx - Jm[i][3] * dz, cz = Jm[i][3] * dy - Jm[i][4] * dx;
        Jm[i][0] = cx; Jm[i][1] = cy; Jm[i][2] = cz;
    }
    if constexpr (TOOLC) {   // A4 (vf:321-332): the field is evaluated at the tool pose; the flange frame is recomposed where /pose_no_tool asks for it
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double x = R[3 * r], y = R[3 * r + 1], z = R[3 * r + 2];
#pragma unroll
            for (int c = 0; c < 3; ++c) R[3 * r + c] = x * kc->tool[c] + y * kc->tool[4 + c] + z * kc->tool[8 + c];
            p[r] += x * kc->tool[3] + y * kc->tool[7] + z * kc->tool[11];
        }
    }

    // ---- A5: the repellers, one per lane and round; the group's sum through LDS in a fixed order
    double part[3] = {0.0, 0.0, 0.0};
    const int n0 = a.fast_order;
    for (int m0 = 0; m0 < a.slots_used; m0 += 8) {  // wave-uniform
        if (m0 > 0) load_slot(m0 + j, s8);
        const double dx = s8[0] - p[0], dy = s8[1] - p[1], dz = s8[2] - p[2];
        const double di = fmin(rsqrt_1nr(dx * dx + dy * dy + dz * dz), 1.0 / D_FLOOR);
        const double rb = (s8[3] + s8[4]) * di;
        const double k = s8[6] * fmin(powi_uniform(rb, n0), MAG_CAP) * di;
        part[0] += dx * k; part[1] += dy * k; part[2] += dz * k;
    }
    L[64 + 4 * j] = part[0]; L[64 + 4 * j + 1] = part[1]; L[64 + 4 * j + 2] = part[2];  // (a region of its own: no hazard with the reads above)
    VFIK_WAVE_LDS_SYNC();
    double tot[6] = {0, 0, 0, 0, 0, 0}, sc[2] = {1.0, 1.0}, gdist[2];
    {
        double GR[9], Gp[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) GR[3 * r + c] = gq[4 * r + c];
            Gp[r] = gq[4 * r + 3];
        }
        attractor(R, p, GR, Gp, gq[13], gq[14], kc->rot_slow, kc->cos_slow, gq[12] != 0.0, tot, sc, gdist);
    }
#pragma unroll
    for (int l = 0; l < 8; ++l) { tot[0] += L[64 + 4 * l]; tot[1] += L[64 + 4 * l + 1]; tot[2] += L[64 + 4 * l + 2]; }
    double tw[6];
    {
        double nt, nti, nr, nri;
        sqrt_rsqrt(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2], nt, nti);
        sqrt_rsqrt(tot[3] * tot[3] + tot[4] * tot[4] + tot[5] * tot[5], nr, nri);
        const double speed = gq[15];
        const double kt = nt > EPS_LEN ? speed * sc[0] * nti : 0.0;
        const double kr = nr > EPS_LEN ? speed * sc[1] * nri : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { tw[k] = tot[k] * kt; tw[3 + k] = tot[3 + k] * kr; }
    }
    if constexpr (TOOLC) {  // A6 (vf:456-459): the twist at the flange, shifted by p_ee - p_tip = -Rt (Rtool^T t)
        double c3[3], r3[3];
#pragma unroll
        for (int jx = 0; jx < 3; ++jx) c3[jx] = kc->tool[jx] * kc->tool[3] + kc->tool[4 + jx] * kc->tool[7] + kc->tool[8 + jx] * kc->tool[11];
#pragma unroll
        for (int r = 0; r < 3; ++r) r3[r] = -(R[3 * r] * c3[0] + R[3 * r + 1] * c3[1] + R[3 * r + 2] * c3[2]);
        tw[0] += tw[4] * r3[2] - tw[5] * r3[1];
        tw[1] += tw[5] * r3[0] - tw[3] * r3[2];
        tw[2] += tw[3] * r3[1] - tw[4] * r3[0];
    }

    // ---- A7: damped least squares (replicated)
    double A[6][6], dinv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) A[r][c] = (r == c && !WTSC) ? kc->lambda2 : 0.0;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        double t[6];
        if constexpr (WTSC) {   // A = Wy (J Wq^2 J^T) Wy + lambda^2 I, qdot = Wq^2 J^T (Wy y): as cycle_body
            const double w2 = kc->wq[i] * kc->wq[i];
#pragma unroll
            for (int r = 0; r < 6; ++r) t[r] = w2 * Jm[i][r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) A[r][c] = __builtin_fma(WTSC ? t[r] : Jm[i][r], Jm[i][c], A[r][c]);
    }
    if constexpr (WTSC) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) A[r][c] = __builtin_fma(kc->wy[r] * kc->wy[c], A[r][c], r == c ? kc->lambda2 : 0.0);
    }
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
        double v[6];
#pragma unroll
        for (int k = 0; k < jj; ++k) v[k] = A[jj][k] * A[k][k];
        double dj = A[jj][jj];
#pragma unroll
        for (int k = 0; k < jj; ++k) dj = __builtin_fma(-A[jj][k], v[k], dj);
        A[jj][jj] = dj;
        dinv[jj] = rcp_nr(dj);
#pragma unroll
        for (int k = 0; k < jj; ++k)
#pragma unroll
            for (int i = jj + 1; i < 6; ++i) A[i][jj] = __builtin_fma(-A[i][k], v[k], A[i][jj]);
#pragma unroll
        for (int i = jj + 1; i < 6; ++i) A[i][jj] *= dinv[jj];
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double t = WTSC ? kc->wy[i] * tw[i] : tw[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= A[i][k] * y[k];
        y[i] = t;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] *= dinv[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double t = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) t -= A[k][i] * y[k];
        y[i] = t;
    }
    if constexpr (WTSC) {
#pragma unroll
        for (int r = 0; r < 6; ++r) y[r] *= kc->wy[r];
    }
    double qv[NJ], qn[NJ], qo[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        qv[i] = Jm[i][0] * y[0];
#pragma unroll
        for (int r = 1; r < 6; ++r) qv[i] = __builtin_fma(Jm[i][r], y[r], qv[i]);
        if constexpr (WTSC) qv[i] *= kc->wq[i] * kc->wq[i];
        qn[i] = 0.0;
    }

    // ---- A10-A13: nullspace module, A15: mixer and limiter (replicated; the same code as cycle_kernel)
    bool advanced = false;
    if constexpr (NS) {
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int r = 0; r < 6; ++r) asm volatile("" : "+v"(Jm[i][r]) : "v"(qv[0]));  // the IK is done with J before it is orthonormalised
        advanced = nullspace_core<NJ>(Jm, lv_r, sig_r, has_vec, c0, (flags & VFIK_F_JOINT_LIMIT_TASK) != 0,
                                      [&](double* z) {
#pragma unroll
                                          for (int i = 0; i < NJ; ++i) z[i] = -kc->jl_k[i] * (q[i] - kc->q_mid[i]);
                                      },
                                      qn, status);
        const double look = kc->lookahead, ngain = kc->null_gain;
        int bad = 0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const double d = q[i] + look * qn[i];
            bad |= (int)(d < kc->q_lo[i]) | (int)(d > kc->q_hi[i]);
        }
        const bool stop = bad != 0;
        if (stop) status |= VFIK_ST_LIMIT_STOP;
#pragma unroll
        for (int i = 0; i < NJ; ++i) qn[i] = stop ? 0.0 : qn[i] * ngain;
    }
    if (flags & VFIK_F_MIXER) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) qo[i] = mac_unfused(mul_unfused(qv[i], kc->mix_w[0]), qn[i], kc->mix_w[1]);
    } else {
#pragma unroll
        for (int i = 0; i < NJ; ++i) qo[i] = qv[i];
    }
    if (flags & VFIK_F_LIMITER) {
        double lead = 0.0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) lead = fmax(lead, fabs(qo[i]));
        if (lead > kc->max_vel) {
            const double ratio = kc->max_vel * rcp_nr(lead);
#pragma unroll
            for (int i = 0; i < NJ; ++i) qo[i] *= ratio;
            status |= VFIK_ST_LIMITED;
        }
    }
    int nan = 0;
#pragma unroll
    for (int i = 0; i < NJ; ++i) nan |= (int)(qo[i] != qo[i]);
    if (nan) status |= VFIK_ST_NAN;

    // ---- outputs: every lane of the group holds the arm's results.  The mixed command goes through the arm's LDS area so that
    // lane j stores joint j (one store instruction for the wave); the other rows are stored by lane 0 of the group, element by
    // element.  (All rows through LDS, lanes j and j + 8 storing their elements: 0.4-0.6 us SLOWER at 1-512 arms -- two more
    // LDS round trips on a lone wave's critical path -- and no faster at 4 096; profiles/r03_latency_small_*.txt.)
    VFIK_WAVE_LDS_SYNC();  // (the area's earlier contents have been read)
#pragma unroll
    for (int i = 0; i < NJ; ++i) L[i] = qo[i];  // (all 8 lanes hold the same value)
    VFIK_WAVE_LDS_SYNC();
    if (live && j < NJ) static_cast<T*>(a.qdot_out)[(long)arm * NJ + j] = (T)L[j];
    if (live && j == 0) {
        if (a.qdot_vf) {
            T* o = static_cast<T*>(a.qdot_vf) + (long)arm * NJ;
#pragma unroll
            for (int i = 0; i < NJ; ++i) o[i] = (T)qv[i];
        }
        if (a.qdot_null) {
            T* o = static_cast<T*>(a.qdot_null) + (long)arm * NJ;
#pragma unroll
            for (int i = 0; i < NJ; ++i) o[i] = (T)qn[i];
        }
        if (a.pose || a.pose_nt) {  // without a tool: one frame
            T fr[16];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) fr[4 * r + c] = (T)R[3 * r + c];
                fr[4 * r + 3] = (T)p[r];
            }
            fr[12] = fr[13] = fr[14] = (T)0.0; fr[15] = (T)1.0;
            if (a.pose) {
                T* o = static_cast<T*>(a.pose) + (long)arm * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = fr[i];
            }
            if (a.pose_nt) {
                if constexpr (TOOLC) {   // the flange frame: R = Rt Rtool^T, p = pt - R t
                    double Rf[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Rf[3 * r + c] = R[3 * r] * kc->tool[4 * c] + R[3 * r + 1] * kc->tool[4 * c + 1] + R[3 * r + 2] * kc->tool[4 * c + 2];
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) fr[4 * r + c] = (T)Rf[3 * r + c];
                        fr[4 * r + 3] = (T)(p[r] - (Rf[3 * r] * kc->tool[3] + Rf[3 * r + 1] * kc->tool[7] + Rf[3 * r + 2] * kc->tool[11]));
                    }
                }
                T* o = static_cast<T*>(a.pose_nt) + (long)arm * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = fr[i];
            }
        }
        if (a.qdist) {
            T* o = static_cast<T*>(a.qdist) + (long)arm * NJ;
#pragma unroll
            for (int i = 0; i < NJ; ++i) o[i] = (T)(fabs(q[i] - kc->q_mid[i]) * kc->inv_half[i]);
        }
        if (a.status) a.status[arm] = status;
        if constexpr (NS) {
            if (advanced) {  // the sign memory, as cycle_kernel stores it
                f4s* sp = reinterpret_cast<f4s*>(a.lastvec) + arm;
                float sv[NS_PLANES * 4];
#pragma unroll
                for (int i = 0; i < NS_PLANES * 4; ++i) sv[i] = 0.0f;
#pragma unroll
                for (int i = 0; i < NJ; ++i) sv[i] = (float)lv_r[i];
                sv[NJ] = (float)sig_r * (has_vec ? 2.0f : 1.0f);
#pragma unroll
                for (int k = 0; k < NS_PLANES; ++k) {
                    f4s v;
                    v.x = sv[4 * k]; v.y = sv[4 * k + 1]; v.z = sv[4 * k + 2]; v.w = sv[4 * k + 3];
                    sp[(long)k * a.Bpad] = v;
                }
            }
        }
    }
}

// (entry points of the eight-lanes-per-arm kernel: the argument block alone, or the prologue's arguments as preloaded scalars in front)
template <typename T, int NJ, bool NS, int DHP = 0>
__global__ void __launch_bounds__(64) cycle_sub8_kernel(const KArgs a) {
    cycle_sub8_body<T, NJ, NS, DHP>(a);
}
template <typename T, int NJ, bool NS, int DHP = 0>
__global__ void __launch_bounds__(64)
cycle_sub8_kernel_x(const void* base, const void* q, void* qdot_out, const void* null_control, const void* slots, int B, int Bpad, int slots_used, unsigned flags,
                    const KArgs a_in) {
    KArgs a = a_in;
    args_from_scalars<T, NJ>(a, base, q, qdot_out, a_in.status, B, Bpad, slots_used, flags);
    a.slots = slots;
    a.null_control = null_control;   // (read up front; status is stored last and stays in the block)
    cycle_sub8_body<T, NJ, NS, DHP>(a);
}

// The argument block a kernel variant takes: KLean for the lean single-cycle straight-line variants, KArgs otherwise
template <int LEAN, bool ROLL, bool FASTF>
typename std::conditional<SmallArgs<LEAN, ROLL, FASTF>::value, KLean, KArgs>::type args_for(const KArgs& a) {
    if constexpr (SmallArgs<LEAN, ROLL, FASTF>::value) {
        KLean k;
        k.base = a.arena; k.q = a.q; k.qdot_out = a.qdot_out; k.status = a.status;
        k.B = a.B; k.Bpad = a.Bpad; k.slots_used = a.slots_used; k.fast_order = a.fast_order; k.flags = a.flags; k.block = a.block;
        return k;
    } else {
        return a;
    }
}

// Launch of a KLean variant (lean, single cycle, straight-line field path): scalar kernel arguments, or the argument block.
// UNI variants read the uniform repeller image, which sits a.uni_planes quad planes behind the compact one: the offset rides in the
// upper bits of fast_order (KLean's fourteen dwords are all taken).
// The DH pattern a variant is built with: the requested one for the lean / publishing-lean single-cycle straight-line float variants and the lean rollout
// (not the persistent one), none for every other variant -- so that asking for a pattern never multiplies the kernels of the rest.
// float64 I/O (what a port-level caller's bottles are): the pattern alone, for the 7-joint chain (the LWR) -- C3's batch with float64 I/O
// 6.12 -> 5.5 us; the shared-tool / shared-weights bits are float32-I/O only (launch_v).
template <typename T, bool PL, bool ROLL, bool FASTF, int LEAN, bool PERS, int NJ>
constexpr int dhp_of(int dhp) {
    return (PL && FASTF && (ROLL ? LEAN == 1 : LEAN != 0) && !PERS) ? (sizeof(T) == 4 ? dhp : (NJ == 7 ? (dhp & 1) : 0)) : 0;
}

template <typename T, int NJ, bool NS, bool PL, int CF = -1, bool PERS = false, bool FUN = false, int WAVES = 1, bool UNI = false, int DHP = 0>
void launch_lean(const KArgs& a_in, dim3 grid, dim3 blk, size_t lds, hipStream_t stream) {
    // (the two-waves build on the compact image spilled 9 registers per lane with the pattern: it keeps the general DH form)
    constexpr int D = (WAVES == 2 && !UNI) ? 0 : dhp_of<T, PL, false, true, 1, PERS, NJ>(DHP);
    KArgs a = a_in;
    if constexpr (UNI) a.fast_order = (a.fast_order & 255) | (a.uni_planes << 8);
    if constexpr (SmallArgs<1, false, true>::value && VFIK_SCALAR_KERNARG) {
        hipLaunchKernelGGL((cycle_kernel_s<T, NJ, NS, PL, false, true, 1, CF, PERS, FUN, WAVES, UNI, D>), grid, blk, lds, stream, (const void*)a.arena, a.q, a.qdot_out,
                           a.status, a.B, a.Bpad, a.slots_used, a.fast_order, a.flags, a.block);
    } else {
        hipLaunchKernelGGL((cycle_kernel<T, NJ, NS, PL, false, true, 1, CF, PERS, FUN, WAVES, UNI, false, D>), grid, blk, lds, stream, args_for<1, false, true>(a));
    }
}

// Launch of any other variant: the prologue's arguments as scalars in front of the argument block, or the block alone
template <typename T, int NJ, bool NS, bool PL, bool ROLL, bool FASTF, int LEAN, int CF = -1, bool PERS = false, bool FUN = false, bool UNI = false, bool MIXO = false, int DHP = 0>
void launch_full(const KArgs& a_in, dim3 grid, dim3 blk, size_t lds, hipStream_t stream) {
    constexpr int D = dhp_of<T, PL, ROLL, FASTF, LEAN, PERS, NJ>(DHP);
    if constexpr (SmallArgs<LEAN, ROLL, FASTF, MIXO>::value) {
        launch_lean<T, NJ, NS, PL, CF, PERS, FUN, 1, UNI, DHP>(a_in, grid, blk, lds, stream);
    } else {
        KArgs a = a_in;
        if constexpr (UNI) a.fast_order = (a.fast_order & 255) | (a.uni_planes << 8);
        if constexpr (VFIK_SCALAR_KERNARG) {
            hipLaunchKernelGGL((cycle_kernel_x<T, NJ, NS, PL, ROLL, FASTF, LEAN, CF, PERS, FUN, 1, UNI, MIXO, D>), grid, blk, lds, stream, (const void*)a.arena, a.q, a.qdot_out,
                               a.active, a.B, a.Bpad, a.slots_used, a.fast_order, a.flags, a.block, a);
        } else {
            hipLaunchKernelGGL((cycle_kernel<T, NJ, NS, PL, ROLL, FASTF, LEAN, CF, PERS, FUN, 1, UNI, MIXO, D>), grid, blk, lds, stream, a);
        }
    }
}

template <typename T, int NJ, bool NS, bool PL, int DHP = 0>
void launch_v(const KArgs& a_in, dim3 grid, dim3 blk, size_t lds, hipStream_t stream, int* sub8) {
    // FASTF: the straight-line repeller path and the general field path are separate kernels -- compiled into
    // one, the general path's code cost the straight-line launches 2.7 % (register allocation and layout).
    KArgs a = a_in;
    bool fastf = a.fast_order >= 0;
    // a funnel block in the batch (the goalAndNormal scene): the straight-line path has FUN variants for the lean single-cycle
    // launches of PLAIN chains; every other launch of such a batch takes the general path
    bool fun = false;
    if (fastf && a.has_funnel) {
        bool lean13 = false;
        if constexpr (PL)
            lean13 = (NS || a.flags == 0) && !a.tool_stride && !a.mixw && !a.wts && !a.ext && !a.q_ref && !a.q_cmded && !a.q_lo && !a.q_ref_out &&
                     !a.q_out && a.n_cycles == 0;
        if (lean13) fun = true;
        else { fastf = false; a.fast_order = -1; }
    }
    // decay orders that differ (between slots or arms): the MIXO variants serve the lean and the publishing-lean single-cycle launches
    // of PLAIN chains, with or without the aux block; every other launch of such a batch takes the general path
    bool mixo = false;
    if (fastf && a.mixed) {
        bool lean13 = false;
        if constexpr (PL)
            lean13 = (NS || a.flags == 0) && !a.tool_stride && !a.mixw && !a.wts && !a.ext && !a.q_ref && !a.q_cmded && !a.q_lo && !a.q_ref_out &&
                     !a.q_out && a.n_cycles == 0;
        if (lean13) mixo = true;
        else { fastf = false; fun = false; a.fast_order = -1; }
    }
    if (fastf) a.slots_used = a.slots_used_fast;  // (the straight-line path counts the slots of the compact image)
    // every decay repeller of the batch with one safe distance and one force (what the object feeder sends): the lean single-cycle
    // variants read the uniform image -- one quad per slot, the pair in the constants (cycle_body, UNI)
    const bool uni = fastf && !fun && !mixo && a.uni;
    // LEAN launches touch only the head of the region.  They ask for no more than that while the launch is at most one
    // wave per SIMD (C5 -2 %, C3N -0.6 %, C3 +-0 at 65 536 arms); beyond, the full size keeps the launch in rounds of one
    // wave per SIMD -- with eight waves resident per CU a 131 072-arm launch took 12.5 instead of 11.0 us (two waves
    // per SIMD compete for the same HBM time; profiles/r02_batch_scaling.txt).
    const size_t lds_lean = ((long)grid.x * (blk.x / 64) <= (long)a.n_simd) ? (size_t)(blk.x / 64) * Stage<T>::lean_bytes(NJ) : lds;
    bool lean = false, lean_any = false;  // lean: on the straight-line field path; lean_any: whatever the field path
    if constexpr (PL)
        lean_any = (NS || a.flags == 0) && !a.tool_stride && !a.mixw && !a.wts && !a.null_control && !a.ext && !a.q_ref && !a.q_cmded &&
               !a.qdot_vf && !a.qdot_null && !a.pose && !a.pose_nt && !a.v6 && !a.qdist && !a.goal_dist && !a.active && !a.q_lo &&
               !a.q_ref_out;
    lean = lean_any && fastf;
    // In-kernel rollouts (ROLL) exist for PLAIN chains of up to 7 joints; with a tool, IK weights or prismatic joints the loop-carried
    // state no longer fits the registers (12-268 B of scratch per lane until round 3) and the rollout is stepped by the host side
    // (vfik_abi.cpp, launch_cycles), as for the long chains.
    if constexpr (NJ <= VFIK_ROLL_MAX_NJ && PL) {
        if (a.n_cycles > 0) {
            if (lean) {
                launch_full<T, NJ, NS, PL, true, true, 1, -1, false, false, false, false, DHP>(a, grid, blk, lds_lean, stream);
                return;
            }
            if (fastf) launch_full<T, NJ, NS, PL, true, true, 0, -1, false, false, false, false, DHP>(a, grid, blk, lds, stream);
            else launch_full<T, NJ, NS, PL, true, false, 0, -1, false, false, false, false, DHP>(a, grid, blk, lds, stream);
            return;
        }
    }
    bool small8 = false;   // a launch the eight-lanes kernel serves, at a size where it wins
    if constexpr (PL && NJ <= (NS ? 7 : 8)) {
        // small batches: eight lanes per arm (cycle_sub8_kernel) for the launches it serves -- the straight-line field path, no
        // per-arm option, the outputs the per-arm processes publish every cycle.  VFIK_SUB8_MAX_BATCH = 0 switches it off.
        const bool served = fastf && !fun && !mixo && !a.tool_stride && !a.mixw && !a.wts && !a.ext && !a.q_ref && !a.q_cmded && !a.active && !a.q_lo &&
                            !a.q_ref_out && !a.v6 && !a.goal_dist && !a.q_out && a.n_cycles == 0 && a.qdot_out &&
                            (NS || (a.flags == 0 && !a.null_control));
        // Adopted where the same-box A/B wins (profiles/r03_latency_small_*.txt, 1 ... 4 096 arms): launches that publish the
        // per-cycle rows (pose, pose_no_tool, qdotOut, qdotout, qdist) -18 ... -22 % at every size; qdot_out alone without the
        // nullspace module -4 ... -20 %; qdot_out alone WITH it -7 ... -11 % for a handful of arms, +-3 % from 64 arms on.
        const bool rows = a.qdot_vf || a.qdot_null || a.pose || a.pose_nt || a.qdist;
        const int cap = rows ? a.sub8_max_batch_full : (NS ? a.sub8_max_batch_ns : a.sub8_max_batch);
        small8 = served && a.B <= cap;   // (with a shared tool / shared IK weights: the option block below launches this kernel's variants)
        if (served && a.B <= cap && a.plain == 1) {
            const dim3 g8((a.B + 7) / 8), b8(64);
            // (with the nullspace module the scalar entry measures 1-2 % SLOWER -- one arm 5.83 against 5.73 us, C2F 7.17 against 7.07 --
            // with or without a batch fetch of the block's other members: that variant keeps the block entry)
            if constexpr (VFIK_SCALAR_KERNARG && !NS)
                hipLaunchKernelGGL((cycle_sub8_kernel_x<T, NJ, NS, DHP>), g8, b8, 8 * 1024, stream, (const void*)a.arena, a.q, a.qdot_out, a.null_control, a.slots, a.B, a.Bpad,
                                   a.slots_used, a.flags, a);
            else
                hipLaunchKernelGGL((cycle_sub8_kernel<T, NJ, NS, DHP>), g8, b8, 8 * 1024, stream, a);
            if (sub8) *sub8 = 1;
            return;
        }
    }
    if constexpr (PL) {
        if (a.plain >= 2) {
            // The batch's shared options on the kernels built for plain chains: ONE tool for the batch (`set tool`, old/README.old:84;
            // a.plain - 1 bit 0) and / or IK weights other than one (/weight, vf:295-309; bit 1).  The lean and the publishing-lean
            // single-cycle float32 launches on the straight-line path -- what the default process set asks of an array caller and of
            // ControlCycleBatch -- and the eight-lanes kernel (both I/O types) have variants that apply them (cycle_body / cycle_sub8_body:
            // TOOLC = DHP bit 1, WTSC = DHP bit 2, the latter for chains of up to 7 joints; run-time flags; with the uniform image, the aux
            // block, the order planes), for the chain's DH pattern where one is built for the joint count.  Every other launch of such a
            // handle (float64 I/O beyond the eight-lanes sizes, a rollout, per-arm options, the general field path, a chain off its
            // pattern) takes the general variants, as until round 4.
            constexpr bool PAT = DHP == (DhPattern<NJ, 1>::SWAP != 0 ? 1 : 0);
            auto launch_opt = [&](auto dt) -> bool {
                constexpr int DT = decltype(dt)::value;
                if constexpr (PL && NJ <= (NS ? 7 : 8)) {
                    if (small8) {
                        const dim3 g8((a.B + 7) / 8), b8(64);
                        if constexpr (VFIK_SCALAR_KERNARG && !NS)
                            hipLaunchKernelGGL((cycle_sub8_kernel_x<T, NJ, NS, DT>), g8, b8, 8 * 1024, stream, (const void*)a.arena, a.q, a.qdot_out, a.null_control, a.slots, a.B,
                                               a.Bpad, a.slots_used, a.flags, a);
                        else
                            hipLaunchKernelGGL((cycle_sub8_kernel<T, NJ, NS, DT>), g8, b8, 8 * 1024, stream, a);
                        if (sub8) *sub8 = 1;
                        return true;
                    }
                }
                if constexpr (sizeof(T) == 4) {
                    const bool lean1 = lean && !a.q_out && a.n_cycles == 0;
                    const bool lean3 = fastf && (NS || a.flags == 0) && !a.tool_stride && !a.mixw && !a.wts && !a.ext && !a.q_ref && !a.q_cmded &&
                                       !a.q_lo && !a.q_ref_out && !a.q_out && a.n_cycles == 0;
                    if (!(lean1 || lean3)) return false;
                    if (mixo) {
                        const dim3 g64((unsigned)((a.B + 63) / 64)), b64(64);
                        size_t lds_m = (long)g64.x <= (long)a.n_simd ? Stage<T>::lean_bytes(NJ) : Stage<T>::bytes(NJ);
                        if (fun) lds_m = std::max(lds_m, (size_t)(Stage<T>::lean_bytes(NJ) + 6 * Stage<T>::QSTEP));
                        lds_m += 1024;
                        a.block = 64;
#define VFIK_LAUNCH_MT(LEANV, FUNV)                                                                                                         \
    hipLaunchKernelGGL((cycle_kernel_m<T, NJ, NS, LEANV, FUNV, DT>), g64, b64, lds_m, stream, (const void*)a.arena, a.q, a.qdot_out, a.active, a.orders, a.B, \
                       a.Bpad, a.slots_used, a.flags, a)
                        if (lean1 && fun) VFIK_LAUNCH_MT(1, true);
                        else if (lean1) VFIK_LAUNCH_MT(1, false);
                        else if (fun) VFIK_LAUNCH_MT(3, true);
                        else VFIK_LAUNCH_MT(3, false);
#undef VFIK_LAUNCH_MT
                        return true;
                    }
                    const size_t lds_funt = std::max(lds_lean, (size_t)(blk.x / 64) * (Stage<T>::lean_bytes(NJ) + 6 * Stage<T>::QSTEP));
                    if (lean1) {
                        if (fun) launch_lean<T, NJ, NS, PL, -1, false, true, 1, false, DT>(a, grid, blk, lds_funt, stream);
                        else if (uni) launch_lean<T, NJ, NS, PL, -1, false, false, 1, true, DT>(a, grid, blk, lds_lean, stream);
                        else launch_lean<T, NJ, NS, PL, -1, false, false, 1, false, DT>(a, grid, blk, lds_lean, stream);
                    } else {
                        if (fun) launch_full<T, NJ, NS, PL, false, true, 3, -1, false, true, false, false, DT>(a, grid, blk, lds_funt, stream);
                        else if (uni) launch_full<T, NJ, NS, PL, false, true, 3, -1, false, false, true, false, DT>(a, grid, blk, lds_lean, stream);
                        else launch_full<T, NJ, NS, PL, false, true, 3, -1, false, false, false, false, DT>(a, grid, blk, lds_lean, stream);
                    }
                    return true;
                }
                return false;
            };
            if constexpr (PAT) {
                bool done = false;
                const int opt = a.plain - 1;   // bit 0: shared tool, bit 1: shared IK weights
                if (opt == 1) done = launch_opt(std::integral_constant<int, DHP | 2>());
                if constexpr (NJ <= 7) {
                    if (opt == 2) done = launch_opt(std::integral_constant<int, DHP | 4>());
                    if (opt == 3) done = launch_opt(std::integral_constant<int, DHP | 6>());
                }
                if (done) return;
            }
            launch_v<T, NJ, NS, false, 0>(a_in, grid, blk, lds, stream, sub8);
            return;
        }
    }
    if constexpr (PL) {
        if (mixo) {   // (run-time flags: the compile-time flag sets of the default process set are worth ~1 % and a dozen more kernels)
            const bool lean1 = lean_any && !a.q_out;
            const dim3 g64((unsigned)((a.B + 63) / 64)), b64(64);    // one wave per block (cycle_kernel_m)
            size_t lds_m = (long)g64.x <= (long)a.n_simd ? Stage<T>::lean_bytes(NJ) : Stage<T>::bytes(NJ);   // (beyond one wave per SIMD: rounds, as above)
            if (fun) lds_m = std::max(lds_m, (size_t)(Stage<T>::lean_bytes(NJ) + 6 * Stage<T>::QSTEP));
            lds_m += 1024;
            a.block = 64;
#define VFIK_LAUNCH_M(LEANV, FUNV)                                                                                                              \
    hipLaunchKernelGGL((cycle_kernel_m<T, NJ, NS, LEANV, FUNV, dhp_of<T, PL, false, true, LEANV, false, NJ>(DHP)>), g64, b64, lds_m, stream, (const void*)a.arena, a.q, a.qdot_out, a.active, a.orders, a.B, \
                       a.Bpad, a.slots_used, a.flags, a)
            if (lean1 && fun) VFIK_LAUNCH_M(1, true);
            else if (lean1) VFIK_LAUNCH_M(1, false);
            else if (fun) VFIK_LAUNCH_M(3, true);
            else VFIK_LAUNCH_M(3, false);
#undef VFIK_LAUNCH_M
            return;
        }
    }
    if constexpr (PL && sizeof(T) == 4 && NJ <= 7) {
        // Batches beyond one wave per SIMD: the persistent launch -- one wave per SIMD, each striding over the 64-arm chunks
        // with the next chunk's inputs in flight under the current chunk's arithmetic (cycle_kernel, PERS).  Two per-arm
        // areas per wave: lean_bytes + kin_off = 37.5 KB for 7 joints, four waves per CU.
        const long nchunks = (a.B + 63) / 64;
        if (lean && !fun && !a.q_out && a.n_cycles == 0 && a.pers && nchunks > (long)a.n_simd) {
            const dim3 gp((unsigned)a.n_simd), bp(64);
            const size_t lds_p = Stage<T>::lean_bytes(NJ) + Stage<T>::kin_off(NJ);
            launch_lean<T, NJ, NS, PL, -1, true, false, 1, false, DHP>(a, gp, bp, lds_p, stream);
            return;
        }
    }
    // (FUN launches: the lean region + the aux block's six rows per wave; beyond one wave per SIMD the full region, as above)
    const size_t lds_fun = std::max(lds_lean, (size_t)(blk.x / 64) * (Stage<T>::lean_bytes(NJ) + 6 * Stage<T>::QSTEP));
    if constexpr (PL) {
        if (fun && lean && !a.q_out) {
            launch_lean<T, NJ, NS, PL, -1, false, true, 1, false, DHP>(a, grid, blk, lds_fun, stream);
            return;
        }
        if (lean && !a.q_out) {
            // (chains of up to 7 joints: C3N -1 %; the 14-joint kernel got 10 % SLOWER with its flags fixed -- the
            // compiler then hoists the joint-limit task's constants over the whole kernel -- and keeps them run-time)
            if constexpr (sizeof(T) == 4 && NJ <= 7) {
                // beyond one wave per SIMD: the two-waves-per-SIMD build, two blocks' lean regions resident per CU
                // (with the nullspace module only on the uniform repeller image: the compact-image variants of that build spilled 6 registers
                // per lane for 2-6 % -- they stay in rounds of one wave per SIMD)
                if (a.waves2 && (long)grid.x * (blk.x / 64) > (long)a.n_simd && (!NS || uni)) {
                    const size_t lds2 = (size_t)(blk.x / 64) * Stage<T>::lean_bytes(NJ);
                    constexpr int NSMIX = VFIK_F_NULLSPACE | VFIK_F_MIXER, NSJLMIX = NSMIX | VFIK_F_JOINT_LIMIT_TASK;
                    if constexpr (NS) {
                        if (a.flags == (unsigned)NSMIX) launch_lean<T, NJ, NS, PL, NSMIX, false, false, 2, true, DHP>(a, grid, blk, lds2, stream);
                        else if (a.flags == (unsigned)NSJLMIX) launch_lean<T, NJ, NS, PL, NSJLMIX, false, false, 2, true, DHP>(a, grid, blk, lds2, stream);
                        else launch_lean<T, NJ, NS, PL, -1, false, false, 2, true, DHP>(a, grid, blk, lds2, stream);
                    } else {
                        if (uni) launch_lean<T, NJ, NS, PL, -1, false, false, 2, true, DHP>(a, grid, blk, lds2, stream);
                        else launch_lean<T, NJ, NS, PL, -1, false, false, 2, false, DHP>(a, grid, blk, lds2, stream);
                    }
                    return;
                }
            }
            if constexpr (NS && NJ <= 7) {  // the flag sets of the default process set, as compile-time constants
                constexpr int NSMIX = VFIK_F_NULLSPACE | VFIK_F_MIXER, NSJLMIX = NSMIX | VFIK_F_JOINT_LIMIT_TASK;
                if (a.flags == (unsigned)NSMIX) {
                    { if (uni) launch_lean<T, NJ, NS, PL, NSMIX, false, false, 1, true, DHP>(a, grid, blk, lds_lean, stream); else launch_lean<T, NJ, NS, PL, NSMIX, false, false, 1, false, DHP>(a, grid, blk, lds_lean, stream); }
                    return;
                }
                if (a.flags == (unsigned)NSJLMIX) {
                    { if (uni) launch_lean<T, NJ, NS, PL, NSJLMIX, false, false, 1, true, DHP>(a, grid, blk, lds_lean, stream); else launch_lean<T, NJ, NS, PL, NSJLMIX, false, false, 1, false, DHP>(a, grid, blk, lds_lean, stream); }
                    return;
                }
            }
            { if (uni) launch_lean<T, NJ, NS, PL, -1, false, false, 1, true, DHP>(a, grid, blk, lds_lean, stream); else launch_lean<T, NJ, NS, PL, -1, false, false, 1, false, DHP>(a, grid, blk, lds_lean, stream); }
            return;
        }
        if constexpr (NJ > VFIK_ROLL_MAX_NJ) {  // a cycle of a stepped rollout: lean, but it integrates q on the way out
            if (lean) {
                launch_full<T, NJ, NS, PL, false, true, 2, -1, false, false, false, false, DHP>(a, grid, blk, lds_lean, stream);
                return;
            }
        }
    }
    if constexpr (PL) {
        // publishing lean launches (LEAN 3): the straight-line path, no per-arm option, single cycle
        const bool lean3 = fastf && (NS || a.flags == 0) && !a.tool_stride && !a.mixw && !a.wts && !a.ext && !a.q_ref && !a.q_cmded &&
                           !a.q_lo && !a.q_ref_out && !a.q_out && a.n_cycles == 0;
        if (lean3 && fun) {
            launch_full<T, NJ, NS, PL, false, true, 3, -1, false, true, false, false, DHP>(a, grid, blk, lds_fun, stream);
            return;
        }
        if (lean3) {
            if constexpr (NS && NJ <= 7) {
                constexpr int NSMIX = VFIK_F_NULLSPACE | VFIK_F_MIXER, NSJLMIX = NSMIX | VFIK_F_JOINT_LIMIT_TASK;
                if (a.flags == (unsigned)NSMIX) {
                    { if (uni) launch_full<T, NJ, NS, PL, false, true, 3, NSMIX, false, false, true, false, DHP>(a, grid, blk, lds_lean, stream); else launch_full<T, NJ, NS, PL, false, true, 3, NSMIX, false, false, false, false, DHP>(a, grid, blk, lds_lean, stream); }
                    return;
                }
                if (a.flags == (unsigned)NSJLMIX) {
                    { if (uni) launch_full<T, NJ, NS, PL, false, true, 3, NSJLMIX, false, false, true, false, DHP>(a, grid, blk, lds_lean, stream); else launch_full<T, NJ, NS, PL, false, true, 3, NSJLMIX, false, false, false, false, DHP>(a, grid, blk, lds_lean, stream); }
                    return;
                }
            }
            { if (uni) launch_full<T, NJ, NS, PL, false, true, 3, -1, false, false, true, false, DHP>(a, grid, blk, lds_lean, stream); else launch_full<T, NJ, NS, PL, false, true, 3, -1, false, false, false, false, DHP>(a, grid, blk, lds_lean, stream); }
            return;
        }
    }
    if constexpr (PL) {
        // the general field path (funnel / hemisphere / further attractors, mixed decay orders -- a goalAndNormal scene,
        // object_feeder:248-303) with nothing but q -> qdot_out asked for: its own LEAN variant (the optional inputs and
        // outputs as compile-time nulls free the registers the 14-joint kernel otherwise spills)
        if (lean_any && !fastf && !a.q_out) {
            launch_full<T, NJ, NS, PL, false, false, 1, -1, false, false, false, false, DHP>(a, grid, blk, lds, stream);
            return;
        }
    }
    // The non-lean single-cycle variants.  Those of the long chains sit at the register file's limit (512 per lane, the Jacobian alone
    // is 168) and are compiled as an object of their own with its own scheduling and allocation flags (Makefile, HEAVY): with the
    // flags that suit the lean kernels five of them spilled 12-128 B per lane.
#if defined(VFIK_ONLY_NJ) && VFIK_ONLY_NJ >= VFIK_HEAVY_MIN_NJ && !defined(VFIK_HEAVY_PART)
    VFIK_CAT(launch_heavy_nj, VFIK_ONLY_NJ)(sizeof(T) == 4 ? 32 : 64, NS, PL, fastf, a, grid, blk, lds, stream);
#else
    if (fastf) launch_full<T, NJ, NS, PL, false, true, 0, -1, false, false, false, false, DHP>(a, grid, blk, lds, stream);
    else launch_full<T, NJ, NS, PL, false, false, 0, -1, false, false, false, false, DHP>(a, grid, blk, lds, stream);
#endif
}

template <typename T, int NJ, bool NS>   // NS: the launch's flags carry VFIK_F_NULLSPACE (decided by the caller: one object per (n, T, NS))
hipError_t launch_t(const KArgs& a0, int block, hipStream_t stream, int* sub8) {
    KArgs a = a0;
    // (VFIK_BLOCK is a tuning knob: a block's waves must fit the CU's 160 KB of LDS with their regions)
    // A full region that does not fit the CU four times (float64 I/O from 10 joints on: 42-44 KB) would leave one SIMD of every CU idle
    // and run a 65 536-arm launch in two rounds.  Such launches go as one wave per block, each block asking for the part of the region
    // its options use: the rows of a per-arm tool and of per-arm mixer weights are the region's tail (14 joints, float64: 34 of 44 KB
    // without them, four blocks per CU again -- 34.3 -> 17 us with a shared tool, profiles/r04_heavy_variants.txt).
    size_t lds;
    if (4 * (size_t)Stage<T>::bytes(NJ) > 160u * 1024u) {
        block = 64;
        lds = (a.tool_stride || a.mixw) ? Stage<T>::bytes(NJ) : Stage<T>::tool_off(NJ);
    } else {
        while (block > 64 && (size_t)(block / 64) * Stage<T>::bytes(NJ) > 160u * 1024u) block -= 64;
        lds = (size_t)(block / 64) * Stage<T>::bytes(NJ);
    }
    a.block = block;
    const dim3 grid((a.B + block - 1) / block), blk(block);
    if (a.n_cycles > 0 && (a.plain != 1 || NJ > VFIK_ROLL_MAX_NJ)) return hipErrorInvalidValue;   // (stepped by the caller: launch_cycles; a tool too -- the rollout variants have no register to spare for it)
    if (a.plain) {
        if constexpr (DhPattern<NJ, 1>::SWAP != 0) {   // (float64 I/O: only the eight-lanes kernel has pattern variants, dhp_of)
            if (a.dhp == 1) {   // the chain matches the DH pattern built for this joint count (vfik_abi.cpp: upload_kconst)
                launch_v<T, NJ, NS, true, 1>(a, grid, blk, lds, stream, sub8);
                return hipGetLastError();
            }
        }
        launch_v<T, NJ, NS, true>(a, grid, blk, lds, stream, sub8);
    } else {
        launch_v<T, NJ, NS, false>(a, grid, blk, lds, stream, sub8);
    }
    return hipGetLastError();
}

}  // namespace

// The library is built from this one source compiled several times (csrc/Makefile): per joint count, I/O type and with / without the
// nullspace module with -DVFIK_ONLY_NJ=<n> -DVFIK_ONLY_T=<32|64> -DVFIK_ONLY_NS=<0|1> (the kernels of that combination, in parallel make
// jobs: sixteen objects of 20-70 kernels each instead of four of 85-160), once per long chain with -DVFIK_HEAVY_PART, and once with
// -DVFIK_DISPATCH (launch dispatch, mixer kernel, host-side constant preparation).
#ifdef VFIK_ONLY_NJ
#ifdef VFIK_HEAVY_PART
// -DVFIK_ONLY_NJ=<n> -DVFIK_HEAVY_PART: the non-lean single-cycle variants of a long chain, and nothing else
void VFIK_CAT(launch_heavy_nj, VFIK_ONLY_NJ)(int io_dtype, bool ns, bool plain, bool fastf, const KArgs& a, dim3 grid, dim3 blk, size_t lds, hipStream_t stream) {
#define VFIK_HEAVY(T, NS, PL)                                                                                                      \
    do {                                                                                                                            \
        if (fastf) launch_full<T, VFIK_ONLY_NJ, NS, PL, false, true, 0, -1, false, false, false, false, 0>(a, grid, blk, lds, stream);  \
        else launch_full<T, VFIK_ONLY_NJ, NS, PL, false, false, 0, -1, false, false, false, false, 0>(a, grid, blk, lds, stream);       \
    } while (0)
    if (io_dtype == 32) {
        if (ns) { if (plain) VFIK_HEAVY(float, true, true); else VFIK_HEAVY(float, true, false); }
        else { if (plain) VFIK_HEAVY(float, false, true); else VFIK_HEAVY(float, false, false); }
    } else {
        if (ns) { if (plain) VFIK_HEAVY(double, true, true); else VFIK_HEAVY(double, true, false); }
        else { if (plain) VFIK_HEAVY(double, false, true); else VFIK_HEAVY(double, false, false); }
    }
#undef VFIK_HEAVY
}
#else
#if VFIK_ONLY_T == 32
typedef float VfikOnlyT;
#else
typedef double VfikOnlyT;
#endif
#define VFIK_PART_NAME VFIK_CAT(VFIK_CAT(VFIK_CAT(VFIK_CAT(VFIK_CAT(launch_cycle_nj, VFIK_ONLY_NJ), _t), VFIK_ONLY_T), _ns), VFIK_ONLY_NS)
hipError_t VFIK_PART_NAME(const KArgs& kargs, int block, hipStream_t stream, int* sub8) {
    return launch_t<VfikOnlyT, VFIK_ONLY_NJ, VFIK_ONLY_NS != 0>(kargs, block, stream, sub8);
}
#endif
}  // namespace vfik
#else  // VFIK_DISPATCH

#define X(n)                                                                                                      \
    hipError_t launch_cycle_nj##n##_t32_ns0(const KArgs& kargs, int block, hipStream_t stream, int* sub8);       \
    hipError_t launch_cycle_nj##n##_t32_ns1(const KArgs& kargs, int block, hipStream_t stream, int* sub8);       \
    hipError_t launch_cycle_nj##n##_t64_ns0(const KArgs& kargs, int block, hipStream_t stream, int* sub8);       \
    hipError_t launch_cycle_nj##n##_t64_ns1(const KArgs& kargs, int block, hipStream_t stream, int* sub8);
VFIK_NJ_LIST
#undef X

namespace {
// ------------------------------------------------------------------------------------------------
// host: z-normal form -> DH form.  Every fixed transform factors as
//     B = Rz(theta) Tz(d) Tx(a) Rx(alpha) Rz(phi) Tz(e)          (ZXZ Euler angles + common normal)
// and the z-screws commute with the joint motions on either side, so the trailing (phi, e) of B[i]
// joins the leading (theta, d) of B[i+1] and the variable of joint i+1.  B[0] stays general.
// ------------------------------------------------------------------------------------------------
struct Screws { double theta, d, a, alpha, phi, e; };

static Screws dh_factor(const double* B) {
    const double R00 = B[0], R10 = B[4], R02 = B[2], R12 = B[6], R20 = B[8], R21 = B[9], R22 = B[10];
    const double tx = B[3], ty = B[7], tz = B[11];
    Screws s{};
    const double sa = std::sqrt(R02 * R02 + R12 * R12);
    s.alpha = std::atan2(sa, R22);
    if (sa > 1e-12) {
        s.theta = std::atan2(R02, -R12);
        s.phi = std::atan2(R20, R21);
        const double ct = std::cos(s.theta), st = std::sin(s.theta);
        const double ux = ct * tx + st * ty, uy = -st * tx + ct * ty;
        s.a = ux;
        s.e = -uy / sa;
        s.d = tz - s.e * R22;
    } else {  // consecutive axes parallel (alpha = 0) or anti-parallel (alpha = pi)
        s.theta = (tx * tx + ty * ty > 1e-24) ? std::atan2(ty, tx) : 0.0;
        s.a = std::sqrt(tx * tx + ty * ty);
        s.e = 0.0;
        s.d = tz;
        const double g = std::atan2(R10, R00);
        s.phi = R22 > 0.0 ? g - s.theta : s.theta - g;
    }
    return s;
}

static void mat_mul(const double* A, const double* B, double* C) {  // 3x4 row-major frames
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) C[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
        C[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
    }
}

static double dh_recompose_error(const Screws& s, const double* B) {
    auto rz = [](double t, double d, double* M) { const double c = std::cos(t), sn = std::sin(t); const double m[12] = {c, -sn, 0, 0, sn, c, 0, 0, 0, 0, 1, d}; memcpy(M, m, sizeof m); };
    auto rx = [](double al, double a, double* M) { const double c = std::cos(al), sn = std::sin(al); const double m[12] = {1, 0, 0, a, 0, c, -sn, 0, 0, sn, c, 0}; memcpy(M, m, sizeof m); };
    double A[12], X[12], C[12], T1[12], T2[12];
    rz(s.theta, s.d, A); rx(s.alpha, s.a, X); rz(s.phi, s.e, C);
    mat_mul(A, X, T1); mat_mul(T1, C, T2);
    double err = 0.0;
    for (int k = 0; k < 12; ++k) err = std::fmax(err, std::fabs(T2[k] - B[k]));
    return err;
}

template <int NJ>
double kconst_fill_t(void* dst, const vfik_chain& ch, const vfik_params& p, const double* tool12, int* plain, int* dhp) {
    KConst<NJ>& c = *static_cast<KConst<NJ>*>(dst);
    memset(&c, 0, sizeof c);
    memcpy(c.base, ch.B[0], sizeof c.base);
    double phi_prev = 0.0, e_prev = 0.0, worst = 0.0;
    unsigned m_swap = 0, m_none = 0, m_d0 = 0, m_off0 = 0, m_offpi = 0;   // the chain's DH pattern (vfik_kernel.h: DhPattern)
    for (int i = 0; i < NJ; ++i) {
        const Screws s = dh_factor(ch.B[i + 1]);
        worst = std::fmax(worst, dh_recompose_error(s, ch.B[i + 1]));
        const double off = phi_prev + s.theta;
        const bool pris = ch.jtype[i] == 1;
        c.dh[i].off = off;
        c.dh[i].crev = pris ? 0.0 : 1.0;
        c.dh[i].cprs = pris ? std::cos(off) : 0.0;
        c.dh[i].sprs = pris ? std::sin(off) : 0.0;
        c.dh[i].qd = pris ? 1.0 : 0.0;
        c.dh[i].d = e_prev + s.d;
        c.dh[i].a = s.a;
        c.dh[i].ca = std::cos(s.alpha);
        c.dh[i].sa = std::sin(s.alpha);
        // exact zeros and ones where the geometry has them (cos(pi/2) is 6e-17 in floating point): the DH pattern masks below -- and
        // the kernel variants that assume them -- rest on EXACT values; the snap moves a frame by less than 1e-15
        if (std::fabs(c.dh[i].a) < 1e-15) c.dh[i].a = 0.0;
        if (std::fabs(c.dh[i].d) < 1e-15) c.dh[i].d = 0.0;
        if (std::fabs(c.dh[i].ca) < 1e-15 && c.dh[i].sa > 0.0) { c.dh[i].ca = 0.0; c.dh[i].sa = 1.0; }
        if (std::fabs(c.dh[i].sa) < 1e-15 && c.dh[i].ca > 0.0) { c.dh[i].sa = 0.0; c.dh[i].ca = 1.0; }
        if (!pris && c.dh[i].a == 0.0 && c.dh[i].ca == 0.0 && c.dh[i].sa == 1.0) m_swap |= 1u << i;
        if (!pris && c.dh[i].a == 0.0 && c.dh[i].ca == 1.0 && c.dh[i].sa == 0.0) m_none |= 1u << i;
        if (!pris && c.dh[i].d == 0.0) m_d0 |= 1u << i;
        {   // an offset that is a multiple of pi: snapped to it, so that the pattern's "no offset" / "negate" is exact for this chain
            const double kpi = off / 3.14159265358979323846;
            const double kr = std::nearbyint(kpi);
            if (!pris && std::fabs(kpi - kr) < 1e-12) {
                c.dh[i].off = kr * 3.14159265358979323846;
                if (((long)kr) % 2 == 0) m_off0 |= 1u << i; else m_offpi |= 1u << i;
            }
        }
        phi_prev = s.phi;
        e_prev = s.e;
        const double half = 0.5 * (ch.q_hi[i] - ch.q_lo[i]);
        c.q_lo[i] = ch.q_lo[i];
        c.q_hi[i] = ch.q_hi[i];
        c.q_mid[i] = 0.5 * (ch.q_lo[i] + ch.q_hi[i]);
        c.inv_half[i] = 1.0 / half;
        c.jl_k[i] = p.jl_gain / (half * half);
        c.wq[i] = p.wq[i];
        if (ch.jtype[i] == 1) c.prismatic_mask |= 1u << i;
    }
    c.tail_c = std::cos(phi_prev);
    c.tail_s = std::sin(phi_prev);
    c.tail_e = e_prev;
    for (int i = 0; i < 6; ++i) c.wy[i] = p.wy[i];
    for (int i = 0; i < VFIK_MIX_CHANNELS; ++i) c.mix_w[i] = p.mix_w[i];
    for (int k = 0; k < 12; ++k) c.tool[k] = tool12[k];
    c.speed = p.speed_scale;
    c.lambda2 = p.lambda * p.lambda;
    c.rot_slow = p.rot_slowdown;
    c.cos_slow = p.rot_slowdown > 0.0 && p.rot_slowdown < 3.14159265358979323846 ? std::cos(p.rot_slowdown) : -2.0;  // -2: always evaluate the angle
    c.null_gain = p.null_gain;
    c.lookahead = p.lookahead;
    c.max_vel = p.max_vel;
    c.jp_kp = p.jp_kp;
    c.jp_delta = p.jp_delta;
    c.jl_gain = p.jl_gain;
    // PLAIN variant of the kernel: revolute joints only, no trailing screw.  *plain: 0 general variants; else 1 + (the batch's shared tool
    // is not the identity ? 1 : 0) + (its IK weights are not all one ? 2 : 0): the lean float32 kernels and the eight-lanes kernel have
    // variants that apply a shared tool and shared weights themselves (cycle_body: TOOLC, WTSC), every other launch of such a handle takes
    // the general variants (launch_v).
    bool pl = c.prismatic_mask == 0 && c.tail_c == 1.0 && c.tail_s == 0.0 && c.tail_e == 0.0;
    static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    bool tool_ident = true;
    for (int k = 0; k < 12; ++k) tool_ident = tool_ident && tool12[k] == ident[k];
    bool unit_w = true;
    for (int i = 0; i < 6; ++i) unit_w = unit_w && p.wy[i] == 1.0;
    for (int i = 0; i < NJ; ++i) unit_w = unit_w && p.wq[i] == 1.0;
    *plain = pl ? 1 + (tool_ident ? 0 : 1) + (unit_w ? 0 : 2) : 0;
    bool base_i = true;
    {
        static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        for (int k = 0; k < 12; ++k) base_i = base_i && c.base[k] == ident[k];
    }
    if (dhp) *dhp = pl ? dh_pattern_of(NJ, m_swap, m_none, m_d0, m_off0, m_offpi, base_i) : 0;
    // (sin, cos)(k pi/32), k = 0 .. 63, for sincos_tab_n
    double* tab = reinterpret_cast<double*>(static_cast<char*>(dst) + KTab<NJ>::OFFSET);
    for (int k = 0; k < 64; ++k) {
        const long double a = (long double)k * 3.14159265358979323846264338327950288L / 32.0L;
        tab[2 * k] = (double)sinl(a);
        tab[2 * k + 1] = (double)cosl(a);
    }
    return worst;
}

}  // namespace

int dh_pattern_of(int nj, unsigned swap, unsigned none, unsigned d0, unsigned off0, unsigned offpi, bool base_identity) {
    unsigned ps = 0, pn = 0, pd = 0, p0 = 0, ppi = 0;
    bool pb = false;
    switch (nj) {
#define X(n) case n: ps = DhPattern<n, 1>::SWAP; pn = DhPattern<n, 1>::NONE; pd = DhPattern<n, 1>::D0; p0 = DhPattern<n, 1>::OFF0; ppi = DhPattern<n, 1>::OFFPI; pb = DhPattern<n, 1>::BASE_I; break;
        VFIK_NJ_LIST
#undef X
        default: break;
    }
    if (ps == 0) return 0;
    // the chain's zeros must CONTAIN the pattern's (more zeros are fine: they are multiplied out); its pi offsets must be the pattern's
    return ((swap & ps) == ps && (none & pn) == pn && (d0 & pd) == pd && (off0 & p0) == p0 && (offpi & ppi) == ppi && (!pb || base_identity)) ? 1 : 0;
}

uint32_t supported_joints_mask() {
    uint32_t m = 0;
#define X(n) m |= 1u << n;
    VFIK_NJ_LIST
#undef X
    return m;
}

hipError_t launch_cycle(int io_dtype, int nj, const KArgs& kargs, int block, hipStream_t stream, int* sub8) {
    const bool ns = kargs.flags & VFIK_F_NULLSPACE;
    switch (nj) {
#define X(n)                                                                                                                        \
    case n:                                                                                                                         \
        return io_dtype == 32 ? (ns ? launch_cycle_nj##n##_t32_ns1(kargs, block, stream, sub8) : launch_cycle_nj##n##_t32_ns0(kargs, block, stream, sub8))  \
                              : (ns ? launch_cycle_nj##n##_t64_ns1(kargs, block, stream, sub8) : launch_cycle_nj##n##_t64_ns0(kargs, block, stream, sub8));
        VFIK_NJ_LIST
#undef X
        default: return hipErrorInvalidValue;
    }
}

size_t kconst_bytes(int nj) {  // the constants, padded to 1 KiB, and the sin / cos table behind them
    switch (nj) {
#define X(n) case n: return KTab<n>::OFFSET + 1024;
        VFIK_NJ_LIST
#undef X
        default: return 0;
    }
}

double kconst_fill(int nj, void* dst, const vfik_chain& chain, const vfik_params& p, const double* tool12, int* plain, int* dhp) {
    switch (nj) {
#define X(n) case n: return kconst_fill_t<n>(dst, chain, p, tool12, plain, dhp);
        VFIK_NJ_LIST
#undef X
        default: return 1e300;
    }
}

hipError_t launch_mix(int io_dtype, const void* cmds, const double* w_dev, int K, long count, long chan_stride,
                      void* out, hipStream_t stream) {
    const int block = 256;
    const dim3 grid((unsigned)((count + block - 1) / block)), blk(block);
    if (io_dtype == 32)
        hipLaunchKernelGGL(mix_kernel<float>, grid, blk, 0, stream, static_cast<const float*>(cmds), w_dev, K, count,
                           chan_stride, static_cast<float*>(out));
    else
        hipLaunchKernelGGL(mix_kernel<double>, grid, blk, 0, stream, static_cast<const double*>(cmds), w_dev, K, count,
                           chan_stride, static_cast<double*>(out));
    return hipGetLastError();
}

}  // namespace vfik

namespace vfik {
hipError_t launch_probe(int io_dtype, const void* pose, const void* goal, const void* slots, int B, long Bp, int slots_used,
                        double rot_slow, double cos_slow, void* out, hipStream_t stream) {
    const int block = 64;
    const dim3 grid((B + block - 1) / block), blk(block);
    if (io_dtype == 32)
        hipLaunchKernelGGL(probe_kernel<float>, grid, blk, 0, stream, static_cast<const float*>(pose), static_cast<const float*>(goal),
                           static_cast<const float*>(slots), B, Bp, slots_used, rot_slow, cos_slow, static_cast<float*>(out));
    else
        hipLaunchKernelGGL(probe_kernel<double>, grid, blk, 0, stream, static_cast<const double*>(pose), static_cast<const double*>(goal),
                           static_cast<const double*>(slots), B, Bp, slots_used, rot_slow, cos_slow, static_cast<double*>(out));
    return hipGetLastError();
}

hipError_t launch_monitor(int io_dtype, const void* pose, const void* frames, int O, long count, void* out, hipStream_t stream, const int* active) {
    const int block = 256;
    const dim3 grid((unsigned)((count + block - 1) / block)), blk(block);
    if (io_dtype == 32)
        hipLaunchKernelGGL(monitor_kernel<float>, grid, blk, 0, stream, static_cast<const float*>(pose),
                           static_cast<const float*>(frames), O, count, static_cast<float*>(out), active);
    else
        hipLaunchKernelGGL(monitor_kernel<double>, grid, blk, 0, stream, static_cast<const double*>(pose),
                           static_cast<const double*>(frames), O, count, static_cast<double*>(out), active);
    return hipGetLastError();
}

hipError_t launch_track(int io_dtype, const void* pose, const void* v6, double* state, void* out, const int* active, int B, hipStream_t stream) {
    const int block = 256;
    const dim3 grid((B + block - 1) / block), blk(block);
    if (io_dtype == 32)
        hipLaunchKernelGGL(track_kernel<float>, grid, blk, 0, stream, static_cast<const float*>(pose), static_cast<const float*>(v6),
                           state, static_cast<float*>(out), active, B);
    else
        hipLaunchKernelGGL(track_kernel<double>, grid, blk, 0, stream, static_cast<const double*>(pose),
                           static_cast<const double*>(v6), state, static_cast<double*>(out), active, B);
    return hipGetLastError();
}
}  // namespace vfik
#endif  // VFIK_ONLY_NJ / VFIK_DISPATCH
