// vfik_kernel.hip -- the fused control-cycle kernel for gfx950 (MI355X), one LANE per arm.
//
// One launch = one control cycle of B arms = the loop bodies of the reference's per-arm processes:
//   scripts/vf:311-347,455-466        q -> FK -> tool -> field -> twist -> RefPoint -> getIKV -> qdot
//   scripts/nullspace:162-184         J -> nullspace vector (sign memory) -> control -> check_limits
//   scripts/debug_jointlimits:61-73   distToCenter
//   src/command_mixer.py:78-82        weighted sum of the command channels (+ bridge:188-195 limiter)
//
// Mapping (DESIGN.md "Kernel"): every arm is an independent ~2 kflop float64 problem whose largest
// matrix is 6 x n (n <= 16); a wavefront evaluates 64 arms, one per lane, entirely in registers,
// with the chain constants as scalar (SGPR) operands from the kernarg block.  No cross-lane traffic
// is needed, no lane idles, and the per-arm field list is read as a structure of arrays so every
// wave-level load is one contiguous 256/512-byte row.  No MFMA: there is no contraction to feed it.
//
// Arithmetic is float64 whatever the io dtype (DESIGN.md "Precision").
#include "vfik_kernel.h"

namespace vfik {
namespace {

constexpr double EPS_LEN = 1e-12;  // lengths below this are zero (unit vector := 0)
constexpr double D_FLOOR = 1e-9;   // distance floor inside decay laws
constexpr double MAG_CAP = 1e6;    // cap of a repeller's magnitude

__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt(x * x + y * y + z * z); }

// acc + x*w with the product and the sum rounded separately, as CPython evaluates
// `result[i] += v[i] * w` (command_mixer.py:81).  HIP's __dmul_rn/__dadd_rn are plain operators that
// the compiler may still fuse, so contraction is switched off for this statement block.
__device__ __forceinline__ double mac_unfused(double acc, double x, double w) {
#pragma clang fp contract(off)
    const double prod = x * w;
    return acc + prod;
}

// x^order for x > 0.  Decay orders are small integers in every message the reference sends
// (object_feeder:277,279,302; README.old:75 uses 20), so take the multiply chain when we can and the
// general pow() only for a fractional order.
__device__ __forceinline__ double pow_order(double x, double order) {
    const int n = (int)order;
    if ((double)n == order && n >= 0 && n < 128) {
        double r = 1.0, b = x;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            r = (n & (1 << k)) ? r * b : r;
            b = b * b;
        }
        return r;
    }
    return pow(x, order);
}

// Rotation vector (base frame) taking R to G: log(G R^T) = KDL diff(R, G).rot.  Returns |r|.
__device__ double rot_log(const double* R, const double* G, double* r) {
    double E[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E[3 * i + j] = G[3 * i] * R[3 * j] + G[3 * i + 1] * R[3 * j + 1] + G[3 * i + 2] * R[3 * j + 2];
    const double a0 = 0.5 * (E[7] - E[5]), a1 = 0.5 * (E[2] - E[6]), a2 = 0.5 * (E[3] - E[1]);
    const double c = 0.5 * (E[0] + E[4] + E[8] - 1.0);
    const double s = norm3(a0, a1, a2);
    const double th = atan2(s, c);
    if (s < 1e-4 && c < 0.0) {
        // theta near pi (rare): axis from the symmetric part  c I + (1-c) a a^T
        const double omc = 1.0 - c;
        double x, y, z;
        if (E[0] >= E[4] && E[0] >= E[8]) {
            x = sqrt(fmax((E[0] - c) / omc, 0.0));
            y = 0.5 * (E[3] + E[1]) / (omc * x);
            z = 0.5 * (E[6] + E[2]) / (omc * x);
        } else if (E[4] >= E[8]) {
            y = sqrt(fmax((E[4] - c) / omc, 0.0));
            x = 0.5 * (E[1] + E[3]) / (omc * y);
            z = 0.5 * (E[7] + E[5]) / (omc * y);
        } else {
            z = sqrt(fmax((E[8] - c) / omc, 0.0));
            x = 0.5 * (E[2] + E[6]) / (omc * z);
            y = 0.5 * (E[5] + E[7]) / (omc * z);
        }
        if (x * a0 + y * a1 + z * a2 < 0.0) { x = -x; y = -y; z = -z; }
        const double k = th / norm3(x, y, z);
        r[0] = x * k; r[1] = y * k; r[2] = z * k;
        return th;
    }
    if (s < EPS_LEN) { r[0] = r[1] = r[2] = 0.0; return th; }
    const double k = th / s;
    r[0] = a0 * k; r[1] = a1 * k; r[2] = a2 * k;
    return th;
}

// type 1, point attractor: G = goal rotation (9) + position (3); adds force*vector to tot, scales sc
__device__ __forceinline__ void attractor(const double* R, const double* p, const double* GR, const double* Gp,
                                          double slow, double force, double rot_slow, double* tot, double* sc) {
    const double dx = Gp[0] - p[0], dy = Gp[1] - p[1], dz = Gp[2] - p[2];
    const double D = norm3(dx, dy, dz);
    if (D > EPS_LEN) {
        const double k = force / D;
        tot[0] += dx * k; tot[1] += dy * k; tot[2] += dz * k;
    }
    double r[3];
    const double th = rot_log(R, GR, r);
    if (th > EPS_LEN) {
        const double k = force / th;
        tot[3] += r[0] * k; tot[4] += r[1] * k; tot[5] += r[2] * k;
    }
    sc[0] *= slow > 0.0 ? fmin(1.0, D / slow) : 1.0;
    sc[1] *= rot_slow > 0.0 ? fmin(1.0, th / rot_slow) : 1.0;
}

template <typename T, int NJ, bool NULLSP>
__global__ void __launch_bounds__(256) cycle_kernel(const KArgs<NJ> a) {
    const int arm = blockIdx.x * blockDim.x + threadIdx.x;
    if (arm >= a.B) return;
    const long Bs = a.B;
    int status = 0;

    // ---------------- q ----------------------------------------------------------------------
    double q[NJ];
    {
        const T* qin = static_cast<const T*>(a.q) + (long)arm * NJ;
#pragma unroll
        for (int i = 0; i < NJ; ++i) q[i] = (double)qin[i];
    }

    // ---------------- A3: forward kinematics (vf:316-318) -------------------------------------
    double R[9], p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[3 * r + c] = a.CB[0][4 * r + c];
        p[r] = a.CB[0][4 * r + 3];
    }
    double Jv[NJ][3], Jw[NJ][3];  // first the joint origins / axes, then the Jacobian columns
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        Jw[i][0] = R[2]; Jw[i][1] = R[5]; Jw[i][2] = R[8];
        Jv[i][0] = p[0]; Jv[i][1] = p[1]; Jv[i][2] = p[2];
        if ((a.prismatic_mask >> i) & 1u) {
            p[0] += q[i] * R[2]; p[1] += q[i] * R[5]; p[2] += q[i] * R[8];
        } else {
            double s, c;
            sincos(q[i], &s, &c);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double x = R[3 * r], y = R[3 * r + 1];
                R[3 * r] = c * x + s * y;
                R[3 * r + 1] = c * y - s * x;
            }
        }
        double Rn[9], pn[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Rn[3 * r + c] = R[3 * r] * a.CB[i + 1][c] + R[3 * r + 1] * a.CB[i + 1][4 + c] + R[3 * r + 2] * a.CB[i + 1][8 + c];
            pn[r] = R[3 * r] * a.CB[i + 1][3] + R[3 * r + 1] * a.CB[i + 1][7] + R[3 * r + 2] * a.CB[i + 1][11] + p[r];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = pn[k];
    }
    // geometric Jacobian at the flange, base frame
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        if ((a.prismatic_mask >> i) & 1u) {
            Jv[i][0] = Jw[i][0]; Jv[i][1] = Jw[i][1]; Jv[i][2] = Jw[i][2];
            Jw[i][0] = Jw[i][1] = Jw[i][2] = 0.0;
        } else {
            const double dx = p[0] - Jv[i][0], dy = p[1] - Jv[i][1], dz = p[2] - Jv[i][2];
            Jv[i][0] = Jw[i][1] * dz - Jw[i][2] * dy;
            Jv[i][1] = Jw[i][2] * dx - Jw[i][0] * dz;
            Jv[i][2] = Jw[i][0] * dy - Jw[i][1] * dx;
        }
    }

    // ---------------- A4: tool offset (vf:321-332) --------------------------------------------
    double Rt[9], pt[3], rr[3];
    {
        double tl[12];
        const T* tp = static_cast<const T*>(a.tool);
        if (a.tool_per_arm) {
#pragma unroll
            for (int k = 0; k < 12; ++k) tl[k] = (double)tp[k * Bs + arm];
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) tl[k] = (double)tp[k];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[3 * r] * tl[c] + R[3 * r + 1] * tl[4 + c] + R[3 * r + 2] * tl[8 + c];
            rr[r] = -(R[3 * r] * tl[3] + R[3 * r + 1] * tl[7] + R[3 * r + 2] * tl[11]);  // p_ee - p_tip
            pt[r] = p[r] - rr[r];
        }
    }

    // ---------------- A5: vector field at the tool pose (vf:276-293,344-347) -------------------
    double tot[6] = {0, 0, 0, 0, 0, 0}, sc[2] = {1.0, 1.0};
    {
        const T* g = static_cast<const T*>(a.goal) + arm;
        if ((double)g[15 * Bs] != 0.0) {  // goal block = the arm's lowest-id attractor
            double GR[9], Gp[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) GR[3 * r + c] = (double)g[(4 * r + c) * Bs];
                Gp[r] = (double)g[(4 * r + 3) * Bs];
            }
            attractor(Rt, pt, GR, Gp, (double)g[16 * Bs], (double)g[17 * Bs], a.rot_slow, tot, sc);
        }
        const T* sp = static_cast<const T*>(a.slots) + arm;
        for (int m = 0; m < a.slots_used; ++m) {
            const T* s = sp + (long)m * 8 * Bs;
            const int type = (int)s[7 * Bs];
            if (type <= 0) continue;
            const double p0 = (double)s[0], p1 = (double)s[Bs], p2 = (double)s[2 * Bs], p3 = (double)s[3 * Bs],
                         p4 = (double)s[4 * Bs], p5 = (double)s[5 * Bs], force = (double)s[6 * Bs];
            if (type == VFIK_FIELD_REPELLER) {  // x y z radius safeDist order
                const double dx = p0 - pt[0], dy = p1 - pt[1], dz = p2 - pt[2];
                const double D = fmax(norm3(dx, dy, dz), D_FLOOR);
                const double mag = fmin(pow_order((p3 + p4) / D, p5), MAG_CAP);
                const double k = force * mag / D;
                tot[0] += dx * k; tot[1] += dy * k; tot[2] += dz * k;
            } else if (type == VFIK_FIELD_HEMISPHERE) {  // x y z nx ny nz | safeDist order
                const double safe = (double)s[8 * Bs], order = (double)s[9 * Bs];
                const double nn = norm3(p3, p4, p5);
                if (nn > EPS_LEN) {
                    const double h = ((pt[0] - p0) * p3 + (pt[1] - p1) * p4 + (pt[2] - p2) * p5) / nn;
                    const double mag = fmin(pow_order(safe / fmax(h, D_FLOOR), order), MAG_CAP);
                    const double k = -force * mag / nn;
                    tot[0] += p3 * k; tot[1] += p4 * k; tot[2] += p5 * k;
                }
            } else if (type == VFIK_FIELD_FUNNEL) {  // x y z ax ay az | cutAngle angleOrder cutDist distOrder
                const double cutA = (double)s[8 * Bs], ordA = (double)s[9 * Bs], cutD = (double)s[10 * Bs],
                             ordD = (double)s[11 * Bs];
                const double an = norm3(p3, p4, p5);
                if (an > EPS_LEN) {
                    const double ax = p3 / an, ay = p4 / an, az = p5 / an;
                    const double wx = pt[0] - p0, wy = pt[1] - p1, wz = pt[2] - p2;
                    const double along = wx * ax + wy * ay + wz * az;
                    const double ex = wx - along * ax, ey = wy - along * ay, ez = wz - along * az;
                    const double P = norm3(ex, ey, ez), dist = norm3(wx, wy, wz);
                    const double phi = atan2(P, along);
                    const double ga = cutA > 0.0 ? fmin(1.0, pow_order(phi / cutA, ordA)) : 1.0;
                    const double gd = fmin(1.0, pow_order(cutD / fmax(dist, D_FLOOR), ordD));
                    const double k = -force * ga * gd / fmax(P, D_FLOOR);
                    tot[0] += ex * k; tot[1] += ey * k; tot[2] += ez * k;
                }
            } else if (type == VFIK_FIELD_ATTRACTOR) {  // a second attractor: frame16 + slow over 3 slots
                double GR[9], Gp[3];
                GR[0] = p0; GR[1] = p1; GR[2] = p2; Gp[0] = p3; GR[3] = p4; GR[4] = p5;
                GR[5] = (double)s[8 * Bs]; Gp[1] = (double)s[9 * Bs];
                GR[6] = (double)s[10 * Bs]; GR[7] = (double)s[11 * Bs]; GR[8] = (double)s[12 * Bs];
                Gp[2] = (double)s[13 * Bs];
                attractor(Rt, pt, GR, Gp, (double)s[20 * Bs], force, a.rot_slow, tot, sc);
            }
        }
    }
    // normCart + speedScale * scalars (vf:292,346-347)
    double v[3], w[3];
    {
        const double nt = norm3(tot[0], tot[1], tot[2]), nr = norm3(tot[3], tot[4], tot[5]);
        const double kt = nt > EPS_LEN ? a.speed * sc[0] / nt : 0.0;
        const double kr = nr > EPS_LEN ? a.speed * sc[1] / nr : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k] = tot[k] * kt; w[k] = tot[3 + k] * kr; }
    }

    // ---------------- A6: Twist.RefPoint(p_ee - p_tip) (vf:456-459) ----------------------------
    double tw[6];
    tw[0] = v[0] + (w[1] * rr[2] - w[2] * rr[1]);
    tw[1] = v[1] + (w[2] * rr[0] - w[0] * rr[2]);
    tw[2] = v[2] + (w[0] * rr[1] - w[1] * rr[0]);
    tw[3] = w[0]; tw[4] = w[1]; tw[5] = w[2];

    // ---------------- A7: weighted damped least squares (vf:461) -------------------------------
    double qv[NJ];
    {
        // Jw' = Wy J Wq, kept as scaled columns
        double S[NJ][6];
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                S[i][r] = a.wy[r] * Jv[i][r] * a.wq[i];
                S[i][3 + r] = a.wy[3 + r] * Jw[i][r] * a.wq[i];
            }
        }
        double A[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double acc = (r == c) ? a.lambda2 : 0.0;
#pragma unroll
                for (int i = 0; i < NJ; ++i) acc += S[i][r] * S[i][c];
                A[r][c] = acc;
            }
        // LDL^T (unit lower L stored in A's strict lower part, d on the diagonal)
        double dinv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double dj = A[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) dj -= A[j][k] * A[j][k] * A[k][k];
            A[j][j] = dj;
            dinv[j] = 1.0 / dj;
#pragma unroll
            for (int i = j + 1; i < 6; ++i) {
                double t = A[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) t -= A[i][k] * A[j][k] * A[k][k];
                A[i][j] = t * dinv[j];
            }
        }
        double y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double t = a.wy[i] * tw[i];
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
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) acc += S[i][r] * y[r];
            qv[i] = a.wq[i] * acc;
        }
    }

    // ---------------- A10-A13: nullspace module (nullspace:95-131,162-184) ----------------------
    double qn[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) qn[i] = 0.0;
    if constexpr (NULLSP) {
        // Orthonormal basis Q (rows) of the row space of J by Gram-Schmidt with re-orthogonalisation;
        // I - Q^T Q is restrict(I6, J) = I - pinv(J) J (nullspace:75-79).
        double Q[6][NJ];
        int rank = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double u[NJ];
            double n0 = 0.0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                u[i] = r < 3 ? Jv[i][r < 3 ? r : 0] : Jw[i][r < 3 ? 0 : r - 3];
                n0 += u[i] * u[i];
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int s = 0; s < r; ++s) {
                    double c = 0.0;
#pragma unroll
                    for (int i = 0; i < NJ; ++i) c += Q[s][i] * u[i];
#pragma unroll
                    for (int i = 0; i < NJ; ++i) u[i] -= c * Q[s][i];
                }
            double n1 = 0.0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) n1 += u[i] * u[i];
            const bool keep = n1 > 1e-24 * n0 && n0 > 0.0;
            const double inv = keep ? 1.0 / sqrt(n1) : 0.0;
            rank += keep ? 1 : 0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) Q[r][i] = u[i] * inv;
        }
        const int nullity = NJ - rank;
        if (nullity == 1) {
            // the unique nullspace direction: normalised column of the projector with the largest diagonal
            double best = -1.0;
            int ib = 0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                double d = 1.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) d -= Q[r][i] * Q[r][i];
                if (d > best) { best = d; ib = i; }
            }
            double u[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) u[i] = (i == ib) ? 1.0 : 0.0;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double c = 0.0;
#pragma unroll
                    for (int i = 0; i < NJ; ++i) c += Q[r][i] * u[i];
#pragma unroll
                    for (int i = 0; i < NJ; ++i) u[i] -= c * Q[r][i];
                }
            double nn = 0.0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) nn += u[i] * u[i];
            nn = 1.0 / sqrt(nn);
            // raw sign as LAPACK's SVD leaves it (first non-negligible component negative; oracle + golden)
            bool found = false;
            double sg = 1.0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                u[i] *= nn;
                if (!found && fabs(u[i]) > 1e-9) { found = true; sg = u[i] > 0.0 ? -1.0 : 1.0; }
            }
            // sign continuity against the previous cycle (nullspace:101-105)
            int sig = a.sig[arm];
            double dm = 0.0, dp = 0.0;
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                u[i] *= sg;
                const double lv = a.lastvec[i * Bs + arm];
                const double x = sig * u[i] - lv, y = sig * u[i] + lv;
                dm += x * x; dp += y * y;
            }
            if (sqrt(dm) > sqrt(dp)) sig = -sig;
            a.sig[arm] = sig;
            double c0 = 0.0;
            if (a.null_control) c0 = (double)static_cast<const T*>(a.null_control)[(long)arm * VFIK_NULL_CONTROLS];
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                u[i] *= sig;
                a.lastvec[i * Bs + arm] = u[i];
                qn[i] = u[i] * c0;  // move_in_nullspace (nullspace:113-117): min(n, 4, 1) = 1 row
            }
        } else if (nullity >= 2) {
            status |= VFIK_ST_NULL_AMBIGUOUS;  // SVD basis not unique: /control cannot be honoured
        }
        if (a.flags & VFIK_F_JOINT_LIMIT_TASK) {
            double z[NJ];
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const double mid = 0.5 * (a.q_lo[i] + a.q_hi[i]), half = 0.5 * (a.q_hi[i] - a.q_lo[i]);
                z[i] = -a.jl_gain * (q[i] - mid) / (half * half);
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double c = 0.0;
#pragma unroll
                for (int i = 0; i < NJ; ++i) c += Q[r][i] * z[i];
#pragma unroll
                for (int i = 0; i < NJ; ++i) z[i] -= c * Q[r][i];
            }
#pragma unroll
            for (int i = 0; i < NJ; ++i) qn[i] += z[i];
        }
        // check_limits (nullspace:120-131) then gain (nullspace:183)
        bool stop = false;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const double d = q[i] + a.lookahead * qn[i];
            stop = stop || d < a.q_lo[i] || d > a.q_hi[i];
        }
        if (stop) status |= VFIK_ST_LIMIT_STOP;
#pragma unroll
        for (int i = 0; i < NJ; ++i) qn[i] = stop ? 0.0 : qn[i] * a.null_gain;
    }

    // ---------------- A15: command mixer (command_mixer.py:78-82) + limiter (bridge:188-195) ----
    double qo[NJ];
    if (a.flags & VFIK_F_MIXER) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            qo[i] = mac_unfused(mac_unfused(0.0, qv[i], a.mix_w[0]), qn[i], a.mix_w[1]);
        }
        if (a.ext) {
            const T* e = static_cast<const T*>(a.ext);
#pragma unroll
            for (int ch = 0; ch < VFIK_MIX_CHANNELS - 2; ++ch)
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    qo[i] = mac_unfused(qo[i], (double)e[((long)ch * Bs + arm) * NJ + i], a.mix_w[2 + ch]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NJ; ++i) qo[i] = qv[i];
    }
    if (a.flags & VFIK_F_LIMITER) {
        double lead = 0.0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) lead = fmax(lead, fabs(qo[i]));
        if (lead > a.max_vel) {
            const double ratio = a.max_vel / lead;
#pragma unroll
            for (int i = 0; i < NJ; ++i) qo[i] *= ratio;
            status |= VFIK_ST_LIMITED;
        }
    }
    bool nan = false;
#pragma unroll
    for (int i = 0; i < NJ; ++i) nan = nan || (qo[i] != qo[i]);
    if (nan) status |= VFIK_ST_NAN;

    // ---------------- outputs (vf:341-342,462-466; nullspace:180-184; debug_jointlimits:69-73) --
    if (a.qdot_out) {
        T* o = static_cast<T*>(a.qdot_out) + (long)arm * NJ;
#pragma unroll
        for (int i = 0; i < NJ; ++i) o[i] = (T)qo[i];
    }
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
    if (a.pose) {
        T* o = static_cast<T*>(a.pose) + (long)arm * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[4 * r + c] = (T)Rt[3 * r + c];
            o[4 * r + 3] = (T)pt[r];
        }
        o[12] = o[13] = o[14] = (T)0.0; o[15] = (T)1.0;
    }
    if (a.pose_nt) {
        T* o = static_cast<T*>(a.pose_nt) + (long)arm * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[4 * r + c] = (T)R[3 * r + c];
            o[4 * r + 3] = (T)p[r];
        }
        o[12] = o[13] = o[14] = (T)0.0; o[15] = (T)1.0;
    }
    if (a.v6) {
        T* o = static_cast<T*>(a.v6) + (long)arm * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) { o[k] = (T)v[k]; o[3 + k] = (T)w[k]; }
    }
    if (a.qdist) {
        T* o = static_cast<T*>(a.qdist) + (long)arm * NJ;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const double mid = 0.5 * (a.q_lo[i] + a.q_hi[i]), half = 0.5 * (a.q_hi[i] - a.q_lo[i]);
            o[i] = (T)(fabs(q[i] - mid) / half);
        }
    }
    if (a.status) a.status[arm] = status;
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

template <typename T, int NJ>
hipError_t launch_t(const KArgs<NJ>& a, int B, int block, hipStream_t stream) {
    const dim3 grid((B + block - 1) / block), blk(block);
    if (a.flags & VFIK_F_NULLSPACE)
        hipLaunchKernelGGL((cycle_kernel<T, NJ, true>), grid, blk, 0, stream, a);
    else
        hipLaunchKernelGGL((cycle_kernel<T, NJ, false>), grid, blk, 0, stream, a);
    return hipGetLastError();
}

template <int NJ>
hipError_t launch_nj(int io_dtype, const void* kargs, int B, int block, hipStream_t stream) {
    const KArgs<NJ>& a = *static_cast<const KArgs<NJ>*>(kargs);
    return io_dtype == 32 ? launch_t<float, NJ>(a, B, block, stream) : launch_t<double, NJ>(a, B, block, stream);
}

}  // namespace

uint32_t supported_joints_mask() {
    uint32_t m = 0;
#define X(n) m |= 1u << n;
    VFIK_NJ_LIST
#undef X
    return m;
}

hipError_t launch_cycle(int io_dtype, int nj, const void* kargs, int B, int block, hipStream_t stream) {
    switch (nj) {
#define X(n) case n: return launch_nj<n>(io_dtype, kargs, B, block, stream);
        VFIK_NJ_LIST
#undef X
        default: return hipErrorInvalidValue;
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
