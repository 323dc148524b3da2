/* vfik_oracle.c -- CPU restatement (float64) of vfclik's per-cycle control path.
 *
 * TEST INFRASTRUCTURE: see vfik_oracle.h for who may use this and for the pinning status of
 * every function.  Citations are file:line under /root/reference.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off so that vfo_mix reproduces CPython's
 * left-to-right double arithmetic bit for bit).
 */
#include "vfik_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXJ VFIK_MAX_JOINTS
#define EPS_LEN 1e-12   /* below this a length is treated as zero (unit vector := 0) */
#define D_FLOOR 1e-9    /* distance floor inside decay laws */
#define MAG_CAP 1e6     /* cap of a repeller's magnitude */

/* ------------------------------------------------------------------------------------------
 * small frame algebra (KDL semantics used at vf:330-332,456-459; frames are row-major 3x4)
 * ---------------------------------------------------------------------------------------- */
static void frame_mul(const double A[12], const double B[12], double C[12]) {
    /* PyKDL Frame * Frame (vf:330) */
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            C[4 * i + j] = A[4 * i + 0] * B[0 + j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
        C[4 * i + 3] = A[4 * i + 0] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
    }
}

static void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* A3  lafik.jntsList = q ; lafik.frame  (vf:316-318, debug_jointlimits:63) */
void vfo_fk(const vfik_chain* c, const double* q, double T[12], double* z, double* o) {
    double X[12], Y[12], Jz[12];
    memcpy(X, c->B[0], sizeof X);
    for (int i = 0; i < c->n; ++i) {
        if (z) { z[3 * i + 0] = X[2]; z[3 * i + 1] = X[6]; z[3 * i + 2] = X[10]; }
        if (o) { o[3 * i + 0] = X[3]; o[3 * i + 1] = X[7]; o[3 * i + 2] = X[11]; }
        memset(Jz, 0, sizeof Jz);
        Jz[0] = Jz[5] = Jz[10] = 1.0;
        if (c->jtype[i] == 0) {
            double s = sin(q[i]), co = cos(q[i]);
            Jz[0] = co; Jz[1] = -s; Jz[4] = s; Jz[5] = co;
        } else {
            Jz[11] = q[i];
        }
        frame_mul(X, Jz, Y);
        frame_mul(Y, c->B[i + 1], X);
    }
    memcpy(T, X, sizeof X);
}

/* Geometric Jacobian, reference point = flange origin, expressed in the base frame: what
 * lafik.getIKV (vf:461) and rob.jac_list() (nullspace:175) are built on. */
void vfo_jacobian(const vfik_chain* c, const double* q, double* J, double T[12]) {
    double z[3 * MAXJ], o[3 * MAXJ];
    vfo_fk(c, q, T, z, o);
    const double pe[3] = {T[3], T[7], T[11]};
    int n = c->n;
    for (int i = 0; i < n; ++i) {
        const double* zi = z + 3 * i;
        if (c->jtype[i] == 0) {
            double r[3] = {pe[0] - o[3 * i], pe[1] - o[3 * i + 1], pe[2] - o[3 * i + 2]}, v[3];
            cross3(zi, r, v);
            for (int k = 0; k < 3; ++k) { J[k * n + i] = v[k]; J[(3 + k) * n + i] = zi[k]; }
        } else {
            for (int k = 0; k < 3; ++k) { J[k * n + i] = zi[k]; J[(3 + k) * n + i] = 0.0; }
        }
    }
}

/* rotation vector r (base frame) taking R to Rg: r = log(Rg * R^T) = KDL diff(R, Rg).
 * Returns theta = |r|. */
static double rot_log(const double F[12], const double G[12], double r[3]) {
    double E[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            E[3 * i + j] = G[4 * i] * F[4 * j] + G[4 * i + 1] * F[4 * j + 1] + G[4 * i + 2] * F[4 * j + 2];
    double a[3] = {0.5 * (E[7] - E[5]), 0.5 * (E[2] - E[6]), 0.5 * (E[3] - E[1])};
    double c = 0.5 * (E[0] + E[4] + E[8] - 1.0);
    double s = norm3(a);
    double th = atan2(s, c);
    if (s < 1e-4 && c < 0.0) {
        /* theta near pi: axis from the symmetric part  S = c I + (1-c) a a^T */
        double d[3] = {E[0], E[4], E[8]};
        int k = 0;
        if (d[1] > d[k]) k = 1;
        if (d[2] > d[k]) k = 2;
        double ax[3];
        double akk = sqrt(fmax((d[k] - c) / (1.0 - c), 0.0));
        ax[k] = akk;
        for (int j = 0; j < 3; ++j)
            if (j != k) ax[j] = 0.5 * (E[3 * j + k] + E[3 * k + j]) / ((1.0 - c) * akk);
        if (ax[0] * a[0] + ax[1] * a[1] + ax[2] * a[2] < 0.0) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; }
        double nn = norm3(ax);
        for (int j = 0; j < 3; ++j) r[j] = ax[j] / nn * th;
        return th;
    }
    if (s < EPS_LEN) { r[0] = r[1] = r[2] = 0.0; return th; }
    for (int j = 0; j < 3; ++j) r[j] = a[j] / s * th;
    return th;
}

/* ------------------------------------------------------------------------------------------
 * A5  field primitives (vfl.vfl.vectorFieldLibrary, keys 0,1,2,4,5 -- BUILD-DEFINED, unpinned)
 *     getVector -> 6-vector, getScalar -> 2-vector, evaluated at the tool pose F
 * ---------------------------------------------------------------------------------------- */
static void prim_eval(const vfik_field* f, const double F[12], double rot_slowdown, double v[6], double sc[2]) {
    const double p[3] = {F[3], F[7], F[11]};
    for (int k = 0; k < 6; ++k) v[k] = 0.0;
    sc[0] = sc[1] = 1.0;
    switch (f->type) {
    case VFIK_FIELD_ATTRACTOR: { /* params: frame16 (row-major 4x4), slow-down distance */
        const double* G = f->p; /* rows 0..2 of the 4x4 are a 3x4 */
        double d[3] = {G[3] - p[0], G[7] - p[1], G[11] - p[2]};
        double D = norm3(d);
        if (D > EPS_LEN) for (int k = 0; k < 3; ++k) v[k] = d[k] / D;
        double r[3];
        double th = rot_log(F, G, r);
        if (th > EPS_LEN) for (int k = 0; k < 3; ++k) v[3 + k] = r[k] / th;
        double ds = f->p[16];
        sc[0] = ds > 0.0 ? fmin(1.0, D / ds) : 1.0;
        sc[1] = rot_slowdown > 0.0 ? fmin(1.0, th / rot_slowdown) : 1.0;
        break;
    }
    case VFIK_FIELD_REPELLER: { /* x y z radius safeDist order */
        double d[3] = {f->p[0] - p[0], f->p[1] - p[1], f->p[2] - p[2]};
        double D = fmax(norm3(d), D_FLOOR);
        double m = fmin(pow((f->p[3] + f->p[4]) / D, f->p[5]), MAG_CAP);
        for (int k = 0; k < 3; ++k) v[k] = m * d[k] / D;
        break;
    }
    case VFIK_FIELD_HEMISPHERE: { /* x y z nx ny nz safeDist order */
        double nn = norm3(f->p + 3);
        if (nn > EPS_LEN) {
            double h = 0.0;
            for (int k = 0; k < 3; ++k) h += (p[k] - f->p[k]) * f->p[3 + k] / nn;
            double m = fmin(pow(f->p[6] / fmax(h, D_FLOOR), f->p[7]), MAG_CAP);
            for (int k = 0; k < 3; ++k) v[k] = -m * f->p[3 + k] / nn;
        }
        break;
    }
    case VFIK_FIELD_FUNNEL: { /* x y z ax ay az cutAngle angleOrder cutDist distOrder */
        double an = norm3(f->p + 3);
        if (an > EPS_LEN) {
            double w[3], perp[3], along = 0.0;
            for (int k = 0; k < 3; ++k) { w[k] = p[k] - f->p[k]; along += w[k] * f->p[3 + k] / an; }
            for (int k = 0; k < 3; ++k) perp[k] = w[k] - along * f->p[3 + k] / an;
            double P = norm3(perp), dist = norm3(w);
            double phi = atan2(P, along);
            double ga = f->p[6] > 0.0 ? fmin(1.0, pow(phi / f->p[6], f->p[7])) : 1.0;
            double gd = fmin(1.0, pow(f->p[8] / fmax(dist, D_FLOOR), f->p[9]));
            for (int k = 0; k < 3; ++k) v[k] = -perp[k] / fmax(P, D_FLOOR) * ga * gd;
        }
        break;
    }
    default: /* VFIK_FIELD_NULL (vf:148-151) and unknown types contribute nothing */
        break;
    }
}

/* total = sum_k force_k * VF_k ; totalSF = prod_k SF_k ; totalVF = total.normCart()
 * (vf:276-293), evaluated at F (vf:344-345).  normCart (BUILD-DEFINED): the translational and
 * the rotational 3-vectors are each scaled to unit length (zero stays zero). */
void vfo_field_eval(const vfik_field* f, int nf, const double F[12], double rot_slowdown,
                    double vec6[6], double sc2[2]) {
    double tot[6] = {0, 0, 0, 0, 0, 0};
    sc2[0] = sc2[1] = 1.0;
    for (int i = 0; i < nf; ++i) {
        double v[6], s[2];
        prim_eval(&f[i], F, rot_slowdown, v, s);
        for (int k = 0; k < 6; ++k) tot[k] += v[k] * f[i].force;
        sc2[0] *= s[0];
        sc2[1] *= s[1];
    }
    double nt = norm3(tot), nr = norm3(tot + 3);
    for (int k = 0; k < 3; ++k) {
        vec6[k] = nt > EPS_LEN ? tot[k] / nt : 0.0;
        vec6[3 + k] = nr > EPS_LEN ? tot[3 + k] / nr : 0.0;
    }
}

/* ------------------------------------------------------------------------------------------
 * A7  qdot = lafik.getIKV(tw.vel, tw.rot) (vf:461).  north_star fixes the form:
 *     qdot = Wq Jw^T (Jw Jw^T + lambda^2 I)^-1 Wy tw ,  Jw = Wy J Wq   (weights diagonal, A2)
 * ---------------------------------------------------------------------------------------- */
void vfo_ikv(const double* J, int n, const double tw[6], const double wy[6], const double* wq,
             double lambda, double* qdot) {
    double Jw[6 * MAXJ], A[36], L[36], d[6], y[6];
    for (int r = 0; r < 6; ++r)
        for (int i = 0; i < n; ++i) Jw[r * n + i] = wy[r] * J[r * n + i] * wq[i];
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += Jw[r * n + i] * Jw[c * n + i];
            A[6 * r + c] = s + (r == c ? lambda * lambda : 0.0);
        }
    /* LDL^T */
    memset(L, 0, sizeof L);
    for (int j = 0; j < 6; ++j) {
        double s = A[6 * j + j];
        for (int k = 0; k < j; ++k) s -= L[6 * j + k] * L[6 * j + k] * d[k];
        d[j] = s;
        L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; ++i) {
            double t = A[6 * i + j];
            for (int k = 0; k < j; ++k) t -= L[6 * i + k] * L[6 * j + k] * d[k];
            L[6 * i + j] = t / s;
        }
    }
    for (int i = 0; i < 6; ++i) {
        double s = wy[i] * tw[i];
        for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
        y[i] = s;
    }
    for (int i = 0; i < 6; ++i) y[i] /= d[i];
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * y[k];
        y[i] = s;
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int r = 0; r < 6; ++r) s += Jw[r * n + i] * y[r];
        qdot[i] = wq[i] * s;
    }
}

/* ------------------------------------------------------------------------------------------
 * A10-A13  scripts/nullspace:75-131
 * ---------------------------------------------------------------------------------------- */

/* One-sided Jacobi on G = J^T (n x 6): returns orthogonal columns U*S (n x 6) and sigma[6]. */
static void onesided_jacobi(const double* J, int n, double* G, double sig[6]) {
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 6; ++r) G[i * 6 + r] = J[r * n + i];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int a = 0; a < 5; ++a)
            for (int b = a + 1; b < 6; ++b) {
                double aa = 0, bb = 0, ab = 0;
                for (int i = 0; i < n; ++i) {
                    aa += G[i * 6 + a] * G[i * 6 + a];
                    bb += G[i * 6 + b] * G[i * 6 + b];
                    ab += G[i * 6 + a] * G[i * 6 + b];
                }
                if (fabs(ab) <= 1e-300 || fabs(ab) <= 1e-17 * sqrt(aa * bb)) continue;
                off = fmax(off, fabs(ab) / sqrt(aa * bb));
                double tau = (bb - aa) / (2.0 * ab);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < n; ++i) {
                    double ga = G[i * 6 + a], gb = G[i * 6 + b];
                    G[i * 6 + a] = cs * ga - sn * gb;
                    G[i * 6 + b] = sn * ga + cs * gb;
                }
            }
        if (off < 1e-15) break;
    }
    for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += G[i * 6 + r] * G[i * 6 + r];
        sig[r] = sqrt(s);
    }
}

/* restrict(P, J) with P = I6 (nullspace:75-79,136): I - pinv(J) J.  numpy.linalg.pinv inverts
 * the singular values above rcond * max(s), rcond = 1e-15. */
void vfo_restrict(const double* J, int n, double* Bout) {
    double G[MAXJ * 6], sig[6], smax = 0.0;
    onesided_jacobi(J, n, G, sig);
    for (int r = 0; r < 6; ++r) smax = fmax(smax, sig[r]);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Bout[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int r = 0; r < 6; ++r) {
        if (!(sig[r] > 1e-15 * smax)) continue;
        double inv = 1.0 / (sig[r] * sig[r]);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) Bout[i * n + j] -= G[i * 6 + r] * G[j * 6 + r] * inv;
    }
}

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix; V columns = eigenvectors */
static void sym_jacobi(double* A, int n, double* V, double* w) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 80; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-34) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double tau = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

/* nullspace(P, J) (nullspace:95-107).  u,s,vh = svd(B.T): B is a symmetric projector, so its
 * left singular vectors are its eigenvectors, s in {1, 0}.  Raw sign of each u[:,i] follows what
 * LAPACK's Householder bidiagonalisation produces for these matrices -- the first component of
 * non-negligible size is negative -- which tests/golden/nullspace_golden.npz (n7__traj_raw_u)
 * confirms for the unique (nullity 1) case.  For nullity > 1 the basis inside the eigenspace is
 * whatever LAPACK returns and cannot be restated (VFIK_ST_NULL_AMBIGUOUS). */
int vfo_nullspace_basis(const double* J, int n, double* lastvec, int* sig, double* basis) {
    double B[MAXJ * MAXJ], V[MAXJ * MAXJ], w[MAXJ];
    int order[MAXJ];
    vfo_restrict(J, n, B);
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) B[i * n + j] = B[j * n + i] = 0.5 * (B[i * n + j] + B[j * n + i]);
    sym_jacobi(B, n, V, w);
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i) /* descending, stable */
        for (int j = i + 1; j < n; ++j)
            if (w[order[j]] > w[order[i]] + 1e-12) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    int i = 0;
    while (i < n && w[order[i]] >= 1e-8) { /* nullspace:100 */
        double u[MAXJ];
        for (int k = 0; k < n; ++k) u[k] = V[k * n + order[i]];
        for (int k = 0; k < n; ++k)
            if (fabs(u[k]) > 1e-9) {
                if (u[k] > 0) for (int m = 0; m < n; ++m) u[m] = -u[m];
                break;
            }
        double dm = 0, dp = 0; /* nullspace:101-103 */
        for (int k = 0; k < n; ++k) {
            double a = sig[i] * u[k] - lastvec[k * n + i], b = sig[i] * u[k] + lastvec[k * n + i];
            dm += a * a; dp += b * b;
        }
        if (sqrt(dm) > sqrt(dp)) sig[i] = -sig[i];
        for (int k = 0; k < n; ++k) { /* nullspace:104-105 */
            u[k] *= sig[i];
            lastvec[k * n + i] = u[k];
            basis[i * n + k] = u[k];
        }
        ++i;
    }
    return i;
}

/* move_in_nullspace (nullspace:110-117) */
void vfo_move_in_nullspace(const double* J, int n, const double* control, int ncontrol,
                           double* lastvec, int* sig, double* qdot, int* rank_out) {
    double basis[MAXJ * MAXJ];
    int r = vfo_nullspace_basis(J, n, lastvec, sig, basis);
    int m = n < ncontrol ? n : ncontrol;
    if (r < m) m = r;
    for (int k = 0; k < n; ++k) qdot[k] = 0.0;
    for (int i = 0; i < m; ++i)
        for (int k = 0; k < n; ++k) qdot[k] += basis[i * n + k] * control[i];
    if (rank_out) *rank_out = r;
}

/* check_limits (nullspace:120-131), margin = 0 */
int vfo_check_limits(const double* q, double* qdot, const double* lo, const double* hi, int n,
                     double scale) {
    for (int i = 0; i < n; ++i) {
        double d = q[i] + scale * qdot[i];
        if (d < lo[i] || d > hi[i]) {
            for (int k = 0; k < n; ++k) qdot[k] = 0.0;
            return 1;
        }
    }
    return 0;
}

/* A14  robot.distToCenter(limit_i, q_i) (debug_jointlimits:65-67).  BUILD-DEFINED, unpinned:
 * |q - mid| / half-range, 0 at the centre of the range, 1 on a limit. */
void vfo_dist_to_center(const double* q, const double* lo, const double* hi, int n, double* d) {
    for (int i = 0; i < n; ++i) {
        double mid = 0.5 * (lo[i] + hi[i]), half = 0.5 * (hi[i] - lo[i]);
        d[i] = fabs(q[i] - mid) / half;
    }
}

/* A15  CommandMixer.read, the weighted sum (command_mixer.py:78-82): result starts at 0.0 and
 * accumulates v[i]*w channel by channel.  Built with -ffp-contract=off. */
void vfo_mix(const double* cmd, const double* w, int K, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = 0.0;
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < n; ++i) out[i] += cmd[k * n + i] * w[k];
}

/* LWR_Bridge.set_vel, the limiter part (bridge:188-195) */
int vfo_limiter(double* qdot, int n, double max_vel) {
    double lead = 0.0;
    for (int i = 0; i < n; ++i) lead = fmax(lead, fabs(qdot[i]));
    if (lead > max_vel) {
        double ratio = max_vel / lead;
        for (int i = 0; i < n; ++i) qdot[i] *= ratio;
        return 1;
    }
    return 0;
}

/* scripts/joint_p_controller: check_limits (:89-99), error * kp (:127-128), at_goal (:134-138) */
int vfo_joint_p(const double* ref, const double* q, const double* lo, const double* hi, int n,
                double kp, double delta, double* out) {
    int reached = 1;
    for (int i = 0; i < n; ++i) {
        double r = ref[i];
        if (r < lo[i]) r = lo[i];
        else if (r > hi[i]) r = hi[i];
        double err = r - q[i];
        out[i] = err * kp;
        if (!(err < delta)) reached = 0; /* signed comparison, as the reference writes it */
    }
    return reached;
}

/* LWR_Bridge.set_vel, the command form (bridge:199-203) */
void vfo_lwr_cmd(const double* qdot_lim, const double* q, const double* q_cmded, int n, int direct, double* cmd) {
    for (int i = 0; i < n; ++i) cmd[i] = direct ? qdot_lim[i] : -q_cmded[i] + q[i] + qdot_lim[i];
}

void vfo_state_init(vfo_state* s, int n) {
    memset(s->lastvec, 0, sizeof s->lastvec); /* nullspace:92 */
    for (int i = 0; i < MAXJ; ++i) s->sig[i] = 1; /* nullspace:91 */
    (void)n;
}

/* ------------------------------------------------------------------------------------------
 * one control cycle of one arm
 * ---------------------------------------------------------------------------------------- */
void vfo_cycle(const vfik_chain* c, const vfik_params* p, const double tool[16],
               const vfik_field* fields, int nfields, const double* q,
               const double* null_control, const double* ext_cmd, vfo_state* st, vfo_out* out) {
    const int n = c->n;
    double J[6 * MAXJ], Tee[12], Ttip[12], vec6[6], sc[2], tw[6], qv[MAXJ], qn[MAXJ], qo[MAXJ];
    int status = 0;

    vfo_jacobian(c, q, J, Tee);                       /* vf:316-318 */
    frame_mul(Tee, tool, Ttip);                       /* vf:329-330 (tool16 rows 0..2 are a 3x4) */
    double r[3] = {Tee[3] - Ttip[3], Tee[7] - Ttip[7], Tee[11] - Ttip[11]}; /* diff(new,old).vel vf:331 */
    vfo_field_eval(fields, nfields, Ttip, p->rot_slowdown, vec6, sc); /* vf:344-345 */
    double v[3], w[3], wxr[3];
    for (int k = 0; k < 3; ++k) {                     /* vf:346-347 */
        v[k] = p->speed_scale * sc[0] * vec6[k];
        w[k] = p->speed_scale * sc[1] * vec6[3 + k];
    }
    cross3(w, r, wxr);                                /* Twist.RefPoint(diff.vel) vf:456-459 */
    for (int k = 0; k < 3; ++k) { tw[k] = v[k] + wxr[k]; tw[3 + k] = w[k]; }
    vfo_ikv(J, n, tw, p->wy, p->wq, p->lambda, qv);   /* vf:461 */

    for (int k = 0; k < n; ++k) qn[k] = 0.0;
    if (p->flags & VFIK_F_NULLSPACE) {
        static const double zero4[VFIK_NULL_CONTROLS] = {0, 0, 0, 0}; /* nullspace:137 */
        int rank = 0;
        vfo_move_in_nullspace(J, n, null_control ? null_control : zero4, VFIK_NULL_CONTROLS,
                              st->lastvec, st->sig, qn, &rank);  /* nullspace:175-176 */
        if (rank >= 2) status |= VFIK_ST_NULL_AMBIGUOUS;
        if (p->flags & VFIK_F_JOINT_LIMIT_TASK) {
            double B[MAXJ * MAXJ], z[MAXJ];
            vfo_restrict(J, n, B);
            for (int i = 0; i < n; ++i) {
                double mid = 0.5 * (c->q_lo[i] + c->q_hi[i]), half = 0.5 * (c->q_hi[i] - c->q_lo[i]);
                z[i] = -p->jl_gain * (q[i] - mid) / (half * half);
            }
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += B[i * n + k] * z[k];
                qn[i] += s;
            }
        }
        if (vfo_check_limits(q, qn, c->q_lo, c->q_hi, n, p->lookahead)) status |= VFIK_ST_LIMIT_STOP; /* :178 */
        for (int k = 0; k < n; ++k) qn[k] *= p->null_gain; /* nullspace:183 */
    }

    if (p->flags & VFIK_F_MIXER) {
        double cmd[VFIK_MIX_CHANNELS * MAXJ];
        for (int k = 0; k < n; ++k) { cmd[k] = qv[k]; cmd[n + k] = qn[k]; }
        for (int ch = 2; ch < VFIK_MIX_CHANNELS; ++ch)
            for (int k = 0; k < n; ++k) cmd[ch * n + k] = ext_cmd ? ext_cmd[(ch - 2) * n + k] : 0.0;
        vfo_mix(cmd, p->mix_w, VFIK_MIX_CHANNELS, n, qo);
    } else {
        for (int k = 0; k < n; ++k) qo[k] = qv[k];
    }
    if (p->flags & VFIK_F_LIMITER)
        if (vfo_limiter(qo, n, p->max_vel)) status |= VFIK_ST_LIMITED;
    for (int k = 0; k < n; ++k)
        if (isnan(qo[k])) status |= VFIK_ST_NAN;

    if (out->qdot_vf) memcpy(out->qdot_vf, qv, n * sizeof(double));
    if (out->qdot_null) memcpy(out->qdot_null, qn, n * sizeof(double));
    if (out->qdot_out) memcpy(out->qdot_out, qo, n * sizeof(double));
    static const double last_row[4] = {0, 0, 0, 1};
    if (out->pose) { memcpy(out->pose, Ttip, sizeof Ttip); memcpy(out->pose + 12, last_row, sizeof last_row); }
    if (out->pose_nt) { memcpy(out->pose_nt, Tee, sizeof Tee); memcpy(out->pose_nt + 12, last_row, sizeof last_row); }
    if (out->v6) for (int k = 0; k < 3; ++k) { out->v6[k] = v[k]; out->v6[3 + k] = w[k]; }
    if (out->qdist) vfo_dist_to_center(q, c->q_lo, c->q_hi, n, out->qdist);
    if (out->status) *out->status = status;
}

int vfo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void vfo_cycle_batch(const vfik_chain* c, const vfik_params* p, int B, const double* tool,
                     int tool_stride, const vfik_field* fields, int max_fields, const int* nfields,
                     const double* q, const double* null_control, const double* ext_cmd,
                     vfo_state* st, double* qdot_vf, double* qdot_null, double* qdot_out,
                     double* pose, double* pose_nt, double* v6, double* qdist, int* status,
                     int nthreads, const double* q_lo, const double* q_hi, const int* active) {
    const int n = c->n;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : omp_get_max_threads())
#endif
    for (int b = 0; b < B; ++b) {
        if (active && !active[b]) continue; /* `if qInBottle` (vf:312-313): no fresh joint angles, no cycle */
        vfik_chain cb;                      /* this cycle's limits of this arm (nullspace:167) */
        const vfik_chain* ca = c;
        if (q_lo && q_hi) {
            cb = *c;
            for (int k = 0; k < n; ++k) { cb.q_lo[k] = q_lo[(long)b * n + k]; cb.q_hi[k] = q_hi[(long)b * n + k]; }
            ca = &cb;
        }
        double ext[4 * MAXJ];
        if (ext_cmd)
            for (int ch = 0; ch < 4; ++ch)
                for (int k = 0; k < n; ++k) ext[ch * n + k] = ext_cmd[((long)ch * B + b) * n + k];
        vfo_out o;
        o.qdot_vf = qdot_vf ? qdot_vf + (long)b * n : 0;
        o.qdot_null = qdot_null ? qdot_null + (long)b * n : 0;
        o.qdot_out = qdot_out ? qdot_out + (long)b * n : 0;
        o.pose = pose ? pose + (long)b * 16 : 0;
        o.pose_nt = pose_nt ? pose_nt + (long)b * 16 : 0;
        o.v6 = v6 ? v6 + (long)b * 6 : 0;
        o.qdist = qdist ? qdist + (long)b * n : 0;
        o.status = status ? status + b : 0;
        vfo_cycle(ca, p, tool + (long)b * tool_stride, fields + (long)b * max_fields, nfields[b],
                  q + (long)b * n, null_control ? null_control + (long)b * VFIK_NULL_CONTROLS : 0,
                  ext_cmd ? ext : 0, st ? st + b : 0, &o);
    }
}
