/* selftest.c -- drives every entry point of vfik_oracle.c once; built with AddressSanitizer and
 * UBSan by tests/test_oracle_sanitizers.py (the reference has no sanitizers or tests of its own,
 * SURVEY section 5).  TEST INFRASTRUCTURE. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "vfik_oracle.h"

static void dh(vfik_chain* c, int n) {
    memset(c, 0, sizeof *c);
    c->n = n;
    for (int i = 0; i <= n; ++i) {
        double a = (i % 2) ? 1.5707963267948966 : -1.5707963267948966;
        double B[12] = {1, 0, 0, 0.02 * i, 0, cos(a), -sin(a), 0, 0, sin(a), cos(a), 0.1 + 0.05 * i};
        memcpy(c->B[i], B, sizeof B);
    }
    for (int i = 0; i < n; ++i) { c->jtype[i] = (i == 3); c->q_lo[i] = -2.5; c->q_hi[i] = 2.5; }
}

int main(void) {
    int fails = 0;
    for (int n = 1; n <= VFIK_MAX_JOINTS; ++n) {
        vfik_chain c;
        dh(&c, n);
        vfik_params p;
        memset(&p, 0, sizeof p);
        p.speed_scale = 1; p.lambda = 0.1; p.rot_slowdown = 0.3; p.null_gain = 0.5; p.lookahead = 0.3; p.jl_gain = 0.5; p.max_vel = 0.7;
        for (int i = 0; i < 6; ++i) p.wy[i] = 1;
        for (int i = 0; i < VFIK_MAX_JOINTS; ++i) p.wq[i] = 1;
        p.mix_w[0] = p.mix_w[1] = 1; p.mix_w[2] = 0.5;
        p.flags = VFIK_F_NULLSPACE | VFIK_F_JOINT_LIMIT_TASK | VFIK_F_MIXER | VFIK_F_LIMITER;
        vfik_field f[5];
        memset(f, 0, sizeof f);
        f[0].id = 1; f[0].type = VFIK_FIELD_ATTRACTOR; f[0].force = 1;
        double G[16] = {0, 1, 0, 0.3, -1, 0, 0, 0.2, 0, 0, 1, 0.6, 0, 0, 0, 1};
        memcpy(f[0].p, G, sizeof G); f[0].p[16] = 0.05;
        f[1].id = 4; f[1].type = VFIK_FIELD_REPELLER; f[1].force = -10;
        double r[6] = {0.2, 0.1, 0.5, 0.05, 0.001, 5}; memcpy(f[1].p, r, sizeof r);
        f[2].id = 5; f[2].type = VFIK_FIELD_HEMISPHERE; f[2].force = -50;
        double hm[8] = {0, 0, -0.3, 0, 0, 1, 0.05, 5}; memcpy(f[2].p, hm, sizeof hm);
        f[3].id = 2; f[3].type = VFIK_FIELD_FUNNEL; f[3].force = 30;
        double fu[10] = {0.3, 0.2, 0.6, 0, -1, 0, 0.15, 10, 0.15, 2}; memcpy(f[3].p, fu, sizeof fu);
        f[4].id = 9; f[4].type = VFIK_FIELD_NULL; f[4].force = 1;
        double q[VFIK_MAX_JOINTS], ext[4 * VFIK_MAX_JOINTS], ctrl[4] = {0.3, -0.2, 0.1, 0};
        for (int i = 0; i < n; ++i) q[i] = 0.3 * sin(1.0 + i);
        for (int i = 0; i < 4 * n; ++i) ext[i] = 0.01 * i;
        double tool[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.2, 0, 0, 0, 1};
        double qv[VFIK_MAX_JOINTS], qn[VFIK_MAX_JOINTS], qo[VFIK_MAX_JOINTS], pose[16], pnt[16], v6[6], qd[VFIK_MAX_JOINTS];
        int st = 0;
        vfo_state s;
        vfo_state_init(&s, n);
        vfo_out o = {qv, qn, qo, pose, pnt, v6, qd, &st};
        for (int t = 0; t < 3; ++t) {
            vfo_cycle(&c, &p, tool, f, 5, q, ctrl, ext, &s, &o);
            for (int i = 0; i < n; ++i) {
                if (!isfinite(qo[i]) || fabs(qo[i]) > 0.7 + 1e-12) { printf("n=%d: bad qdot %g\n", n, qo[i]); ++fails; }
                q[i] += 0.01 * qo[i];
            }
        }
        double J[6 * VFIK_MAX_JOINTS], T[12], Bm[VFIK_MAX_JOINTS * VFIK_MAX_JOINTS];
        vfo_jacobian(&c, q, J, T);
        vfo_restrict(J, n, Bm);
        double tr = 0;
        for (int i = 0; i < n; ++i) tr += Bm[i * n + i];
        if (fabs(tr - (n > 6 ? n - 6 : 0)) > 1e-6) { printf("n=%d: trace of the projector %g\n", n, tr); ++fails; }
    }
    /* batch driver with every optional output absent */
    vfik_chain c;
    dh(&c, 7);
    vfik_params p;
    memset(&p, 0, sizeof p);
    p.speed_scale = 1; p.lambda = 0.1; p.rot_slowdown = 0.3;
    for (int i = 0; i < 6; ++i) p.wy[i] = 1;
    for (int i = 0; i < VFIK_MAX_JOINTS; ++i) p.wq[i] = 1;
    double q[3 * 7] = {0}, tool[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, out[3 * 7];
    vfik_field none[3];
    memset(none, 0, sizeof none);
    int cnt[3] = {0, 0, 0};
    vfo_cycle_batch(&c, &p, 3, tool, 0, none, 1, cnt, q, 0, 0, 0, 0, 0, out, 0, 0, 0, 0, 0, 2, 0, 0, 0);
    for (int i = 0; i < 21; ++i) if (out[i] != 0.0) { printf("empty field set moved\n"); ++fails; }
    printf(fails ? "FAILED %d\n" : "selftest OK\n", fails);
    return fails != 0;
}
