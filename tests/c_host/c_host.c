/* A host written in plain C99 against include/vfik.h only -- no Python, no torch: builds a 7-DOF chain
 * from its DH table, gives every arm a goal and three point obstacles, runs one control cycle through
 * vfik_step_host and a 50-cycle rollout through the device-pointer API, and checks both against the
 * CPU oracle (oracle/vfik_oracle.h -- test infrastructure, linked only by this test program).
 * Exit code 0 = parity within 1e-9 rad/s (float64 I/O).  Built and run by tests/test_gpu_c_host.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vfik.h"
#include "vfik_oracle.h"

#define NJ 7
#define NB 1000 /* arms: not a multiple of the wave size on purpose */
#define NF 4    /* goal + 3 obstacles */

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand(double lo, double hi) { /* xorshift64*, 53-bit mantissa */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return lo + (hi - lo) * (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}

/* standard DH row (a, alpha, d): fixed part Tz(d) Tx(a) Rx(alpha) as a row-major 3x4 */
static void dh_fixed(double a, double alpha, double d, double B[12]) {
    const double ca = cos(alpha), sa = sin(alpha);
    const double M[12] = {1, 0, 0, a, 0, ca, -sa, 0, 0, sa, ca, d};
    memcpy(B, M, sizeof M);
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vfik_last_error()); return 2; } \
    } while (0)

int main(void) {
    static const double dh[NJ][3] = {{0, M_PI / 2, 0.31}, {0, -M_PI / 2, 0}, {0, -M_PI / 2, 0.4}, {0, M_PI / 2, 0},
                                     {0, M_PI / 2, 0.39}, {0, -M_PI / 2, 0}, {0, 0, 0.078}};
    static const double lim_deg[NJ] = {170, 120, 170, 120, 170, 120, 170};
    vfik_chain chain;
    memset(&chain, 0, sizeof chain);
    chain.n = NJ;
    const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(chain.B[0], ident, sizeof ident);
    for (int i = 0; i < NJ; ++i) {
        dh_fixed(dh[i][0], dh[i][1], dh[i][2], chain.B[i + 1]);
        chain.q_hi[i] = lim_deg[i] * M_PI / 180.0;
        chain.q_lo[i] = -chain.q_hi[i];
    }

    size_t sizes[4];
    vfik_struct_sizes(sizes);
    if (sizes[0] != sizeof(vfik_field) || sizes[1] != sizeof(vfik_chain) || sizes[2] != sizeof(vfik_params) || sizes[3] != sizeof(vfik_io)) {
        fprintf(stderr, "struct layout mismatch between this C compiler and the library\n");
        return 2;
    }
    if (vfik_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 3; }
    vfik_handle* h = vfik_create(0, 64, NJ, 8, NB);
    if (!h) { fprintf(stderr, "vfik_create: %s\n", vfik_last_error()); return 2; }
    CHECK(vfik_set_chain(h, &chain));

    vfik_params p;
    memset(&p, 0, sizeof p);
    p.speed_scale = 1.0; p.lambda = 0.1; p.rot_slowdown = 0.3; p.null_gain = 0.5; p.lookahead = 0.3;
    p.jl_gain = 0.5; p.max_vel = 0.6; p.jp_kp = 1.5; p.jp_delta = 0.087;
    for (int i = 0; i < 6; ++i) p.wy[i] = 1.0;
    for (int i = 0; i < VFIK_MAX_JOINTS; ++i) p.wq[i] = 1.0;
    p.mix_w[0] = p.mix_w[1] = 1.0;
    p.flags = VFIK_F_NULLSPACE | VFIK_F_MIXER | VFIK_F_LIMITER;
    CHECK(vfik_set_params(h, &p));

    double* q = malloc(sizeof(double) * NB * NJ);
    double* ctrl = malloc(sizeof(double) * NB * VFIK_NULL_CONTROLS);
    vfik_field* fields = calloc((size_t)NB * NF, sizeof(vfik_field));
    int32_t* counts = malloc(sizeof(int32_t) * NB);
    double *got = malloc(sizeof(double) * NB * NJ), *ref = malloc(sizeof(double) * NB * NJ);
    double *pose = malloc(sizeof(double) * NB * 16), *pose_ref = malloc(sizeof(double) * NB * 16);
    int32_t *st = malloc(sizeof(int32_t) * NB), *st_ref = malloc(sizeof(int32_t) * NB);
    const double tool[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int b = 0; b < NB; ++b) {
        double qg[NJ], J[6 * VFIK_MAX_JOINTS], T[12];
        for (int i = 0; i < NJ; ++i) { q[b * NJ + i] = urand(0.8 * chain.q_lo[i], 0.8 * chain.q_hi[i]); qg[i] = urand(0.8 * chain.q_lo[i], 0.8 * chain.q_hi[i]); }
        for (int k = 0; k < VFIK_NULL_CONTROLS; ++k) ctrl[b * VFIK_NULL_CONTROLS + k] = urand(-1, 1);
        vfo_jacobian(&chain, qg, J, T); /* a reachable goal: FK of another configuration */
        vfik_field* f = fields + (size_t)b * NF;
        f[0].id = 1; f[0].type = VFIK_FIELD_ATTRACTOR; f[0].force = 1.0;
        memcpy(f[0].p, T, sizeof T);
        f[0].p[15] = 1.0; f[0].p[16] = 0.05;
        for (int k = 1; k < NF; ++k) {
            f[k].id = 4 + k; f[k].type = VFIK_FIELD_REPELLER; f[k].force = -10.0;
            f[k].p[0] = urand(-0.8, 0.8); f[k].p[1] = urand(-0.8, 0.8); f[k].p[2] = urand(0, 1.2);
            f[k].p[3] = urand(0.03, 0.1); f[k].p[4] = 0.001; f[k].p[5] = 5.0;
        }
        counts[b] = NF;
    }
    CHECK(vfik_set_fields(h, 0, NB, fields, NF, counts));

    /* one cycle, host pointers */
    vfik_io io;
    memset(&io, 0, sizeof io);
    io.q = q; io.null_control = ctrl; io.qdot_out = got; io.pose = pose; io.status = st;
    CHECK(vfik_step_host(h, &io));
    vfo_state* states = malloc(sizeof(vfo_state) * NB);
    for (int b = 0; b < NB; ++b) vfo_state_init(&states[b], NJ);
    vfo_cycle_batch(&chain, &p, NB, tool, 0, fields, NF, counts, q, ctrl, NULL, states, NULL, NULL, ref, pose_ref, NULL, NULL, NULL, st_ref, 0, NULL, NULL, NULL);
    double worst = 0.0, worst_pose = 0.0;
    int bad_status = 0;
    for (int k = 0; k < NB * NJ; ++k) worst = fmax(worst, fabs(got[k] - ref[k]));
    for (int k = 0; k < NB * 16; ++k) worst_pose = fmax(worst_pose, fabs(pose[k] - pose_ref[k]));
    for (int b = 0; b < NB; ++b) bad_status += st[b] != st_ref[b];
    printf("step:    max |qdot - oracle| = %.3e rad/s, max |pose - oracle| = %.3e, status mismatches %d\n", worst, worst_pose, bad_status);
    int fail = !(worst < 1e-9) || !(worst_pose < 1e-12) || bad_status;

    /* 50-cycle rollout, device pointers from the library's own allocator */
    const int K = 50;
    const double dt = 0.004;
    CHECK(vfik_reset_state(h));
    void* dq = vfik_dev_alloc(h, sizeof(double) * NB * NJ);
    void* dqd = vfik_dev_alloc(h, sizeof(double) * NB * NJ);
    void* dqe = vfik_dev_alloc(h, sizeof(double) * NB * NJ);
    if (!dq || !dqd || !dqe) { fprintf(stderr, "vfik_dev_alloc: %s\n", vfik_last_error()); return 2; }
    CHECK(vfik_memcpy_h2d(h, dq, q, sizeof(double) * NB * NJ));
    vfik_io dio;
    memset(&dio, 0, sizeof dio);
    dio.q = dq; dio.qdot_out = dqd;
    CHECK(vfik_rollout(h, &dio, K, dt, 1, dqe));
    CHECK(vfik_sync(h));
    double* q_end = malloc(sizeof(double) * NB * NJ);
    CHECK(vfik_memcpy_d2h(h, q_end, dqe, sizeof(double) * NB * NJ));
    double* qs = malloc(sizeof(double) * NB * NJ);
    memcpy(qs, q, sizeof(double) * NB * NJ);
    for (int b = 0; b < NB; ++b) vfo_state_init(&states[b], NJ);
    for (int t = 0; t < K; ++t) { /* the oracle stepped on the host, joint_sim integration in between */
        vfo_cycle_batch(&chain, &p, NB, tool, 0, fields, NF, counts, qs, NULL, NULL, states, NULL, NULL, ref, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL);
        for (int b = 0; b < NB; ++b)
            for (int i = 0; i < NJ; ++i) {
                double v = fma(dt, ref[b * NJ + i], qs[b * NJ + i]);
                qs[b * NJ + i] = fmin(fmax(v, chain.q_lo[i]), chain.q_hi[i]);
            }
    }
    double worst_q = 0.0;
    for (int k = 0; k < NB * NJ; ++k) worst_q = fmax(worst_q, fabs(q_end[k] - qs[k]));
    printf("rollout: %d cycles, max |q - oracle| = %.3e rad\n", K, worst_q);
    fail |= !(worst_q < 1e-8);

    CHECK(vfik_dev_free(h, dq)); CHECK(vfik_dev_free(h, dqd)); CHECK(vfik_dev_free(h, dqe));
    vfik_destroy(h);
    puts(fail ? "c_host FAILED" : "c_host OK");
    return fail;
}
