// vfik_abi.cpp -- host side of the C-ABI declared in include/vfik.h: handle, device state, field
// packing, launches.  Compiled by hipcc into libvfik_hip.so together with vfik_kernel.hip.
// There is no CPU execution path in this file: every compute entry point launches a HIP kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <limits>
#include <vector>

#include "../../include/vfik.h"
#include "vfik_kernel.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(VFIK_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// number of device slots a field of this type occupies (goal block excluded)
int slots_of(int type) {
    switch (type) {
        case VFIK_FIELD_NULL: return 0;
        case VFIK_FIELD_REPELLER: return 1;
        case VFIK_FIELD_HEMISPHERE: return 2;
        case VFIK_FIELD_FUNNEL: return 2;
        case VFIK_FIELD_ATTRACTOR: return 3;
        default: return -1;
    }
}

}  // namespace

struct vfik_handle {
    int device = 0, io_dtype = 32, n = 0, max_slots = 0, B = 0, Bpad = 0, block = 64;
    // eight-lanes-per-arm kernel for lean launches of batches up to this size (vfik_set_small_batch_kernel,
    // VFIK_SUB8_MAX_BATCH).  Measured crossover (tools/ab_mapping.py, float64 I/O, goal + 4 repellers): 3-7 % faster than
    // one lane per arm up to 4 096 arms, equal at 8 192, 1.4x / 2.0x / 2.8x SLOWER at 16 384 / 32 768 / 65 536.
    // Round 4 re-decided the first two (profiles/r04_latency_small_*.txt): the lane-per-arm lean kernels lost a sixth of their instructions and
    // enter through 56 bytes of scalar arguments, the eight-lanes kernel through the 360-byte block -- back to back it is the host's enqueue that
    // sets its pace (3.5 against 4.9-6.1 us per lean launch at every size), and with one synchronisation per cycle the two are within 3 %.
    int sub8_max_batch = 0;          // q -> qdot_out without the module: one lane per arm at every size (4096 until round 3)
    int sub8_max_batch_ns = 0;       // with the nullspace module and qdot_out only: likewise (32 until round 3)
    int sub8_max_batch_full = 4096;  // launches that publish the per-cycle rows: -4 ... -10 % at every size either way (vfik_set_small_batch_kernel sets all three)
    long sub8_launches = 0;  // how many launches took it (introspection for tests / A/B)
    long epoch = 0;          // moves with every call that can change what a launch bakes in (vfik_launch_epoch)
    int n_simd = 1024;       // 4 per CU of this device
    // Batches beyond one wave per SIMD may take the persistent launch (cycle_kernel PERS; VFIK_PERSISTENT=1).  Off by default:
    // same-box A/B, lean C3 launches, rounds vs persistent: 131 072 arms 10.37 / 10.71 us, 262 144 20.47 / 20.56, 524 288
    // 41.3 / 39.5 -- the wave is bound by its float64 instruction issue, not by the waits the prefetch removes
    // (profiles/r03_batch_scaling.txt).
    int pers = 0;
    // What pays there instead: the lean kernels compiled for two waves per SIMD (cycle_kernel WAVES = 2; float I/O, chains of up to 7
    // joints): 131 072 arms 10.4 -> 9.1 us, 524 288 38.4 -> 33.4 (same file).  VFIK_TWO_WAVES=0 keeps the rounds of one wave per SIMD.
    int waves2 = 1;
    size_t esz = 4;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    vfik_chain chain{};
    bool chain_set = false;
    vfik_params params{};
    // device state
    void* d_arena = nullptr;   // [goal | kconst | lastvec | slots_fast | slots]: d_goal, d_kconst, d_lastvec, d_slots_fast, d_slots point into it
    void* d_goal = nullptr;    // 4 quad planes
    void* d_funnel = nullptr;  // aux block, 6 quad planes: the arm's funnel attractor (3) and hemisphere repeller (3) on the straight-line path (vfik_kernel.h)
    void* d_slots = nullptr;   // 2*S quad planes
    void* d_slots_uni = nullptr;   // uniform repeller image: ONE quad plane per slot (x y z radius); read when every decay repeller of the batch shares safe distance and force
    std::vector<char> arm_pair_state;     // per arm: 0 no decay repeller, 1 all of them share one (safe, force), 2 mixed
    std::vector<double> arm_safe, arm_force;
    int uni_allowed = 1;                  // VFIK_UNIFORM_IMAGE=0: always the compact image (tests, A/B)
    int mixed_allowed = 1;                // VFIK_MIXED_ORDERS=0: differing integer orders take the general path, as until round 3 (tests, A/B)
    int uni_ok = 0;                       // the batch's decay repellers share one pair, now in the device constants (KConst::rep_safe, dh[0].pad)
    double uni_safe = 0.0, uni_force = 0.0;
    void* d_slots_fast = nullptr;  // compact repeller image for the straight-line path: 3 quad planes per PAIR of slots
    void* d_orders = nullptr;      // order planes: 16 bytes per arm and plane, one byte per compact-image slot (vfik_kernel.h); read when `mixed`
    int mixed = 0;                 // the batch's decay repellers have integer orders that differ (between slots or between arms)
    void* d_tool = nullptr;    // 3 quad planes (per-arm tools only)
    double tool_shared[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int tool_per_arm = 0;
    void* d_ext = nullptr;     // [4][B][n], allocated on first use
    float* d_lastvec = nullptr;  // nullspace sign memory, [(n + 4) / 4][Bpad][4] floats (vfik_kernel.h)
    double* d_mixw = nullptr;  // [16]
    unsigned long long* d_stamps = nullptr;  // diagnostic build only
    void* d_rollq[2] = {nullptr, nullptr};  // q ping-pong of the stepped rollout (long chains)
    double* d_wts = nullptr;    // per-arm IK weights [6 + n][Bpad], allocated by vfik_set_arm_weights
    // every arm's bridge state (mixer weights, limiter speed) equal: the launch reads them from the batch constants like a handle that
    // never had per-arm bridge state (a.mixw stays NULL, so the lean / publishing-lean variants keep serving it) -- what a port-level caller
    // produces when every arm's handler sends the same /bridge/weight (handlers.py:189-204,481-497)
    bool bridge_uniform = false;
    double bridge_u[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ... the common row, as the per-arm image holds it (rounded to the I/O type)
    double* d_track = nullptr;  // tracking-error history [38][B], allocated on first use
    void* d_mixw_arm = nullptr;  // per-arm mixer weights (2 quad planes), allocated on first use
    void* d_kconst = nullptr;  // vfik::KConst<n>: chain + parameters, read through the scalar cache
    // observers fused into the cycle call (ABI 4): object frames of the distance monitor [B][n_objects][16] in the io
    // dtype, and the cycle's pose / field twist when the caller did not ask for them itself
    void* d_objects = nullptr;
    int n_objects = 0;
    void* d_obs_pose = nullptr;
    void* d_obs_v6 = nullptr;
    size_t dev_bytes = 0;
    // host bookkeeping
    std::vector<double> bridge_host;  // [B][8] mirror of d_mixw_arm: mixer weights 0..5, max_vel 6
    std::vector<int> slots_per_arm;
    std::vector<int> fast_slots_per_arm;  // slots of the compact repeller image per arm (the funnel is not one)
    std::vector<char> arm_has_funnel;
    int slots_used_fast = 0;
    int any_funnel = 0;
    std::vector<int> arm_order;  // per arm: -1 no repellers, n >= 0 all slots are repellers of integer order n, -2 general
    int slots_used = 0;
    int fast_order = 0;
    int plain = 0;  // the chain allows the PLAIN kernel variants: 1 + (shared tool ? 1 : 0) + (shared IK weights other than one ? 2 : 0); 0: the general variants
    int dhp = 0;    // ... and the chain matches a DH pattern the lean kernels are built for (vfik_kernel.h: DhPattern)
    int dhp_allowed = 1;   // VFIK_DH_PATTERN=0: always the general DH form (tests, A/B)
    bool speed_set = false;
    // vfik_step_host / vfik_rollout_host: one device arena + one pinned host arena for every member of the call
    void* arena_dev = nullptr;
    void* arena_host = nullptr;
    void* arena_host_dev = nullptr;   // the device's address of the pinned host arena (zero-copy calls)
    size_t arena_bytes = 0;
    size_t zero_copy_max = (size_t)64 << 10;   // calls of at most this many bytes: the kernel reads / writes the pinned arena itself (VFIK_ZERO_COPY_MAX)
    struct Scratch { void* p = nullptr; size_t bytes = 0; };
    Scratch sc[20];  // ... and per-member device buffers for calls of few large members
    // pipelined host path (vfik_submit_host / vfik_wait): up to PIPE submissions in flight, each slot with
    // its own device staging buffers and events; s_in / s_out are the side streams
    static constexpr int PIPE = 3;
    struct PipeSlot {
        Scratch sc[20];
        hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
        long ticket = -1;  // submission living in this slot, -1 = free
    };
    PipeSlot pipe[PIPE];
    hipStream_t s_in = nullptr, s_out = nullptr;
    long next_ticket = 0;
};

namespace {

int dev_alloc(vfik_handle* h, void** p, size_t bytes, bool zero) {
    HIP_TRY(hipMalloc(p, bytes));
    h->dev_bytes += bytes;
    if (zero) HIP_TRY(hipMemsetAsync(*p, 0, bytes, h->stream));
    return VFIK_OK;
}

// A setter rewrites device state that kernels of the pipelined host path may still be reading on the
// side streams: let those drain first (the handle's own stream is synchronised by the setters themselves).
// (every vfik_set_* / vfik_reset_state call passes through here: the launch epoch moves with them, vfik_launch_epoch)
int quiesce(vfik_handle* h) {
    ++h->epoch;
    if (h->s_in) HIP_TRY(hipStreamSynchronize(h->s_in));
    if (h->s_out) HIP_TRY(hipStreamSynchronize(h->s_out));
    return VFIK_OK;
}

// true when the GPU can dereference p: device memory, or pinned / registered host memory
bool gpu_visible(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory: not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

template <typename T>
void put(std::vector<char>& buf, size_t idx, double v) {
    reinterpret_cast<T*>(buf.data())[idx] = static_cast<T>(v);
}

// Pack the field sets of n_arms arms into quad-plane staging images (plane P, arm j, component c ->
// (P * n_arms + j) * 4 + c): goal = 4 planes, slots = 2*S planes (vfik_kernel.h).
template <typename T>
void pack_fields(const vfik_field* fields, int max_fields, const int32_t* counts, int n_arms, int S,
                 std::vector<char>& goal, std::vector<char>& slots, std::vector<char>& fast, std::vector<int>& used,
                 std::vector<char>& funnel, std::vector<int>& used_fast, std::vector<char>& has_funnel,
                 std::vector<char>& uni, std::vector<char>& pair_state, std::vector<double>& pair_safe, std::vector<double>& pair_force,
                 std::vector<unsigned char>& ord) {
    funnel.assign((size_t)6 * n_arms * 4 * sizeof(T), 0);  // aux block: funnel planes 0..2, hemisphere planes 3..5
    goal.assign((size_t)4 * n_arms * 4 * sizeof(T), 0);
    slots.assign((size_t)std::max(1, 2 * S) * n_arms * 4 * sizeof(T), 0);
    // compact image: a decay repeller needs 6 of its slot's 8 scalars (x y z radius safe | force; the decay order is
    // one number for the batch on the straight-line path and the type is known), so two slots share three quads:
    // (x0 y0 z0 r0 | s0 f0 x1 y1 | z1 r1 s1 f1).  Slots of another type leave zeros (force 0): they force the general path.
    fast.assign((size_t)3 * ((std::max(1, S) + 1) / 2) * n_arms * 4 * sizeof(T), 0);
    // uniform image: when every decay repeller of the BATCH has the same safe distance and force (what the object feeder sends:
    // 0.001 and -10, object_feeder:301-302,323,331) a slot is one quad (x y z radius) and the pair lives in the constants
    uni.assign((size_t)std::max(1, S) * n_arms * 4 * sizeof(T), 0);
    // (the force being the batch's, a slot an arm does not use cannot carry force 0 as in the other images: its radius is -inf)
    for (size_t q4 = 0; q4 < (size_t)std::max(1, S) * n_arms; ++q4) put<T>(uni, q4 * 4 + 3, -std::numeric_limits<double>::infinity());
    // order planes: byte (m % 16) of plane (m / 16) = the integer decay order of compact-image slot m; slots an arm does not use
    // repeat the order of its last repeller (5 -- what the feeder sends, object_feeder:302,333 -- for an arm without any)
    ord.assign((size_t)((std::max(1, S) + 15) / 16) * n_arms * 16, 5);
    std::vector<int> order;
    for (int j = 0; j < n_arms; ++j) {
        const vfik_field* f = fields + (size_t)j * max_fields;
        order.resize(counts[j]);
        for (int k = 0; k < counts[j]; ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return f[a].id < f[b].id; });
        bool have_goal = false, have_funnel = false, have_hemi = false;
        pair_state[j] = 0; pair_safe[j] = 0.0; pair_force[j] = 0.0;
        int m = 0, mr = 0;  // general slots used; compact-image slots used (repellers only, packed densely)
        auto gq = [&](int e) { return ((size_t)(e >> 2) * n_arms + j) * 4 + (e & 3); };
        for (int k : order) {
            const vfik_field& fd = f[k];
            if (fd.type == VFIK_FIELD_NULL) continue;
            if (fd.type == VFIK_FIELD_ATTRACTOR && !have_goal) {
                have_goal = true;
                for (int e = 0; e < 12; ++e) put<T>(goal, gq(e), fd.p[e]);
                put<T>(goal, gq(12), 1.0);  // "goal present"
                put<T>(goal, gq(13), fd.p[16]);
                put<T>(goal, gq(14), fd.force);
                continue;
            }
            const int ns = slots_of(fd.type);
            auto at = [&](int slot, int e) { return ((size_t)(2 * (m + slot) + (e >> 2)) * n_arms + j) * 4 + (e & 3); };
            for (int e = 0; e < 6; ++e) put<T>(slots, at(0, e), fd.p[e]);
            put<T>(slots, at(0, 6), fd.force);
            put<T>(slots, at(0, 7), (double)fd.type);
            if (fd.type == VFIK_FIELD_FUNNEL && !have_funnel) {
                // the straight-line path's funnel block (used only when this funnel is the arm's one non-repeller entry)
                have_funnel = true;
                const double blk[12] = {fd.p[0], fd.p[1], fd.p[2], fd.p[3], fd.p[4], fd.p[5], fd.p[6], fd.p[7], fd.p[8], fd.p[9], fd.force, 1.0};
                for (int e = 0; e < 12; ++e) put<T>(funnel, ((size_t)(e >> 2) * n_arms + j) * 4 + (e & 3), blk[e]);
            }
            if (fd.type == VFIK_FIELD_HEMISPHERE && !have_hemi) {
                // ... and the hemisphere block (object_feeder:344-353, ObstacleH: a surface with its normal): x y z nx | ny nz safe order | force present
                have_hemi = true;
                const double blk[12] = {fd.p[0], fd.p[1], fd.p[2], fd.p[3], fd.p[4], fd.p[5], fd.p[6], fd.p[7], fd.force, 1.0, 0.0, 0.0};
                for (int e = 0; e < 12; ++e) put<T>(funnel, ((size_t)(3 + (e >> 2)) * n_arms + j) * 4 + (e & 3), blk[e]);
            }
            if (fd.type == VFIK_FIELD_REPELLER) {
                for (int e = 0; e < 4; ++e) put<T>(uni, ((size_t)mr * n_arms + j) * 4 + e, fd.p[e]);
                // (compared as the device will see them: rounded to the I/O type)
                const double sv = (double)static_cast<T>(fd.p[4]), fv = (double)static_cast<T>(fd.force);
                if (pair_state[j] == 0) { pair_state[j] = 1; pair_safe[j] = sv; pair_force[j] = fv; }
                else if (pair_safe[j] != sv || pair_force[j] != fv) pair_state[j] = 2;
                // (the uniform image's kernel clamps radius + safe at 0 -- how it disarms unused slots: a repeller with a NEGATIVE sum
                // keeps the arm, hence the batch, on the compact image)
                if ((double)static_cast<T>(fd.p[3]) + sv < 0.0) pair_state[j] = 2;
                {   // (non-integer or large orders send the batch to the general path: the byte is not read then)
                    const double o = fd.p[5];
                    ord[((size_t)(mr >> 4) * n_arms + j) * 16 + (mr & 15)] = (o >= 0.0 && o < 128.0 && (double)(int)o == o) ? (unsigned char)(int)o : 0;
                }
                const int pair = mr >> 1, half = mr & 1;
                ++mr;
                for (int i = 0; i < 6; ++i) {
                    const int e = 6 * half + i;  // position in the pair's 12 scalars
                    put<T>(fast, ((size_t)(3 * pair + (e >> 2)) * n_arms + j) * 4 + (e & 3), i < 5 ? fd.p[i] : fd.force);
                }
            }
            for (int c = 1; c < ns; ++c) {
                for (int e = 0; e < 6; ++e) {
                    const int pi = 6 * c + e;
                    put<T>(slots, at(c, e), pi < VFIK_MAX_PARAMS ? fd.p[pi] : 0.0);
                }
                put<T>(slots, at(c, 7), -1.0);
            }
            m += ns;
        }
        if (mr > 0) {   // the slots behind the last repeller repeat ITS order: a partly filled chunk adds no distinct order of its own
            const size_t np = ord.size() / ((size_t)n_arms * 16);
            const unsigned char last = ord[((size_t)((mr - 1) >> 4) * n_arms + j) * 16 + ((mr - 1) & 15)];
            for (size_t ms = mr; ms < np * 16; ++ms) ord[((ms >> 4) * n_arms + j) * 16 + (ms & 15)] = last;
        }
        used[j] = m;
        used_fast[j] = mr;
        has_funnel[j] = (have_funnel || have_hemi) ? 1 : 0;
    }
}

void fill_kargs(const vfik_handle* h, const vfik_io* io, vfik::KArgs& a) {
    std::memset(&a, 0, sizeof a);
    a.B = h->B;
    a.Bpad = h->Bpad;
    a.slots_used = h->slots_used;
    a.fast_order = h->fast_order;
    a.flags = h->params.flags;
    a.tool_stride = h->tool_per_arm ? h->Bpad : 0;
    a.plain = (h->plain && !h->tool_per_arm && !h->d_wts) ? h->plain : 0;   // (1 ... 4: the PLAIN kernels, with the batch's shared tool / shared IK weights as vfik_kernel.h says)
    a.wts = h->d_wts;
    a.q = io->q;
    a.goal = h->d_goal;
    a.slots = h->d_slots;
    a.slots_fast = h->d_slots_fast;
    a.tool = h->d_tool;
    a.null_control = io->null_control;
    a.ext = h->d_ext;
    a.mixw = h->bridge_uniform ? nullptr : h->d_mixw_arm;
    a.q_ref = io->q_ref;
    a.q_cmded = io->q_cmded;
    a.lastvec = h->d_lastvec;
    a.qdot_vf = io->qdot_vf;
    a.qdot_null = io->qdot_null;
    a.qdot_out = io->qdot_out;
    a.pose = io->pose;
    a.pose_nt = io->pose_nt;
    a.v6 = io->v6;
    a.qdist = io->qdist;
    a.status = io->status;
    a.goal_dist = io->goal_dist;
    a.active = io->active;
    a.q_lo = io->q_lo;
    a.q_hi = io->q_hi;
    a.q_ref_out = io->q_ref ? io->q_ref_out : nullptr;
    a.sub8_max_batch = h->sub8_max_batch;
    a.sub8_max_batch_full = h->sub8_max_batch_full;
    a.sub8_max_batch_ns = h->sub8_max_batch_ns;
    a.n_simd = h->n_simd;
    a.arena = h->d_arena;
    a.funnel = h->d_funnel;
    a.slots_used_fast = h->slots_used_fast;
    a.has_funnel = h->any_funnel;
    a.pers = h->pers;
    a.waves2 = h->waves2;
    a.uni = h->uni_ok && h->uni_allowed;
    a.uni_planes = 3 * ((std::max(1, h->max_slots) + 1) / 2);
    a.stamps = h->d_stamps;
    a.kc = h->d_kconst;
    a.orders = h->d_orders;
    a.mixed = h->mixed;
    a.dhp = a.plain ? h->dhp : 0;
}

// rewrite the device copy of the batch constants (chain + parameters); rare, synchronous
int upload_kconst(vfik_handle* h) {
    if (!h->chain_set) return VFIK_OK;
    std::vector<char> img(vfik::kconst_bytes(h->n));
    int plain = 0, dhp = 0;
    vfik_params kp = h->params;
    if (h->bridge_uniform) {   // (the arms' common bridge state stands in for the batch-wide values)
        for (int k = 0; k < VFIK_MIX_CHANNELS; ++k) kp.mix_w[k] = h->bridge_u[k];
        kp.max_vel = h->bridge_u[6];
    }
    const double err = vfik::kconst_fill(h->n, img.data(), h->chain, kp, h->tool_shared, &plain, &dhp);
    if (!(err < 1e-9)) return fail(VFIK_E_ARG, "chain: a fixed transform is not a rigid motion (DH recomposition error %.3e)", err);
    std::memcpy(img.data() + VFIK_KCONST_REP_SAFE_OFF(h->n), &h->uni_safe, sizeof(double));   // the batch's uniform repeller pair (vfik_set_fields)
    std::memcpy(img.data() + VFIK_KCONST_REP_FORCE_OFF, &h->uni_force, sizeof(double));
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->d_kconst, img.data(), img.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->plain = plain;
    h->dhp = h->dhp_allowed ? dhp : 0;
    return VFIK_OK;
}

// which arms qualify for the kernel's straight-line repeller path
// (-1: no repeller; n >= 0: every repeller has integer order n; -3: integer orders that differ; -2: the general path)
int classify_arm(const vfik_field* f, int count) {
    int order = -1;
    bool goal = false, funnel = false, hemi = false, mixed = false;
    for (int k = 0; k < count; ++k) {
        const vfik_field& fd = f[k];
        if (fd.type == VFIK_FIELD_NULL) continue;
        if (fd.type == VFIK_FIELD_ATTRACTOR && !goal) { goal = true; continue; }
        if (fd.type == VFIK_FIELD_FUNNEL && !funnel) {
            // one funnel attractor with small integer decay orders (object_feeder:277,279 sends 10 and 2): the straight-line
            // path evaluates it from its own block
            funnel = true;
            const double oa = fd.p[7], od = fd.p[9];
            if (!((double)(int)oa == oa) || oa < 0 || oa >= 128 || !((double)(int)od == od) || od < 0 || od >= 128) return -2;
            continue;
        }
        if (fd.type == VFIK_FIELD_HEMISPHERE && !hemi) {  // one hemisphere repeller with a small integer decay order (object_feeder:353 sends 5)
            hemi = true;
            const double oh = fd.p[7];
            if (!((double)(int)oh == oh) || oh < 0 || oh >= 128) return -2;
            continue;
        }
        if (fd.type != VFIK_FIELD_REPELLER) return -2;
        const double o = fd.p[5];
        const int n = (int)o;
        if (!((double)n == o) || n < 0 || n >= 128) return -2;
        if (order >= 0 && n != order) mixed = true;
        order = n;
    }
    return mixed ? -3 : order;
}

int check_handle(const vfik_handle* h) {
    if (!h) return fail(VFIK_E_ARG, "null handle");
    return VFIK_OK;
}

}  // namespace

extern "C" {

int vfik_abi_version(void) { return VFIK_ABI_VERSION; }

void vfik_struct_sizes(size_t out[4]) {
    out[0] = sizeof(vfik_field); out[1] = sizeof(vfik_chain); out[2] = sizeof(vfik_params); out[3] = sizeof(vfik_io);
}

const char* vfik_last_error(void) { return g_err.c_str(); }

int vfik_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

uint32_t vfik_supported_joints(void) { return vfik::supported_joints_mask(); }

vfik_handle* vfik_create(int device, int io_dtype, int n_joints, int max_slots, int batch) {
    if (io_dtype != 32 && io_dtype != 64) { fail(VFIK_E_ARG, "io_dtype must be 32 or 64, got %d", io_dtype); return nullptr; }
    if (n_joints < 1 || n_joints > VFIK_MAX_JOINTS || !((vfik::supported_joints_mask() >> n_joints) & 1u)) {
        fail(VFIK_E_UNSUPPORTED, "no kernel built for %d joints (mask 0x%x)", n_joints, vfik::supported_joints_mask());
        return nullptr;
    }
    if (batch < 1 || max_slots < 0 || max_slots > 4096) { fail(VFIK_E_ARG, "bad batch %d / max_slots %d", batch, max_slots); return nullptr; }
    int ndev = vfik_device_count();
    if (device < 0 || device >= ndev) {
        fail(VFIK_E_HIP, "device %d not available (%d HIP devices visible); this library has no CPU path", device, ndev);
        return nullptr;
    }
    vfik_handle* h = new vfik_handle();
    h->device = device; h->io_dtype = io_dtype; h->n = n_joints; h->max_slots = max_slots; h->B = batch;
    h->esz = io_dtype == 32 ? 4 : 8;
    if (const char* e = std::getenv("VFIK_BLOCK")) {
        int b = std::atoi(e);
        if (b == 64 || b == 128 || b == 192 || b == 256) h->block = b;  // tuning knob; LDS per block = waves x 27-56 KB
    }
    if (const char* e = std::getenv("VFIK_SUB8_MAX_BATCH")) h->sub8_max_batch = h->sub8_max_batch_full = h->sub8_max_batch_ns = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("VFIK_PERSISTENT")) h->pers = std::atoi(e) != 0;
    if (const char* e = std::getenv("VFIK_TWO_WAVES")) h->waves2 = std::atoi(e) != 0;
    if (const char* e = std::getenv("VFIK_UNIFORM_IMAGE")) h->uni_allowed = std::atoi(e) != 0;
    if (const char* e = std::getenv("VFIK_MIXED_ORDERS")) h->mixed_allowed = std::atoi(e) != 0;
    if (const char* e = std::getenv("VFIK_DH_PATTERN")) h->dhp_allowed = std::atoi(e) != 0;
    if (const char* e = std::getenv("VFIK_ZERO_COPY_MAX")) h->zero_copy_max = (size_t)std::max(0L, std::atol(e));
    auto bail = [&](const char* what) { if (g_err.empty()) fail(VFIK_E_HIP, "%s failed", what); vfik_destroy(h); return (vfik_handle*)nullptr; };
    if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice");
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_simd = 4 * cus;
    }
    const size_t B = batch;
    h->Bpad = (batch + 63) / 64 * 64;
    const size_t quad_plane = (size_t)h->Bpad * 4 * h->esz;
    {   // the state a lean launch reads, in one allocation whose layout the kernel can derive (vfik_kernel.h: arena layout)
        const size_t sz_goal = (4 + 6) * quad_plane;   // goal block + aux block (funnel, hemisphere)
        const size_t sz_kc = VFIK_KCONST_SLOT(vfik::kconst_bytes(n_joints));   // (+ slack inside: the kinematics block is copied in whole 1-KiB rows)
        const size_t sz_lv = (size_t)((n_joints + 4) / 4) * h->Bpad * 4 * sizeof(float);
        const size_t sz_sf = (std::max<size_t>(1, (size_t)max_slots) + 1) / 2 * 3 * quad_plane;
        const size_t sz_su = (std::max<size_t>(1, (size_t)max_slots) + 1) * quad_plane;   // uniform image: an EMPTY plane, then one quad plane per slot
        const size_t sz_sl = std::max<size_t>(1, (size_t)max_slots) * 2 * quad_plane;  // >= 1 slot: the prefetch reads slot 0
        const size_t sz_or = (std::max<size_t>(1, (size_t)max_slots) + 15) / 16 * (size_t)h->Bpad * 16;   // order planes (zeros: order 0, force 0)
        if (dev_alloc(h, &h->d_arena, sz_goal + sz_kc + sz_lv + sz_sf + sz_su + sz_sl + sz_or, true)) return bail("alloc state arena");
        char* a0 = static_cast<char*>(h->d_arena);
        h->d_goal = a0;
        h->d_funnel = a0 + 4 * quad_plane;
        h->d_kconst = a0 + sz_goal;
        h->d_lastvec = reinterpret_cast<float*>(a0 + sz_goal + sz_kc);
        h->d_slots_fast = a0 + sz_goal + sz_kc + sz_lv;
        h->d_slots_uni = a0 + sz_goal + sz_kc + sz_lv + sz_sf;   // (kernel side: slots_fast + uni_planes quad planes)
        h->d_slots = a0 + sz_goal + sz_kc + sz_lv + sz_sf + sz_su;
        h->d_orders = a0 + sz_goal + sz_kc + sz_lv + sz_sf + sz_su + sz_sl;
    }
    {   // the uniform image starts out with every slot unused (radius -inf), like the zeros (force 0) of the other two images
        std::vector<char> plane((size_t)h->Bpad * 4 * h->esz, 0);
        for (int b = 0; b < h->Bpad; ++b) {
            if (io_dtype == 32) put<float>(plane, (size_t)b * 4 + 3, -std::numeric_limits<double>::infinity());
            else put<double>(plane, (size_t)b * 4 + 3, -std::numeric_limits<double>::infinity());
        }
        for (int sidx = 0; sidx < std::max(1, max_slots) + 1; ++sidx)   // (plane 0 stays like this for good: the slot every out-of-range quad reads)
            if (hipMemcpyAsync(static_cast<char*>(h->d_slots_uni) + (size_t)sidx * plane.size(), plane.data(), plane.size(), hipMemcpyHostToDevice, h->stream) != hipSuccess)
                return bail("init uniform image");
        if (hipStreamSynchronize(h->stream) != hipSuccess) return bail("init uniform image");
    }
    if (dev_alloc(h, (void**)&h->d_mixw, 16 * sizeof(double), true)) return bail("alloc mixw");
    h->slots_per_arm.assign(B, 0);
    h->fast_slots_per_arm.assign(B, 0);
    h->arm_has_funnel.assign(B, 0);
    h->arm_pair_state.assign(B, 0);
    h->arm_safe.assign(B, 0.0);
    h->arm_force.assign(B, 0.0);
    h->arm_order.assign(B, -1);
#ifdef VFIK_STAMPS
    if (dev_alloc(h, (void**)&h->d_stamps, ((B + 63) / 64) * 10 * sizeof(unsigned long long), true)) return bail("alloc stamps");
#endif
    // defaults: identity tool (vf:154), sig = 1 (nullspace:91), reference default parameters
    if (vfik_reset_state(h) != VFIK_OK) return bail("reset_state");
    vfik_params p{};
    p.speed_scale = 1.0; p.lambda = 0.1; p.rot_slowdown = 0.3; p.null_gain = 0.5; p.lookahead = 0.3;
    p.jl_gain = 0.5; p.max_vel = 1.0; p.jp_kp = 1.5; p.jp_delta = 0.087;
    for (double& w : p.wy) w = 1.0;
    for (double& w : p.wq) w = 1.0;
    p.mix_w[0] = p.mix_w[1] = 1.0;
    h->params = p;
    {
        std::vector<double> all(B, p.speed_scale);
        if (vfik_set_speed_scale(h, 0, batch, all.data()) != VFIK_OK) return bail("speed scale");
        h->speed_set = true;
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail("sync");
    return h;
}

void vfik_destroy(vfik_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->d_arena, h->d_tool, h->d_ext, h->d_mixw, h->d_stamps, h->d_mixw_arm, h->d_track, h->d_wts, h->d_rollq[0], h->d_rollq[1], h->d_objects, h->d_obs_pose, h->d_obs_v6};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->arena_dev) (void)hipFree(h->arena_dev);
    if (h->arena_host) (void)hipHostFree(h->arena_host);
    for (auto& sc : h->sc) if (sc.p) (void)hipFree(sc.p);
    for (auto& ps : h->pipe) {
        if (ps.ev_out) (void)hipEventSynchronize(ps.ev_out);
        for (auto& s : ps.sc) if (s.p) (void)hipFree(s.p);
        for (hipEvent_t e : {ps.ev_in, ps.ev_k, ps.ev_out}) if (e) (void)hipEventDestroy(e);
    }
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int vfik_set_stream(vfik_handle* h, void* hip_stream) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) HIP_TRY(hipStreamDestroy(h->stream));
    // the caller's stream is used as given; NULL is HIP's default (null) stream of the device,
    // which is what torch.cuda.current_stream().cuda_stream is unless the caller switched streams
    h->stream = static_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
    return VFIK_OK;
}

int vfik_set_chain(vfik_handle* h, const vfik_chain* c) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (!c) return fail(VFIK_E_ARG, "null chain");
    if (c->n != h->n) return fail(VFIK_E_ARG, "chain has %d joints, handle was created for %d", c->n, h->n);
    for (int i = 0; i < c->n; ++i) {
        if (c->jtype[i] != 0 && c->jtype[i] != 1) return fail(VFIK_E_ARG, "joint %d: unknown type %d", i, c->jtype[i]);
        if (!(c->q_hi[i] > c->q_lo[i])) return fail(VFIK_E_ARG, "joint %d: q_hi must exceed q_lo", i);
    }
    for (int i = 0; i <= c->n; ++i)
        for (int k = 0; k < 12; ++k)
            if (!std::isfinite(c->B[i][k])) return fail(VFIK_E_ARG, "chain transform %d has a non-finite entry", i);
    const vfik_chain saved = h->chain;
    const bool was_set = h->chain_set;
    h->chain = *c;
    h->chain_set = true;
    const int rc = upload_kconst(h);
    if (rc != VFIK_OK) { h->chain = saved; h->chain_set = was_set; }
    return rc;
}

static int upload_bridge_state(vfik_handle* h, int first_arm, int n_arms);

int vfik_set_params(vfik_handle* h, const vfik_params* p) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (!p) return fail(VFIK_E_ARG, "null params");
    if (!(p->lambda >= 0.0) || !(p->speed_scale >= 0.0) || !(p->max_vel >= 0.0) || !std::isfinite(p->lambda))
        return fail(VFIK_E_ARG, "lambda, speed_scale and max_vel must be finite and >= 0");
    if ((p->flags & VFIK_F_JOINT_LIMIT_TASK) && !(p->flags & VFIK_F_NULLSPACE))
        return fail(VFIK_E_ARG, "VFIK_F_JOINT_LIMIT_TASK needs VFIK_F_NULLSPACE");
    const bool speed_changed = !h->speed_set || p->speed_scale != h->params.speed_scale;
    bool weights_changed = false;
    for (int k = 0; k < 6; ++k) weights_changed = weights_changed || p->wy[k] != h->params.wy[k];
    for (int k = 0; k < h->n; ++k) weights_changed = weights_changed || p->wq[k] != h->params.wq[k];
    const bool maxvel_changed = p->max_vel != h->params.max_vel;
    bool mixw_changed = false;
    for (int k = 0; k < VFIK_MIX_CHANNELS; ++k) mixw_changed = mixw_changed || p->mix_w[k] != h->params.mix_w[k];
    h->params = *p;
    if ((maxvel_changed || mixw_changed) && h->d_mixw_arm) {
        // vfik_params.max_vel / mix_w are batch-wide: a CHANGED value is written to every arm, like speed_scale
        // (with per-arm bridge state the kernel reads nothing else)
        HIP_TRY(hipSetDevice(h->device));
        for (int b = 0; b < h->B; ++b) {
            if (maxvel_changed) h->bridge_host[(size_t)b * 8 + 6] = p->max_vel;
            for (int k = 0; mixw_changed && k < VFIK_MIX_CHANNELS; ++k) h->bridge_host[(size_t)b * 8 + k] = p->mix_w[k];
        }
        const int rc = upload_bridge_state(h, 0, h->B);
        if (rc != VFIK_OK) return rc;
    }
    if (weights_changed && h->d_wts) {  // new batch-wide IK weights replace every arm's own
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_wts);
        h->d_wts = nullptr;
    }
    if (speed_changed) {  // vfik_params.speed_scale is the batch-wide /max_vel value: written to every arm
        std::vector<double> all(h->B, p->speed_scale);
        const int rc = vfik_set_speed_scale(h, 0, h->B, all.data());
        if (rc != VFIK_OK) return rc;
        h->speed_set = true;
    }
    return upload_kconst(h);
}

// device image of the per-arm IK weights: [6 + n][Bpad] doubles, every arm starting from the batch's
static int ensure_arm_weights(vfik_handle* h) {
    if (h->d_wts) return VFIK_OK;
    const size_t Bp = h->Bpad, rows = 6 + h->n;
    std::vector<double> img(rows * Bp, 1.0);
    for (size_t r = 0; r < rows; ++r) {
        const double v = r < 6 ? h->params.wy[r] : h->params.wq[r - 6];
        for (size_t b = 0; b < Bp; ++b) img[r * Bp + b] = v;
    }
    if (dev_alloc(h, (void**)&h->d_wts, img.size() * sizeof(double), false)) return VFIK_E_HIP;
    HIP_TRY(hipMemcpyAsync(h->d_wts, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

int vfik_set_arm_weights(vfik_handle* h, int first_arm, int n_arms, const double* wy, const double* wq) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (first_arm < 0 || n_arms < 1 || first_arm + n_arms > h->B) return fail(VFIK_E_ARG, "arm range [%d, %d) outside batch %d", first_arm, first_arm + n_arms, h->B);
    if (!wy && !wq) return fail(VFIK_E_ARG, "vfik_set_arm_weights: give wy, wq or both");
    for (int j = 0; j < n_arms; ++j) {
        for (int k = 0; wy && k < 6; ++k)
            if (!std::isfinite(wy[(size_t)j * 6 + k])) return fail(VFIK_E_ARG, "arm %d: task weight %d is not finite", first_arm + j, k);
        for (int k = 0; wq && k < h->n; ++k)
            if (!std::isfinite(wq[(size_t)j * h->n + k])) return fail(VFIK_E_ARG, "arm %d: joint weight %d is not finite", first_arm + j, k);
    }
    HIP_TRY(hipSetDevice(h->device));
    if (first_arm == 0 && n_arms == h->B && wy && wq) {
        // The whole batch, and every arm with the same weights (a port-level caller forwards each arm's /weight bottle): these ARE batch-wide
        // weights -- stored as such, the arms' own dropped, so that the launches stay on the kernels built for plain chains (WTSC) instead
        // of the general variants.
        bool same = true;
        for (int j = 1; j < n_arms && same; ++j)
            same = std::memcmp(wy + (size_t)j * 6, wy, 6 * sizeof(double)) == 0 && std::memcmp(wq + (size_t)j * h->n, wq, h->n * sizeof(double)) == 0;
        if (same) {
            for (int k = 0; k < 6; ++k) h->params.wy[k] = wy[k];
            for (int k = 0; k < h->n; ++k) h->params.wq[k] = wq[k];
            if (h->d_wts) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                (void)hipFree(h->d_wts);
                h->d_wts = nullptr;
            }
            return upload_kconst(h);
        }
    }
    if (ensure_arm_weights(h) != VFIK_OK) return VFIK_E_HIP;
    const size_t Bp = h->Bpad;
    std::vector<double> row(n_arms);
    for (int r = 0; r < 6 + h->n; ++r) {
        const double* src = r < 6 ? wy : wq;
        if (!src) continue;
        const int stride = r < 6 ? 6 : h->n, col = r < 6 ? r : r - 6;
        for (int j = 0; j < n_arms; ++j) row[j] = src[(size_t)j * stride + col];
        HIP_TRY(hipMemcpyAsync(h->d_wts + (size_t)r * Bp + first_arm, row.data(), n_arms * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));  // `row` is reused
    }
    return VFIK_OK;
}

int vfik_set_tool(vfik_handle* h, const double* tool16, int per_arm) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (!tool16) return fail(VFIK_E_ARG, "null tool");
    HIP_TRY(hipSetDevice(h->device));
    if (!per_arm) {  // one sticky tool frame for the batch: lives with the other shared constants
        for (int k = 0; k < 12; ++k) h->tool_shared[k] = tool16[k];
        h->tool_per_arm = 0;
        return upload_kconst(h);
    }
    const size_t B = h->B, Bp = h->Bpad;
    {   // every arm with the same frame (a port-level caller forwards each arm's /tool bottle, and a fleet of one robot type carries one
        // hand): that is the batch's shared tool, with the values as the per-arm image would hold them (rounded to the I/O type) -- the
        // launch stays on the PLAIN kernels (vfik_kernel.hip, TOOLC) instead of the general variants, 6.9 against 10.6 us for C3N
        bool same = true;
        for (size_t b = 1; b < B && same; ++b) same = std::memcmp(tool16 + b * 16, tool16, 12 * sizeof(double)) == 0;
        if (same) {
            for (int k = 0; k < 12; ++k) h->tool_shared[k] = h->io_dtype == 32 ? (double)(float)tool16[k] : tool16[k];
            h->tool_per_arm = 0;
            return upload_kconst(h);
        }
    }
    std::vector<char> buf(3 * Bp * 4 * h->esz, 0);
    for (size_t b = 0; b < B; ++b)
        for (int k = 0; k < 12; ++k) {  // rows 0..2 of the 4x4 -> 3 quad planes
            const double v = tool16[b * 16 + k];
            const size_t idx = ((size_t)(k >> 2) * Bp + b) * 4 + (k & 3);
            if (h->io_dtype == 32) put<float>(buf, idx, v); else put<double>(buf, idx, v);
        }
    if (!h->d_tool && dev_alloc(h, &h->d_tool, buf.size(), false)) return VFIK_E_HIP;
    HIP_TRY(hipMemcpyAsync(h->d_tool, buf.data(), buf.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->tool_per_arm = 1;
    return VFIK_OK;
}

int vfik_set_fields(vfik_handle* h, int first_arm, int n_arms, const vfik_field* fields, int max_fields,
                    const int32_t* counts) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (!fields || !counts) return fail(VFIK_E_ARG, "null fields / counts");
    if (first_arm < 0 || n_arms < 1 || first_arm + n_arms > h->B) return fail(VFIK_E_ARG, "arm range [%d, %d) outside batch %d", first_arm, first_arm + n_arms, h->B);
    if (max_fields < 0) return fail(VFIK_E_ARG, "negative max_fields");
    // validate everything before touching device state
    for (int j = 0; j < n_arms; ++j) {
        if (counts[j] < 0 || counts[j] > max_fields) return fail(VFIK_E_ARG, "arm %d: count %d outside [0, %d]", first_arm + j, counts[j], max_fields);
        int need = 0;
        bool goal = false;
        for (int k = 0; k < counts[j]; ++k) {
            const vfik_field& fd = fields[(size_t)j * max_fields + k];
            const int ns = slots_of(fd.type);
            if (ns < 0) return fail(VFIK_E_ARG, "arm %d field %d: unknown type %d", first_arm + j, fd.id, fd.type);
            if (fd.type == VFIK_FIELD_ATTRACTOR && !goal) { goal = true; continue; }
            need += ns;
        }
        if (need > h->max_slots) return fail(VFIK_E_ARG, "arm %d needs %d slots, handle capacity is %d", first_arm + j, need, h->max_slots);
    }
    HIP_TRY(hipSetDevice(h->device));
    std::vector<char> goal, slots, fast, funnel, hasf(n_arms), uni, pstate(n_arms);
    std::vector<double> psafe(n_arms), pforce(n_arms);
    std::vector<int> used(n_arms), used_fast(n_arms);
    std::vector<unsigned char> ord;
    const int S = h->max_slots;
    if (h->io_dtype == 32) pack_fields<float>(fields, max_fields, counts, n_arms, S, goal, slots, fast, used, funnel, used_fast, hasf, uni, pstate, psafe, pforce, ord);
    else pack_fields<double>(fields, max_fields, counts, n_arms, S, goal, slots, fast, used, funnel, used_fast, hasf, uni, pstate, psafe, pforce, ord);
    const size_t qb = 4 * h->esz, w = (size_t)n_arms * qb, pitch = (size_t)h->Bpad * qb;
    char* dg = static_cast<char*>(h->d_goal) + (size_t)first_arm * qb;
    HIP_TRY(hipMemcpy2DAsync(dg, pitch, goal.data(), w, w, 3, hipMemcpyHostToDevice, h->stream));
    // plane 3 = (present, slow-down, force, speedScale): the 4th component belongs to vfik_set_speed_scale
    HIP_TRY(hipMemcpy2DAsync(dg + 3 * pitch, qb, goal.data() + 3 * w, qb, 3 * h->esz, n_arms, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpy2DAsync(static_cast<char*>(h->d_funnel) + (size_t)first_arm * qb, pitch, funnel.data(), w, w, 6, hipMemcpyHostToDevice, h->stream));
    if (S > 0) {
        char* ds = static_cast<char*>(h->d_slots) + (size_t)first_arm * qb;
        HIP_TRY(hipMemcpy2DAsync(ds, pitch, slots.data(), w, w, (size_t)S * 2, hipMemcpyHostToDevice, h->stream));
        char* df = static_cast<char*>(h->d_slots_fast) + (size_t)first_arm * qb;
        HIP_TRY(hipMemcpy2DAsync(df, pitch, fast.data(), w, w, (size_t)((S + 1) / 2) * 3, hipMemcpyHostToDevice, h->stream));
        char* du = static_cast<char*>(h->d_slots_uni) + pitch + (size_t)first_arm * qb;   // (slot m in plane m + 1)
        HIP_TRY(hipMemcpy2DAsync(du, pitch, uni.data(), w, w, (size_t)S, hipMemcpyHostToDevice, h->stream));
        char* dor = static_cast<char*>(h->d_orders) + (size_t)first_arm * 16;
        HIP_TRY(hipMemcpy2DAsync(dor, (size_t)h->Bpad * 16, ord.data(), (size_t)n_arms * 16, (size_t)n_arms * 16, (size_t)(S + 15) / 16, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int j = 0; j < n_arms; ++j) {
        h->slots_per_arm[first_arm + j] = used[j];
        h->fast_slots_per_arm[first_arm + j] = used_fast[j];
        h->arm_has_funnel[first_arm + j] = hasf[j];
        h->arm_order[first_arm + j] = classify_arm(fields + (size_t)j * max_fields, counts[j]);
        h->arm_pair_state[first_arm + j] = pstate[j];
        h->arm_safe[first_arm + j] = psafe[j];
        h->arm_force[first_arm + j] = pforce[j];
    }
    h->slots_used = *std::max_element(h->slots_per_arm.begin(), h->slots_per_arm.end());
    h->slots_used_fast = *std::max_element(h->fast_slots_per_arm.begin(), h->fast_slots_per_arm.end());
    h->any_funnel = 0;
    for (char f : h->arm_has_funnel) h->any_funnel |= f;
    // One integer order for every decay repeller of the batch: the straight-line path with that order as a launch constant.  Integer
    // orders that differ -- within an arm (-3) or between arms: the straight-line path still, reading the order planes (`mixed`).
    int fo = -1;
    bool general = false, mixed = false;
    for (int o : h->arm_order) {
        if (o == -2) { general = true; break; }
        if (o == -3 || (o >= 0 && fo >= 0 && o != fo)) mixed = true;
        if (o >= 0) fo = o;
    }
    if (!h->mixed_allowed && mixed) general = true;
    h->mixed = (!general && mixed) ? 1 : 0;
    h->fast_order = general ? -1 : (fo < 0 ? 5 : fo);   // (no repeller anywhere: any order >= 1 will do -- see the uniform image below)
    // one (safe distance, force) for every decay repeller of the batch?  Then the pair goes into the device constants and the lean
    // launches read the uniform image.
    // (not with decay order 0: the uniform image disarms an unused slot through its magnitude, ((radius + safe) / D)^order = 0^order,
    // which is 1 for order 0)
    bool uni_ok = !general && h->fast_order != 0, have = false;
    double us = 0.0, uf = 0.0;
    for (int b = 0; b < h->B && uni_ok; ++b) {
        const char st = h->arm_pair_state[b];
        if (st == 0) continue;
        if (st == 2) { uni_ok = false; break; }
        if (!have) { have = true; us = h->arm_safe[b]; uf = h->arm_force[b]; }
        else if (h->arm_safe[b] != us || h->arm_force[b] != uf) uni_ok = false;
    }
    if (uni_ok && have && (us != h->uni_safe || uf != h->uni_force)) {
        h->uni_safe = us; h->uni_force = uf;
        if (h->chain_set) {  // (a chain set later writes the pair with the rest of the constants: upload_kconst)
            char* kc = static_cast<char*>(h->d_kconst);
            HIP_TRY(hipMemcpyAsync(kc + VFIK_KCONST_REP_SAFE_OFF(h->n), &h->uni_safe, sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(kc + VFIK_KCONST_REP_FORCE_OFF, &h->uni_force, sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
    }
    h->uni_ok = uni_ok ? 1 : 0;
    return VFIK_OK;
}

int vfik_set_speed_scale(vfik_handle* h, int first_arm, int n_arms, const double* values) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (!values) return fail(VFIK_E_ARG, "null values");
    if (first_arm < 0 || n_arms < 1 || first_arm + n_arms > h->B) return fail(VFIK_E_ARG, "arm range [%d, %d) outside batch %d", first_arm, first_arm + n_arms, h->B);
    for (int j = 0; j < n_arms; ++j)
        if (!(values[j] >= 0.0) || !std::isfinite(values[j])) return fail(VFIK_E_ARG, "arm %d: speedScale must be finite and >= 0", first_arm + j);
    HIP_TRY(hipSetDevice(h->device));
    std::vector<char> buf((size_t)n_arms * h->esz);
    for (int j = 0; j < n_arms; ++j) { if (h->io_dtype == 32) put<float>(buf, j, values[j]); else put<double>(buf, j, values[j]); }
    const size_t qb = 4 * h->esz;
    char* dst = static_cast<char*>(h->d_goal) + 3 * (size_t)h->Bpad * qb + (size_t)first_arm * qb + 3 * h->esz;
    HIP_TRY(hipMemcpy2DAsync(dst, qb, buf.data(), h->esz, h->esz, n_arms, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

// Per-arm bridge state on the device: 2 quad planes [w0 w1 w2 w3 | w4 w5 max_vel -], mirrored on the host so
// that the two setters below can rewrite an arm's quads without reading them back.
static int upload_bridge_state(vfik_handle* h, int first_arm, int n_arms);
static int ensure_bridge_state(vfik_handle* h) {
    if (h->d_mixw_arm) return VFIK_OK;
    const size_t plane = (size_t)h->Bpad * 4 * h->esz;
    if (dev_alloc(h, &h->d_mixw_arm, 2 * plane, true)) return VFIK_E_HIP;
    h->bridge_host.assign((size_t)h->B * 8, 0.0);
    for (int b = 0; b < h->B; ++b) {  // every arm starts from the batch-wide values
        for (int k = 0; k < VFIK_MIX_CHANNELS; ++k) h->bridge_host[(size_t)b * 8 + k] = h->params.mix_w[k];
        h->bridge_host[(size_t)b * 8 + 6] = h->params.max_vel;
    }
    return upload_bridge_state(h, 0, h->B);
}

static int upload_bridge_state(vfik_handle* h, int first_arm, int n_arms) {
    const size_t qb = 4 * h->esz, plane = (size_t)h->Bpad * qb;
    std::vector<char> buf(2 * (size_t)n_arms * qb, 0);
    for (int j = 0; j < n_arms; ++j)
        for (int k = 0; k < 8; ++k) {
            const size_t idx = ((size_t)(k >> 2) * n_arms + j) * 4 + (k & 3);
            const double v = h->bridge_host[(size_t)(first_arm + j) * 8 + k];
            if (h->io_dtype == 32) put<float>(buf, idx, v); else put<double>(buf, idx, v);
        }
    char* dst = static_cast<char*>(h->d_mixw_arm) + (size_t)first_arm * qb;
    HIP_TRY(hipMemcpy2DAsync(dst, plane, buf.data(), (size_t)n_arms * qb, (size_t)n_arms * qb, 2, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // all arms alike?  (O(B) per call; the callers are setters, not the cycle)
    const bool was = h->bridge_uniform;
    bool same = true;
    for (int b = 1; b < h->B && same; ++b) same = std::memcmp(&h->bridge_host[(size_t)b * 8], &h->bridge_host[0], 7 * sizeof(double)) == 0;
    h->bridge_uniform = same;
    if (same)
        for (int k = 0; k < 7; ++k) h->bridge_u[k] = h->io_dtype == 32 ? (double)(float)h->bridge_host[k] : h->bridge_host[k];
    if (same || was) return upload_kconst(h);
    return VFIK_OK;
}

int vfik_set_mixer_weights(vfik_handle* h, int first_arm, int n_arms, const double* w) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    HIP_TRY(hipSetDevice(h->device));
    if (!w) {  // every arm's mixer weights back to the batch-wide vfik_params.mix_w; per-arm limiter speeds stay
        if (!h->d_mixw_arm) return VFIK_OK;
        for (int b = 0; b < h->B; ++b)
            for (int k = 0; k < VFIK_MIX_CHANNELS; ++k) h->bridge_host[(size_t)b * 8 + k] = h->params.mix_w[k];
        return upload_bridge_state(h, 0, h->B);
    }
    if (first_arm < 0 || n_arms < 1 || first_arm + n_arms > h->B) return fail(VFIK_E_ARG, "arm range [%d, %d) outside batch %d", first_arm, first_arm + n_arms, h->B);
    if (ensure_bridge_state(h) != VFIK_OK) return VFIK_E_HIP;
    for (int j = 0; j < n_arms; ++j)
        for (int k = 0; k < VFIK_MIX_CHANNELS; ++k) h->bridge_host[(size_t)(first_arm + j) * 8 + k] = w[(size_t)j * VFIK_MIX_CHANNELS + k];
    return upload_bridge_state(h, first_arm, n_arms);
}

int vfik_set_max_vel(vfik_handle* h, int first_arm, int n_arms, const double* values) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (!values) return fail(VFIK_E_ARG, "null values");
    if (first_arm < 0 || n_arms < 1 || first_arm + n_arms > h->B) return fail(VFIK_E_ARG, "arm range [%d, %d) outside batch %d", first_arm, first_arm + n_arms, h->B);
    for (int j = 0; j < n_arms; ++j)
        if (!(values[j] >= 0.0) || !std::isfinite(values[j])) return fail(VFIK_E_ARG, "arm %d: max_vel %g must be finite and >= 0", first_arm + j, values[j]);
    HIP_TRY(hipSetDevice(h->device));
    if (ensure_bridge_state(h) != VFIK_OK) return VFIK_E_HIP;
    for (int j = 0; j < n_arms; ++j) h->bridge_host[(size_t)(first_arm + j) * 8 + 6] = values[j];
    return upload_bridge_state(h, first_arm, n_arms);
}

int vfik_set_ext_cmd(vfik_handle* h, int channel, const void* cmd_host) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (channel < 2 || channel >= VFIK_MIX_CHANNELS) return fail(VFIK_E_ARG, "channel %d: only 2..%d are external", channel, VFIK_MIX_CHANNELS - 1);
    HIP_TRY(hipSetDevice(h->device));
    const size_t chan = (size_t)h->B * h->n * h->esz;
    if (!h->d_ext) {
        if (!cmd_host) return VFIK_OK;  // zeroing a channel that was never set
        if (dev_alloc(h, &h->d_ext, chan * (VFIK_MIX_CHANNELS - 2), true)) return VFIK_E_HIP;
    }
    char* dst = static_cast<char*>(h->d_ext) + chan * (channel - 2);
    if (cmd_host) HIP_TRY(hipMemcpyAsync(dst, cmd_host, chan, hipMemcpyHostToDevice, h->stream));
    else HIP_TRY(hipMemsetAsync(dst, 0, chan, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

int vfik_reset_state(vfik_handle* h) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    HIP_TRY(hipSetDevice(h->device));
    // lastvec = 0, sig = +1 (element n of every arm's state)
    const size_t planes = (size_t)(h->n + 4) / 4, Bp = h->Bpad;
    std::vector<float> img(planes * Bp * 4, 0.0f);
    for (size_t b = 0; b < Bp; ++b) img[((size_t)(h->n / 4) * Bp + b) * 4 + (h->n & 3)] = 1.0f;
    HIP_TRY(hipMemcpyAsync(h->d_lastvec, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

static int launch_cycles(vfik_handle* h, const vfik_io* io, int n_cycles, double dt, int clamp, void* q_out, hipStream_t stream) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!io || !io->q) return fail(VFIK_E_ARG, "a control cycle needs io->q");
    if (!h->chain_set) return fail(VFIK_E_STATE, "vfik_set_chain has not been called");
    if (n_cycles > 0 && io->q_cmded) return fail(VFIK_E_ARG, "io->q_cmded (LWR position command form) is for vfik_step only");
    if (!io->q_lo != !io->q_hi) return fail(VFIK_E_ARG, "io->q_lo and io->q_hi come together (both or neither)");
    HIP_TRY(hipSetDevice(h->device));
    vfik::KArgs a;
    fill_kargs(h, io, a);
    a.dt = dt;
    a.clamp = clamp ? 1 : 0;
    if (n_cycles > 0 && (io->track_error || io->obj_dist))
        return fail(VFIK_E_ARG, "io->track_error / io->obj_dist are per control cycle: vfik_step only, not a rollout");
    if (n_cycles > 0 && (h->n > VFIK_ROLL_MAX_NJ || a.plain != 1)) {
        // Long chains, and (round 4) chains with a tool, IK weights or prismatic joints, whose in-kernel loop spilled 12-268 B per
        // lane: the rollout is n_cycles single-cycle launches, each integrating q on its way out
        // (q ping-pongs between two device buffers; the caller's io->q is never written).  The kernel has
        // no registers left for loop-carried state at these sizes -- the in-kernel loop spills and is
        // slower than this (C5: 22 us per cycle against 17.4 us).  Outputs are those of the last cycle,
        // status bits accumulate, the nullspace state advances launch by launch.
        const size_t qbytes = (size_t)h->B * h->n * h->esz;
        for (int k = 0; k < 2; ++k)
            if (!h->d_rollq[k] && dev_alloc(h, &h->d_rollq[k], qbytes, false)) return VFIK_E_HIP;
        a.n_cycles = 0;
        const void* q_in = io->q;
        for (int c = 0; c < n_cycles; ++c) {
            const bool last = c == n_cycles - 1;
            vfik::KArgs k = a;
            k.q = q_in;
            k.q_out = (last && q_out) ? q_out : h->d_rollq[c & 1];
            k.status_or = c > 0;
            if (!last) {  // intermediate cycles produce no outputs but the status bits
                k.qdot_vf = k.qdot_null = k.qdot_out = k.pose = k.pose_nt = k.v6 = k.qdist = k.goal_dist = k.q_ref_out = nullptr;
            }
            hipError_t e = vfik::launch_cycle(h->io_dtype, h->n, k, h->block, stream);
            if (e != hipSuccess) return fail(VFIK_E_HIP, "kernel launch: %s", hipGetErrorString(e));
            q_in = k.q_out;
        }
        return VFIK_OK;
    }
    a.n_cycles = n_cycles;
    a.q_out = q_out;
    // observers of the cycle (ABI 4): they read the cycle's own pose / twist on the device
    const bool want_track = io->track_error != nullptr, want_dist = io->obj_dist != nullptr;
    if (want_track || want_dist) {
        if (want_dist && (!h->d_objects || h->n_objects < 1)) return fail(VFIK_E_STATE, "io->obj_dist needs vfik_set_objects");
        // the observers' buffers are allocated at the first request: never under stream capture (an allocation there would be part of
        // the captured work or fail it) -- a caller that captures vfik_step with observers makes one such call outside the capture first
        if ((!a.pose && !h->d_obs_pose) || (want_track && ((!a.v6 && !h->d_obs_v6) || !h->d_track))) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return fail(VFIK_E_STATE, "io->track_error / io->obj_dist: the observers' device buffers are allocated at the first request -- make one such vfik_step call before capturing the stream");
            (void)hipGetLastError();
        }
        if (!a.pose) {
            if (!h->d_obs_pose && dev_alloc(h, &h->d_obs_pose, (size_t)h->B * 16 * h->esz, true)) return VFIK_E_HIP;
            a.pose = h->d_obs_pose;
        }
        if (want_track && !a.v6) {
            if (!h->d_obs_v6 && dev_alloc(h, &h->d_obs_v6, (size_t)h->B * 6 * h->esz, true)) return VFIK_E_HIP;
            a.v6 = h->d_obs_v6;
        }
        if (want_track && !h->d_track && dev_alloc(h, (void**)&h->d_track, (size_t)38 * h->B * sizeof(double), true)) return VFIK_E_HIP;
    }
    int sub8 = 0;
    hipError_t e = vfik::launch_cycle(h->io_dtype, h->n, a, h->block, stream, &sub8);
    h->sub8_launches += sub8;
    if (e != hipSuccess) return fail(VFIK_E_HIP, "kernel launch: %s", hipGetErrorString(e));
    if (want_track) {
        e = vfik::launch_track(h->io_dtype, a.pose, a.v6, h->d_track, io->track_error, io->active, h->B, stream);
        if (e != hipSuccess) return fail(VFIK_E_HIP, "track launch: %s", hipGetErrorString(e));
    }
    if (want_dist) {
        e = vfik::launch_monitor(h->io_dtype, a.pose, h->d_objects, h->n_objects, (long)h->B * h->n_objects, io->obj_dist, stream, io->active);
        if (e != hipSuccess) return fail(VFIK_E_HIP, "monitor launch: %s", hipGetErrorString(e));
    }
    return VFIK_OK;
}

int vfik_step(vfik_handle* h, const vfik_io* io) { return launch_cycles(h, io, 0, 0.0, 0, nullptr, h ? h->stream : nullptr); }

int vfik_rollout(vfik_handle* h, const vfik_io* io, int n_cycles, double dt, int clamp_to_limits, void* q_out) {
    if (n_cycles < 1 || n_cycles > 1000000) return fail(VFIK_E_ARG, "n_cycles %d outside [1, 1e6]", n_cycles);
    if (!std::isfinite(dt)) return fail(VFIK_E_ARG, "dt must be finite");
    return launch_cycles(h, io, n_cycles, dt, clamp_to_limits, q_out, h ? h->stream : nullptr);
}

int vfik_sync(vfik_handle* h) {
    if (check_handle(h)) return VFIK_E_ARG;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

// host-pointer forms: which vfik_io members are inputs / outputs, and their sizes in bytes
namespace {
constexpr int N_HIN = 7, N_HOUT = 13;
struct HostIo {
    const void* hin[N_HIN];
    size_t bin[N_HIN];
    void* hout[N_HOUT];
    size_t bout[N_HOUT];
};
HostIo host_io(const vfik_handle* h, const vfik_io* io, void* q_out_host) {
    const size_t B = h->B, n = h->n, e = h->esz;
    HostIo x{{io->q, io->null_control, io->q_ref, io->q_cmded, io->active, io->q_lo, io->q_hi},
             {B * n * e, B * VFIK_NULL_CONTROLS * e, B * n * e, B * n * e, B * sizeof(int32_t), B * n * e, B * n * e},
             {io->qdot_vf, io->qdot_null, io->qdot_out, io->pose, io->pose_nt, io->v6, io->qdist, io->status, q_out_host, io->goal_dist,
              io->q_ref ? io->q_ref_out : nullptr, io->track_error, io->obj_dist},
             {B * n * e, B * n * e, B * n * e, B * 16 * e, B * 16 * e, B * 6 * e, B * n * e, B * sizeof(int32_t), B * n * e, B * 2 * e, B * n * e,
              B * 8 * e, B * (size_t)(h->n_objects > 0 ? h->n_objects : 1) * 2 * e}};
    return x;
}
void device_io(void* const* din, void* const* dout, vfik_io& d) {
    d = vfik_io{};
    d.q = din[0]; d.null_control = din[1]; d.q_ref = din[2]; d.q_cmded = din[3];
    d.active = static_cast<const int32_t*>(din[4]); d.q_lo = din[5]; d.q_hi = din[6];
    d.qdot_vf = dout[0]; d.qdot_null = dout[1]; d.qdot_out = dout[2]; d.pose = dout[3]; d.pose_nt = dout[4];
    d.v6 = dout[5]; d.qdist = dout[6]; d.status = static_cast<int32_t*>(dout[7]); d.goal_dist = dout[9]; d.q_ref_out = dout[10];
    d.track_error = dout[11]; d.obj_dist = dout[12];
}
}  // namespace

// Host-pointer form, calls whose bytes are small against their member count: ONE copy in, the launches, ONE copy out, one synchronisation.  Inputs and
// outputs live in one device arena and one pinned host arena (same layout, every member 256-byte aligned); the caller's arrays are packed into /
// unpacked from the pinned arena on the host.  (Until round 3 every member was its own hipMemcpyAsync: with the eleven
// outputs the port-level host layer asks for that was 197 us per call for ONE arm, ~15 us per copy, against 34 us for
// qdot_out alone.)
static int cycles_host(vfik_handle* h, const vfik_io* io, int n_cycles, double dt, int clamp, void* q_out_host) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!io || !io->q) return fail(VFIK_E_ARG, "a control cycle needs io->q");
    HIP_TRY(hipSetDevice(h->device));
    const HostIo x = host_io(h, io, q_out_host);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off_in[N_HIN], off_out[N_HOUT], total = 0;
    for (int i = 0; i < N_HIN; ++i) { off_in[i] = total; if (x.hin[i]) total += up(x.bin[i]); }
    const size_t in_bytes = total;
    for (int i = 0; i < N_HOUT; ++i) { off_out[i] = total; if (x.hout[i]) total += up(x.bout[i]); }
    int members = 0;
    for (int i = 0; i < N_HIN; ++i) members += x.hin[i] != nullptr;
    for (int i = 0; i < N_HOUT; ++i) members += x.hout[i] != nullptr;
    // The arena saves ~15 us per member beyond two and costs a host memcpy of every byte (~38 GB/s): measured 197 -> 40 us for
    // one arm with 13 members, 306 -> 190 us for 4 096 arms (2 MB), but 142 -> 236 us for a C3 step (q and qdot_out, 3.6 MB).
    if (total > ((size_t)256 << 10) && total > (size_t)std::max(0, members - 2) * ((size_t)512 << 10)) {
        // Few large members: the copies are bandwidth, not count -- every member straight between the caller's array and its
        // own device buffer.
        auto need = [&](int i, size_t bytes) -> void* {
            auto& sc = h->sc[i];
            if (sc.bytes < bytes) {
                if (sc.p) (void)hipFree(sc.p);
                sc.p = nullptr; sc.bytes = 0;
                if (hipMalloc(&sc.p, bytes) != hipSuccess) return nullptr;
                sc.bytes = bytes;
            }
            return sc.p;
        };
        void* din[N_HIN] = {};
        for (int i = 0; i < N_HIN; ++i)
            if (x.hin[i]) {
                din[i] = need(i, x.bin[i]);
                if (!din[i]) return fail(VFIK_E_HIP, "scratch allocation failed");
                HIP_TRY(hipMemcpyAsync(din[i], x.hin[i], x.bin[i], hipMemcpyHostToDevice, h->stream));
            }
        void* dout[N_HOUT];
        for (int i = 0; i < N_HOUT; ++i) {
            dout[i] = x.hout[i] ? need(N_HIN + i, x.bout[i]) : nullptr;
            if (x.hout[i] && !dout[i]) return fail(VFIK_E_HIP, "scratch allocation failed");
            // gated arms store nothing: their rows of the caller's arrays must come back as they went in
            if (x.hout[i] && io->active) HIP_TRY(hipMemcpyAsync(dout[i], x.hout[i], x.bout[i], hipMemcpyHostToDevice, h->stream));
        }
        vfik_io d;
        device_io(din, dout, d);
        const int rc = n_cycles > 0 ? vfik_rollout(h, &d, n_cycles, dt, clamp, dout[8]) : vfik_step(h, &d);
        if (rc != VFIK_OK) return rc;
        for (int i = 0; i < N_HOUT; ++i)
            if (x.hout[i]) HIP_TRY(hipMemcpyAsync(x.hout[i], dout[i], x.bout[i], hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return VFIK_OK;
    }
    if (h->arena_bytes < total) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->arena_dev) (void)hipFree(h->arena_dev);
        if (h->arena_host) (void)hipHostFree(h->arena_host);
        h->arena_dev = h->arena_host = nullptr;
        h->arena_bytes = 0;
        const size_t cap = total + total / 4 + 4096;
        if (hipMalloc(&h->arena_dev, cap) != hipSuccess || hipHostMalloc(&h->arena_host, cap, hipHostMallocMapped) != hipSuccess)
            return fail(VFIK_E_HIP, "arena allocation of %zu bytes failed", cap);
        if (hipHostGetDevicePointer(&h->arena_host_dev, h->arena_host, 0) != hipSuccess) { (void)hipGetLastError(); h->arena_host_dev = nullptr; }
        h->arena_bytes = cap;
    }
    char* const hostA = static_cast<char*>(h->arena_host);
    // A handful of arms (what vfclik itself runs: one arm per process set): no copy at all -- the kernels read the inputs from, and write
    // the outputs to, the pinned arena over the bus.  Two copy submissions and their DMA round trips cost such a call more than the
    // kernel's few PCIe transactions (ccb_rate: one arm, qdot_out only 23 -> 17 us; profiles/r03_ccb_rate.txt).
    const bool zero_copy = total <= h->zero_copy_max && h->arena_host_dev;
    char* const devA = zero_copy ? static_cast<char*>(h->arena_host_dev) : static_cast<char*>(h->arena_dev);
    void* din[N_HIN] = {};
    void* dout[N_HOUT] = {};
    for (int i = 0; i < N_HIN; ++i)
        if (x.hin[i]) { std::memcpy(hostA + off_in[i], x.hin[i], x.bin[i]); din[i] = devA + off_in[i]; }
    size_t h2d = in_bytes;
    for (int i = 0; i < N_HOUT; ++i)
        if (x.hout[i]) {
            dout[i] = devA + off_out[i];
            // gated arms store nothing: their rows of the caller's arrays must come back as they went in
            if (io->active) { std::memcpy(hostA + off_out[i], x.hout[i], x.bout[i]); h2d = total; }
        }
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(devA, hostA, h2d, hipMemcpyHostToDevice, h->stream));
    vfik_io d;
    device_io(din, dout, d);
    const int rc = n_cycles > 0 ? vfik_rollout(h, &d, n_cycles, dt, clamp, dout[8]) : vfik_step(h, &d);
    if (rc != VFIK_OK) return rc;
    if (!zero_copy && total > in_bytes) HIP_TRY(hipMemcpyAsync(hostA + in_bytes, devA + in_bytes, total - in_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < N_HOUT; ++i)
        if (x.hout[i]) std::memcpy(x.hout[i], hostA + off_out[i], x.bout[i]);
    return VFIK_OK;
}

int vfik_step_host(vfik_handle* h, const vfik_io* io) { return cycles_host(h, io, 0, 0.0, 0, nullptr); }

int vfik_rollout_host(vfik_handle* h, const vfik_io* io, int n_cycles, double dt, int clamp_to_limits, void* q_out) {
    if (n_cycles < 1) return fail(VFIK_E_ARG, "n_cycles must be >= 1");
    return cycles_host(h, io, n_cycles, dt, clamp_to_limits, q_out);
}

// ---- pipelined host path -------------------------------------------------------------------------
void* vfik_host_alloc(vfik_handle* h, size_t bytes) {
    if (!h || bytes == 0) { fail(VFIK_E_ARG, "vfik_host_alloc: bad arguments"); return nullptr; }
    void* p = nullptr;
    if (hipSetDevice(h->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        fail(VFIK_E_HIP, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

int vfik_host_free(vfik_handle* h, void* p) {
    if (check_handle(h)) return VFIK_E_ARG;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipHostFree(p));
    return VFIK_OK;
}

int vfik_wait(vfik_handle* h, long ticket) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (ticket < 0 || ticket >= h->next_ticket) return fail(VFIK_E_ARG, "vfik_wait: unknown ticket %ld", ticket);
    auto& ps = h->pipe[ticket % vfik_handle::PIPE];
    if (ps.ticket != ticket) return VFIK_OK;  // already waited for (or its slot was recycled, which waits)
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(ps.ev_out));
    ps.ticket = -1;
    return VFIK_OK;
}

int vfik_submit_host(vfik_handle* h, const vfik_io* io, long* ticket) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!io || !io->q || !ticket) return fail(VFIK_E_ARG, "vfik_submit_host needs io->q and a ticket");
    if (!h->chain_set) return fail(VFIK_E_STATE, "vfik_set_chain has not been called");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->s_in) {  // non-blocking: no implicit ordering against the NULL stream a caller may have chosen
        HIP_TRY(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    }
    auto& ps = h->pipe[h->next_ticket % vfik_handle::PIPE];
    if (!ps.ev_in) {
        HIP_TRY(hipEventCreateWithFlags(&ps.ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ps.ev_k, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ps.ev_out, hipEventDisableTiming));
    }
    if (ps.ticket >= 0) {  // the slot's previous submission must have left the device before its buffers are reused
        HIP_TRY(hipEventSynchronize(ps.ev_out));
        ps.ticket = -1;
    }
    const HostIo x = host_io(h, io, nullptr);
    auto need = [&](int i, size_t bytes) -> void* {
        auto& sc = ps.sc[i];
        if (sc.bytes < bytes) {
            if (sc.p) (void)hipFree(sc.p);
            sc.p = nullptr; sc.bytes = 0;
            if (hipMalloc(&sc.p, bytes) != hipSuccess) return nullptr;
            sc.bytes = bytes;
        }
        return sc.p;
    };
    // Zero-copy: when every buffer is pinned (device-visible) host memory the kernel writes qdot across
    // PCIe itself, and reads q the same way when nothing else is in flight (lowest latency: 83 us per C3
    // step, 1.8 MB each way).  While an earlier submission is still running, the inputs go through the
    // copy engine instead, which overlaps that kernel: 56 us per step at 2-3 in flight, against 66 us for
    // reads by the kernel and 95 us for the three-stream staging below.
    bool direct = true;
    for (int i = 0; i < N_HIN && direct; ++i) direct = !x.hin[i] || gpu_visible(x.hin[i]);
    for (int i = 0; i < N_HOUT && direct; ++i) direct = !x.hout[i] || gpu_visible(x.hout[i]);
    if (direct) {
        bool busy = false;
        for (auto& o : h->pipe)
            if (o.ticket >= 0 && hipEventQuery(o.ev_out) == hipErrorNotReady) busy = true;
        (void)hipGetLastError();
        vfik_io d = *io;
        if (busy) {
            void* din[N_HIN] = {};
            for (int i = 0; i < N_HIN; ++i)
                if (x.hin[i]) {
                    void* dp = need(i, x.bin[i]);
                    if (!dp) return fail(VFIK_E_HIP, "staging allocation failed");
                    HIP_TRY(hipMemcpyAsync(dp, x.hin[i], x.bin[i], hipMemcpyHostToDevice, h->s_in));
                    din[i] = dp;
                }
            d.q = din[0]; d.null_control = din[1]; d.q_ref = din[2]; d.q_cmded = din[3];
            d.active = static_cast<const int32_t*>(din[4]); d.q_lo = din[5]; d.q_hi = din[6];
            HIP_TRY(hipEventRecord(ps.ev_in, h->s_in));
            HIP_TRY(hipStreamWaitEvent(h->stream, ps.ev_in, 0));
        }
        const int rc = launch_cycles(h, &d, 0, 0.0, 0, nullptr, h->stream);
        if (rc != VFIK_OK) return rc;
        HIP_TRY(hipEventRecord(ps.ev_out, h->stream));
        ps.ticket = h->next_ticket;
        *ticket = h->next_ticket++;
        return VFIK_OK;
    }
    void* din[N_HIN] = {};
    void* dout[N_HOUT];
    for (int i = 0; i < N_HIN; ++i)
        if (x.hin[i] && !(din[i] = need(i, x.bin[i]))) return fail(VFIK_E_HIP, "staging allocation failed");
    for (int i = 0; i < N_HOUT; ++i) {
        dout[i] = x.hout[i] ? need(N_HIN + i, x.bout[i]) : nullptr;
        if (x.hout[i] && !dout[i]) return fail(VFIK_E_HIP, "staging allocation failed");
    }
    for (int i = 0; i < N_HIN; ++i)
        if (x.hin[i]) HIP_TRY(hipMemcpyAsync(din[i], x.hin[i], x.bin[i], hipMemcpyHostToDevice, h->s_in));
    if (io->active)  // gated arms store nothing: their rows must come back as they went in
        for (int i = 0; i < N_HOUT; ++i)
            if (x.hout[i]) HIP_TRY(hipMemcpyAsync(dout[i], x.hout[i], x.bout[i], hipMemcpyHostToDevice, h->s_in));
    HIP_TRY(hipEventRecord(ps.ev_in, h->s_in));
    HIP_TRY(hipStreamWaitEvent(h->stream, ps.ev_in, 0));
    vfik_io d;
    device_io(din, dout, d);
    const int rc = vfik_step(h, &d);
    if (rc != VFIK_OK) return rc;
    HIP_TRY(hipEventRecord(ps.ev_k, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->s_out, ps.ev_k, 0));
    for (int i = 0; i < N_HOUT; ++i)
        if (x.hout[i]) HIP_TRY(hipMemcpyAsync(x.hout[i], dout[i], x.bout[i], hipMemcpyDeviceToHost, h->s_out));
    HIP_TRY(hipEventRecord(ps.ev_out, h->s_out));
    ps.ticket = h->next_ticket;
    *ticket = h->next_ticket++;
    return VFIK_OK;
}

int vfik_track_reset(vfik_handle* h) {
    if (check_handle(h)) return VFIK_E_ARG;
    HIP_TRY(hipSetDevice(h->device));
    if (h->d_track) HIP_TRY(hipMemsetAsync(h->d_track, 0, (size_t)38 * h->B * sizeof(double), h->stream));
    return VFIK_OK;
}

int vfik_track_error(vfik_handle* h, const void* pose, const void* v6, void* out, const int32_t* active) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!pose || !v6 || !out) return fail(VFIK_E_ARG, "vfik_track_error: pose, v6 and out are required");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->d_track && dev_alloc(h, (void**)&h->d_track, (size_t)38 * h->B * sizeof(double), true)) return VFIK_E_HIP;
    hipError_t e = vfik::launch_track(h->io_dtype, pose, v6, h->d_track, out, active, h->B, h->stream);
    if (e != hipSuccess) return fail(VFIK_E_HIP, "track launch: %s", hipGetErrorString(e));
    return VFIK_OK;
}

int vfik_probe_field(vfik_handle* h, const void* pose, void* v6) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!pose || !v6) return fail(VFIK_E_ARG, "vfik_probe_field: pose and v6 are required");
    HIP_TRY(hipSetDevice(h->device));
    const double rs = h->params.rot_slowdown;
    const double cos_slow = rs > 0.0 && rs < 3.14159265358979323846 ? std::cos(rs) : -2.0;  // as kconst_fill
    hipError_t e = vfik::launch_probe(h->io_dtype, pose, h->d_goal, h->d_slots, h->B, h->Bpad, h->slots_used, rs, cos_slow, v6, h->stream);
    if (e != hipSuccess) return fail(VFIK_E_HIP, "probe launch: %s", hipGetErrorString(e));
    return VFIK_OK;
}

int vfik_object_distances(vfik_handle* h, const void* pose, const void* frames, int max_objects, void* out) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!pose || !frames || !out) return fail(VFIK_E_ARG, "vfik_object_distances: pose, frames and out are required");
    if (max_objects < 1 || max_objects > 4096) return fail(VFIK_E_ARG, "max_objects %d outside [1, 4096]", max_objects);
    HIP_TRY(hipSetDevice(h->device));
    hipError_t e = vfik::launch_monitor(h->io_dtype, pose, frames, max_objects, (long)h->B * max_objects, out, h->stream);
    if (e != hipSuccess) return fail(VFIK_E_HIP, "monitor launch: %s", hipGetErrorString(e));
    return VFIK_OK;
}

int vfik_set_objects(vfik_handle* h, int first_arm, int n_arms, const double* frames, int n_objects) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (quiesce(h) != VFIK_OK) return VFIK_E_HIP;
    if (!frames) return fail(VFIK_E_ARG, "null frames");
    if (n_objects < 1 || n_objects > 4096) return fail(VFIK_E_ARG, "n_objects %d outside [1, 4096]", n_objects);
    if (first_arm < 0 || n_arms < 1 || first_arm + n_arms > h->B) return fail(VFIK_E_ARG, "arm range [%d, %d) outside batch %d", first_arm, first_arm + n_arms, h->B);
    if (n_objects != h->n_objects && (first_arm != 0 || n_arms != h->B))
        return fail(VFIK_E_ARG, "n_objects changes from %d to %d: the call must cover every arm", h->n_objects, n_objects);
    const size_t count = (size_t)n_arms * n_objects * 16;
    for (size_t k = 0; k < count; ++k)
        if (!std::isfinite(frames[k])) return fail(VFIK_E_ARG, "object frame entry %zu is not finite", k);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (n_objects != h->n_objects) {
        if (h->d_objects) { (void)hipFree(h->d_objects); h->d_objects = nullptr; h->n_objects = 0; }
        if (dev_alloc(h, &h->d_objects, (size_t)h->B * n_objects * 16 * h->esz, true)) return VFIK_E_HIP;
        h->n_objects = n_objects;
    }
    std::vector<char> buf(count * h->esz);
    for (size_t k = 0; k < count; ++k) { if (h->io_dtype == 32) put<float>(buf, k, frames[k]); else put<double>(buf, k, frames[k]); }
    char* dst = static_cast<char*>(h->d_objects) + (size_t)first_arm * n_objects * 16 * h->esz;
    HIP_TRY(hipMemcpyAsync(dst, buf.data(), buf.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

int vfik_mix(vfik_handle* h, const void* cmds, const double* weights, int K, void* out) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!cmds || !weights || !out || K < 1 || K > 16) return fail(VFIK_E_ARG, "vfik_mix: bad arguments (K=%d)", K);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->d_mixw, weights, K * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const long count = (long)h->B * h->n;
    hipError_t e = vfik::launch_mix(h->io_dtype, cmds, h->d_mixw, K, count, count, out, h->stream);
    if (e != hipSuccess) return fail(VFIK_E_HIP, "mix launch: %s", hipGetErrorString(e));
    return VFIK_OK;
}

void* vfik_dev_alloc(vfik_handle* h, size_t bytes) {
    if (!h || bytes == 0) { fail(VFIK_E_ARG, "vfik_dev_alloc: bad arguments"); return nullptr; }
    void* p = nullptr;
    if (hipSetDevice(h->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { fail(VFIK_E_HIP, "hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}

int vfik_dev_free(vfik_handle* h, void* p) {
    if (check_handle(h)) return VFIK_E_ARG;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipFree(p));
    return VFIK_OK;
}

int vfik_memcpy_h2d(vfik_handle* h, void* dst, const void* src, size_t bytes) {
    if (check_handle(h)) return VFIK_E_ARG;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

int vfik_memcpy_d2h(vfik_handle* h, void* dst, const void* src, size_t bytes) {
    if (check_handle(h)) return VFIK_E_ARG;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VFIK_OK;
}

int vfik_time_steps(vfik_handle* h, const vfik_io* io, int warmup, int steps, float* ms_total) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (!ms_total || steps < 1 || warmup < 0) return fail(VFIK_E_ARG, "vfik_time_steps: bad arguments");
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < warmup; ++i) {
        int rc = vfik_step(h, io);
        if (rc != VFIK_OK) return rc;
    }
    hipEvent_t t0, t1;
    HIP_TRY(hipEventCreate(&t0));
    HIP_TRY(hipEventCreate(&t1));
    HIP_TRY(hipEventRecord(t0, h->stream));
    for (int i = 0; i < steps; ++i) {
        int rc = vfik_step(h, io);
        if (rc != VFIK_OK) { (void)hipEventDestroy(t0); (void)hipEventDestroy(t1); return rc; }
    }
    HIP_TRY(hipEventRecord(t1, h->stream));
    HIP_TRY(hipEventSynchronize(t1));
    HIP_TRY(hipEventElapsedTime(ms_total, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return VFIK_OK;
}

#ifdef VFIK_STAMPS
// diagnostic build: copy the per-wave section stamps ([waves][8]) to the host
int vfik_debug_read_stamps(vfik_handle* h, unsigned long long* dst) {
    if (check_handle(h)) return VFIK_E_ARG;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(dst, h->d_stamps, (size_t)((h->B + 63) / 64) * 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return VFIK_OK;
}
#endif

int vfik_slots_in_use(vfik_handle* h) { return h ? h->slots_used : VFIK_E_ARG; }

int vfik_field_path(vfik_handle* h) {
    if (!h) return VFIK_E_ARG;
    if (h->fast_order < 0) return 0;
    return h->any_funnel ? 2 : 1;
}

long vfik_launch_epoch(vfik_handle* h) { return h ? h->epoch : (long)VFIK_E_ARG; }

int vfik_dh_pattern(vfik_handle* h) { return h ? ((h->plain && !h->tool_per_arm && !h->d_wts) ? h->dhp : 0) : VFIK_E_ARG; }

int vfik_mixed_orders(vfik_handle* h) {
    if (!h) return VFIK_E_ARG;
    return (h->fast_order >= 0 && h->mixed) ? 1 : 0;
}

int vfik_uniform_repellers(vfik_handle* h) {
    if (!h) return VFIK_E_ARG;
    return (h->fast_order >= 0 && h->uni_ok && h->uni_allowed) ? 1 : 0;
}

int vfik_set_small_batch_kernel(vfik_handle* h, int max_batch) {
    if (check_handle(h)) return VFIK_E_ARG;
    if (max_batch < 0) return fail(VFIK_E_ARG, "max_batch must be >= 0");
    ++h->epoch;
    h->sub8_max_batch = max_batch;
    h->sub8_max_batch_full = max_batch;
    h->sub8_max_batch_ns = max_batch;
    return VFIK_OK;
}

long vfik_small_batch_launches(vfik_handle* h) { return h ? h->sub8_launches : -1; }

size_t vfik_device_bytes(vfik_handle* h) { return h ? h->dev_bytes : 0; }

}  // extern "C"
