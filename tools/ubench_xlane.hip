// ubench_xlane.hip -- what it costs ONE wave (alone on its SIMD: a small-batch launch) to combine float64 values across the
// 8 lanes that share an arm in the lanes-per-arm mapping, against doing the arithmetic on every lane:
//   (a) a 7-term dot product computed on every lane (replicated: 1 mul + 6 fma),
//   (b) the same sum as a cross-lane reduction over 8 lanes with DPP moves (3 stages of 2 x v_mov_b32_dpp + v_add_f64;
//       gfx950 has no DPP form of the 64-bit VALU ops, so every exchanged double is two 32-bit moves),
//   (c) the same through LDS (ds_write_b64, wait, 7 x ds_read_b64 + adds),
// for N independent values at a time (the scheduler interleaves them): cycles per value, s_memtime over 64 repetitions.
// The nullspace block needs 21 such sums per arm (6 norms + 15 projections of the Gram-Schmidt pass).
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double dpp_xor1(double x) {   // lane ^ 1 within a quad: quad_perm [1,0,3,2]
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_xor2(double x) {   // lane ^ 2 within a quad: quad_perm [2,3,0,1]
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_half_mirror(double x) {   // row_half_mirror: lane i <-> 7 - i within each group of 8
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double reduce8(double x) {  // sum over the 8 lanes of a group, result on every lane
    x += dpp_xor1(x);
    x += dpp_xor2(x);
    x += dpp_half_mirror(x);
    return x;
}

template <int MODE, int N>
__global__ void __launch_bounds__(64) k(const double* src, double* out, unsigned long long* cyc) {
    __shared__ double lds[64 * 8];
    const int lane = threadIdx.x;
    double a[N][7], b[7], acc[N];
    for (int i = 0; i < 7; ++i) b[i] = src[lane * 7 + i];
    for (int n = 0; n < N; ++n) { acc[n] = 0.0; for (int i = 0; i < 7; ++i) a[n][i] = src[(n + 1) * 448 + lane * 7 + i]; }
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int rep = 0; rep < 64; ++rep) {
        if (MODE == 0) {          // replicated dot products
#pragma unroll
            for (int n = 0; n < N; ++n) {
                double s = a[n][0] * b[0];
#pragma unroll
                for (int i = 1; i < 7; ++i) s = __builtin_fma(a[n][i], b[i], s);
                acc[n] += s;
            }
        } else if (MODE == 1) {   // one product per lane, DPP reduction
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] += reduce8(a[n][0] * b[0]);
        } else {                  // one product per lane, exchange through LDS
#pragma unroll
            for (int n = 0; n < N; ++n) lds[n * 64 + lane] = a[n][0] * b[0];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < N; ++n) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < 8; ++l) s += lds[n * 64 + (lane & 56) + l];
                acc[n] += s;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(b[i]));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += acc[n];
    out[lane] = s;
    if (lane == 0) cyc[0] = t1 - t0;
}

template <int MODE, int N>
static void run(const char* name, const double* src, double* out, unsigned long long* cyc) {
    unsigned long long h = 0, best = ~0ull;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL((k<MODE, N>), dim3(1), dim3(64), 0, 0, src, out, cyc);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (h < best) best = h;
    }
    printf("  %-44s N = %d: %6.1f cycles per value\n", name, N, (double)best / 64.0 / N);
}

int main() {
    double *src, *out;
    unsigned long long* cyc;
    hipMalloc(&src, 16 * 448 * 8); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    hipMemset(src, 0, 16 * 448 * 8);
    printf("one wave alone on its SIMD, float64, 8 lanes per arm: cost of a 7-term sum (cycles, s_memtime)\n");
    run<0, 1>("replicated 7-term dot product on every lane", src, out, cyc);
    run<0, 5>("replicated 7-term dot product on every lane", src, out, cyc);
    run<1, 1>("DPP reduction over 8 lanes (3 stages)", src, out, cyc);
    run<1, 5>("DPP reduction over 8 lanes (3 stages)", src, out, cyc);
    run<2, 1>("LDS exchange (write, wait, 8 reads, adds)", src, out, cyc);
    run<2, 5>("LDS exchange (write, wait, 8 reads, adds)", src, out, cyc);
    return 0;
}
