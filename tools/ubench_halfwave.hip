// ubench_halfwave.hip -- does a wave64 whose upper 32 lanes are masked off issue float64 VALU work in
// half the time on gfx950 (SIMD-32)?  If so, two half-full waves per SIMD would do the work of one full
// wave with twice the latency hiding.  Also: aggregate fp64 FMA throughput of 1 vs 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define N 512
template <int CHAINS>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, double seed, int active_lanes) {
    if ((int)threadIdx.x >= active_lanes) return;
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = seed + threadIdx.x * 1e-3 + c;
    const double a = 1.0000001, b = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N / 8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS> void run(const char* name, int waves, int lanes, double* out, unsigned long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<CHAINS>), dim3(waves), dim3(64), 0, 0, out, cyc, 1.5, lanes);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS>), dim3(waves), dim3(64), 0, 0, out, cyc, 1.5, lanes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-34s chains=%d waves=%5d lanes=%2d: median %7.0f ticks/wave = %5.2f per fma; kernel %.2f us\n", name, CHAINS, waves, lanes,
           (double)h[waves / 2], (double)h[waves / 2] / (N * CHAINS), ms * 1e3);
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8192 * 64 * 8); hipMalloc(&cyc, 8192 * 8);
    run<1>("1 wave/SIMD, full", 1024, 64, out, cyc);
    run<1>("1 wave/SIMD, half-masked", 1024, 32, out, cyc);
    run<1>("2 waves/SIMD, full", 2048, 64, out, cyc);
    run<1>("2 waves/SIMD, half-masked", 2048, 32, out, cyc);
    run<1>("4 waves/SIMD, full", 4096, 64, out, cyc);
    run<4>("1 wave/SIMD, full", 1024, 64, out, cyc);
    run<4>("1 wave/SIMD, half-masked", 1024, 32, out, cyc);
    run<4>("2 waves/SIMD, full", 2048, 64, out, cyc);
    run<4>("2 waves/SIMD, half-masked", 2048, 32, out, cyc);
    run<4>("4 waves/SIMD, full", 4096, 64, out, cyc);
    return 0;
}
