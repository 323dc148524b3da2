// ubench_fp64.hip -- issue / dependent-latency cost of the float64 instructions the control-cycle
// kernel is made of, measured with s_memtime on one wave per SIMD (the C3 launch geometry).
// Build: hipcc -O3 --offload-arch=gfx950 ubench_fp64.hip -o ubench_fp64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define N 256

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* cyc, double seed) {
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = seed + threadIdx.x * 1e-3 + c;
    const double a = 1.0000001, b = 1e-9;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N / 8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == 0) x[c] = __builtin_fma(x[c], a, b);
                if (MODE == 1) x[c] = x[c] * a;
                if (MODE == 2) x[c] = x[c] + b;
                if (MODE == 3) x[c] = __builtin_amdgcn_rcp(x[c]);
                if (MODE == 4) x[c] = __builtin_amdgcn_rsq(x[c]);
                if (MODE == 5) x[c] = 1.0 / x[c];
                if (MODE == 6) x[c] = sqrt(x[c]);
                if (MODE == 7) { double s, co; sincos(x[c], &s, &co); x[c] = s + co; }
                if (MODE == 8) x[c] = atan2(x[c], a);
                if (MODE == 9) x[c] = pow(x[c], 2.5);
                if (MODE == 10) { float f = (float)x[c]; f = __builtin_fmaf(f, 1.0000001f, 1e-9f); x[c] = f; }
                if (MODE == 11) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(x[c])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[c]) : "v"(f)); }
                if (MODE == 12) x[c] = fmin(x[c], a);
                if (MODE == 13) x[c] = x[c] > a ? b : x[c] + 1.0;
                if (MODE == 14) { double r = __builtin_amdgcn_rsq(x[c]); x[c] = __builtin_fma(r, 0.5, 1.0); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int MODE, int CHAINS>
void run(const char* name, double* out, unsigned long long* cyc) {
    const int grid = 256, block = 256, waves = grid * block / 64;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(grid), dim3(block), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[waves / 2];
    printf("%-10s chains=%d  median %8.0f ticks/wave -> %6.2f ticks per op (%6.2f per op-group)\n", name, CHAINS, med,
           med / (N * CHAINS), med / N);
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(double));
    hipMalloc(&cyc, 1024 * sizeof(unsigned long long));
#define R(M, NAME) run<M, 1>(NAME, out, cyc); run<M, 2>(NAME, out, cyc); run<M, 4>(NAME, out, cyc); run<M, 8>(NAME, out, cyc);
    R(0, "fma_f64") R(1, "mul_f64") R(2, "add_f64") R(3, "rcp_f64") R(4, "rsq_f64") R(5, "div_f64") R(6, "sqrt_f64")
    R(7, "sincos") R(8, "atan2") R(9, "pow") R(10, "fma_f32cv") R(11, "cvt32+cvt64") R(12, "min_f64") R(13, "cmp+cndmask+add") R(14, "rsq+fma")
    // s_memtime tick calibration: ticks per microsecond
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(256), 0, 0, out, cyc, 1.5); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("one fma launch: %.2f us wall\n", ms * 1e3);
    return 0;
}
