// ubench_loads.hip -- issue cost (cycles the issuing wave spends) of the load forms the control-cycle
// kernel can use to bring a lane's records in: global_load_dword / dwordx4 to VGPRs and
// global_load_lds_dword / dwordx4 (LDS-DMA), 1 wave per SIMD, data L2-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* GPtr;
typedef __attribute__((address_space(3))) void* LPtr;
#define NL 16
template <int MODE>
__global__ void __launch_bounds__(64) k(const float* src, float* out, unsigned long long* cyc, long plane) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x, arm = blockIdx.x * 64 + lane;
    const char* g = (const char*)src + (size_t)arm * 16;
    f4 v[NL]; float s[NL];
    unsigned long long t0, t1, t2;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(g + (size_t)i * plane) : "memory");
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("global_load_dword %0, %1, off" : "=v"(s[i]) : "v"(g + (size_t)i * plane) : "memory");
    } else if (MODE == 4) {
        const unsigned voff = (unsigned)arm * 16u;
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[i]) : "v"(voff), "s"((const char*)src + (size_t)i * plane) : "memory");
    } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NL; ++i) __builtin_amdgcn_global_load_lds((GPtr)(g + (size_t)i * plane), (LPtr)(lds + i * 1024), 16, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) __builtin_amdgcn_global_load_lds((GPtr)(g + (size_t)i * plane), (LPtr)(lds + i * 1024), 4, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
    float acc = 0;
    if (MODE == 0 || MODE == 4) { for (int i = 0; i < NL; ++i) { asm volatile("" : "+v"(v[i])); acc += v[i].x + v[i].w; } }
    else if (MODE == 1) { for (int i = 0; i < NL; ++i) { asm volatile("" : "+v"(s[i])); acc += s[i]; } }
    else { for (int i = 0; i < NL; ++i) acc += *(const float*)(lds + i * 1024 + lane * 16); }
    out[arm] = acc;
    if (lane == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = t2 - t0; }
}
template <int MODE> void run(const char* name, const float* src, float* out, unsigned long long* cyc, long plane, int grid = 1024) {
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), NL * 1024, 0, src, out, cyc, plane);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < grid; ++i) { a.push_back((double)h[2 * i]); b.push_back((double)h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-26s issue %6.0f ticks for %d loads (%5.1f each); issue+landed %6.0f ticks\n", name, a[grid / 2], NL, a[grid / 2] / NL, b[grid / 2]);
}
int main() {
    const long B = 65536, plane = B * 16;
    float *src, *out; unsigned long long* cyc;
    hipMalloc(&src, plane * NL); hipMalloc(&out, B * 4); hipMalloc(&cyc, 2048 * 8);
    hipMemset(src, 0, plane * NL);
    run<0>("global_load_dwordx4", src, out, cyc, plane);
    run<1>("global_load_dword", src, out, cyc, plane);
    run<2>("global_load_lds_dwordx4", src, out, cyc, plane);
    run<3>("global_load_lds_dword", src, out, cyc, plane);
    run<4>("dwordx4 saddr+voffset", src, out, cyc, plane);
    printf("-- 256 blocks (one wave per CU)\n");
    run<0>("global_load_dwordx4", src, out, cyc, plane, 256);
    run<2>("global_load_lds_dwordx4", src, out, cyc, plane, 256);
    run<4>("dwordx4 saddr+voffset", src, out, cyc, plane, 256);
    return 0;
}
