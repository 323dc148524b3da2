// ubench_launch.hip -- the floor under a small-batch control cycle: launch period of back-to-back kernels on one stream
// that do (a) nothing, (b) one dependent global load + store per lane (a lone wave's memory round trip), (c) the same with
// a kernarg-indirect constant block read first (what cycle_kernel's prologue does), for 1 .. 1024 one-wave workgroups.
// The small-batch kernels of libvfik_hip.so cannot be faster than (b)/(c) plus their arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(64) k_empty() {}

__global__ void __launch_bounds__(64) k_load_store(const float* src, float* dst) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    dst[i] = src[i] + 1.0f;
}

struct Args { const float* src; float* dst; const double* kc; int pad[20]; };
__global__ void __launch_bounds__(64) k_const_load_store(const Args a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    typedef const double __attribute__((address_space(4))) * CP;
    const CP kc = (CP)(unsigned long long)a.kc;
    const double c = kc[3];            // scalar load of a batch constant (a dependent round trip through the scalar cache)
    a.dst[i] = (float)((double)a.src[i] + c);
}

template <typename F>
static double period_us(F launch, hipStream_t s, int n = 2000, int reps = 7) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> v;
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < 50; ++i) launch();
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        v.push_back(ms * 1e3 / n);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

struct Big { const float* src; float* dst; char pad[320]; };   // a kernarg block the size of vfik::KArgs
template <int N> struct Sized { const float* src; float* dst; char pad[N - 16]; };
template <int N> __global__ void __launch_bounds__(64) k_sized(const Sized<N> a) {
    if (a.pad[0]) a.dst[threadIdx.x] = 0.0f;
}
__global__ void __launch_bounds__(64) k_big(const Big a) {
    if (a.pad[0]) a.dst[threadIdx.x] = 0.0f;
}

// what one launch costs the HOST thread: n launches into an idle stream, clock stopped BEFORE the synchronize
template <typename F>
static double enqueue_us(F launch, hipStream_t s, int n = 200, int reps = 15) {
    std::vector<double> v;
    for (int r = 0; r < reps; ++r) {
        hipStreamSynchronize(s);
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < n; ++i) launch();
        clock_gettime(CLOCK_MONOTONIC, &t1);
        hipStreamSynchronize(s);
        v.push_back(((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / n);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float *src, *dst; double* kc;
    hipMalloc(&src, 1024 * 64 * 4); hipMalloc(&dst, 1024 * 64 * 4); hipMalloc(&kc, 1024);
    hipMemset(src, 0, 1024 * 64 * 4); hipMemset(kc, 0, 1024);
    printf("launch period of back-to-back kernels on one stream, us (median of 7 x 2000 launches, HIP events)\n");
    printf("%10s %10s %14s %20s\n", "workgroups", "empty", "load+store", "const+load+store");
    for (int g : {1, 8, 64, 256, 512, 1024}) {
        const double a = period_us([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(64), 0, s); }, s);
        const double b = period_us([&] { hipLaunchKernelGGL(k_load_store, dim3(g), dim3(64), 0, s, src, dst); }, s);
        Args ar{src, dst, kc, {0}};
        const double c = period_us([&] { hipLaunchKernelGGL(k_const_load_store, dim3(g), dim3(64), 0, s, ar); }, s);
        printf("%10d %10.3f %14.3f %20.3f\n", g, a, b, c);
    }
    // host-side cost of one launch (the enqueue loop of bench.py pays this per step)
    {
        Big big{src, dst, {0}};
        hipFunction_t f_big = nullptr;
        hipError_t e = hipGetFuncBySymbol(&f_big, reinterpret_cast<const void*>(&k_big));
        printf("\nhost enqueue cost per launch (200 launches into an idle stream, median of 15), 1024 workgroups:\n");
        printf("  hipLaunchKernelGGL, no arguments          %.2f us\n", enqueue_us([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, s); }, s));
        printf("  hipLaunchKernelGGL, 336-byte kernarg      %.2f us\n", enqueue_us([&] { hipLaunchKernelGGL(k_big, dim3(1024), dim3(64), 0, s, big); }, s));
        {
            Sized<32> a32{src, dst, {0}}; Sized<64> a64{src, dst, {0}}; Sized<128> a128{src, dst, {0}}; Sized<192> a192{src, dst, {0}}; Sized<256> a256{src, dst, {0}};
            printf("  by kernarg size: 32 B %.2f, 64 B %.2f, 128 B %.2f, 192 B %.2f, 256 B %.2f us\n",
                   enqueue_us([&] { hipLaunchKernelGGL(k_sized<32>, dim3(1024), dim3(64), 0, s, a32); }, s),
                   enqueue_us([&] { hipLaunchKernelGGL(k_sized<64>, dim3(1024), dim3(64), 0, s, a64); }, s),
                   enqueue_us([&] { hipLaunchKernelGGL(k_sized<128>, dim3(1024), dim3(64), 0, s, a128); }, s),
                   enqueue_us([&] { hipLaunchKernelGGL(k_sized<192>, dim3(1024), dim3(64), 0, s, a192); }, s),
                   enqueue_us([&] { hipLaunchKernelGGL(k_sized<256>, dim3(1024), dim3(64), 0, s, a256); }, s));
            printf("  empty kernel, grid 1 / 1024 workgroups: %.2f / %.2f us\n",
                   enqueue_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, s),
                   enqueue_us([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, s); }, s));
        }
        if (e == hipSuccess && f_big) {
            size_t sz = sizeof(big);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &big, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            printf("  hipModuleLaunchKernel (function resolved once, packed kernarg) %.2f us\n",
                   enqueue_us([&] { hipModuleLaunchKernel(f_big, 1024, 1, 1, 64, 1, 1, 0, s, nullptr, cfg); }, s));
        } else {
            printf("  hipGetFuncBySymbol failed: %s\n", hipGetErrorString(e));
        }
        printf("  + hipSetDevice + hipGetLastError per launch %.2f us\n",
               enqueue_us([&] { hipSetDevice(0); hipLaunchKernelGGL(k_big, dim3(1024), dim3(64), 0, s, big); (void)hipGetLastError(); }, s));
        // a graph of 20 kernel nodes replayed: per-node host cost
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_big, dim3(1024), dim3(64), 0, s, big);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("  hipGraphLaunch of 20 captured launches     %.2f us per captured launch (host)\n", enqueue_us([&] { hipGraphLaunch(ge, s); }, s, 10) / 20.0);
        printf("  ... and its GPU period                      %.2f us per captured launch\n", period_us([&] { hipGraphLaunch(ge, s); }, s, 100) / 20.0);
    }
    return 0;
}
