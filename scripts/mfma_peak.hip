// Issue-rate probes on gfx950: v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64, v_mfma_f32_16x16x4_f32, v_fma_f64.
// Prints TF/s and the in-kernel clock (s_memtime / s_memrealtime) so cycles per instruction can be told
// apart from DVFS throttling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, rt; };
#define STAMP_BEGIN unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP_END   unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
template <int NACC>
__global__ void __launch_bounds__(256) k64(double* out, Stamp* st, int iters, double a0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    STAMP_END
}
template <int NACC>
__global__ void __launch_bounds__(256) k64s(double* out, Stamp* st, int iters, double a0) {  // 4x4x4, 4 blocks
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = a0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    STAMP_END
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(double* out, Stamp* st, int iters, float a0) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    STAMP_END
}
template <int NACC>
__global__ void __launch_bounds__(256) kfma(double* out, Stamp* st, int iters, double a0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = 1e-3;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    STAMP_END
}
static double* o; static Stamp* st; static Stamp hst[4096];
template <typename F>
static void run(const char* name, int grid, double flop_per_thread_iter_total, int iters, double insts_per_wave, F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hst, st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < grid; ++i) { ghz.push_back((double)hst[i].cyc / (double)hst[i].rt * 0.1); cyc.push_back((double)hst[i].cyc); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double wps = grid / 256.0;  // waves per SIMD
    printf("%-22s grid=%4d: %7.2f TF/s  clock %.2f GHz  %.1f cyc/inst/SIMD (in-kernel)\n", name, grid,
           flop_per_thread_iter_total / (ms * 1e-3) / 1e12, ghz[grid / 2], cyc[grid / 2] / (insts_per_wave * wps));
}
int main() {
    (void)hipMalloc(&o, 4096 * 256 * 8); (void)hipMalloc(&st, sizeof(Stamp) * 4096);
    const int iters = 20000;
    for (int grid : {256, 512, 1024, 2048}) {
        const double waves = grid * 4.0;
        run("mfma_f64_16x16x4 x1", grid, waves * iters * 2048.0, iters, iters, [&] { hipLaunchKernelGGL(k64<1>, dim3(grid), dim3(256), 0, 0, o, st, iters, 0.5); });
        run("mfma_f64_16x16x4 x4", grid, waves * iters * 4 * 2048.0, iters, iters * 4.0, [&] { hipLaunchKernelGGL(k64<4>, dim3(grid), dim3(256), 0, 0, o, st, iters, 0.5); });
        run("mfma_f64_4x4x4_4b x4", grid, waves * iters * 4 * 512.0, iters, iters * 4.0, [&] { hipLaunchKernelGGL(k64s<4>, dim3(grid), dim3(256), 0, 0, o, st, iters, 0.5); });
        run("mfma_f32_16x16x4 x4", grid, waves * iters * 4 * 2048.0, iters, iters * 4.0, [&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, o, st, iters, 0.5f); });
        run("v_fma_f64 x8", grid, waves * iters * 8 * 128.0, iters, iters * 8.0, [&] { hipLaunchKernelGGL(kfma<8>, dim3(grid), dim3(256), 0, 0, o, st, iters, 0.999); });
    }
    return 0;
}
