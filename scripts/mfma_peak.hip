// Back-to-back v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 issue-rate probe (one or two waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k64(double* out, int iters, double a0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a0) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static double timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}
int main() {
    double* o; hipMalloc(&o, 1024 * 256 * 8);
    const int iters = 20000;
    for (int grid : {256, 512, 1024}) {
        double t1 = timeit([&] { hipLaunchKernelGGL(k64<1>, dim3(grid), dim3(256), 0, 0, o, iters, 0.5); });
        double t4 = timeit([&] { hipLaunchKernelGGL(k64<4>, dim3(grid), dim3(256), 0, 0, o, iters, 0.5); });
        double f1 = (double)grid * 4 * iters * 2048.0, f4_ = f1 * 4;
        printf("f64 16x16x4 grid=%4d: 1 acc %.2f TF/s (%.1f cyc/mfma/SIMD @2.4GHz)   4 acc %.2f TF/s\n", grid,
               f1 / t1 / 1e12, t1 * 2.4e9 / (iters * (grid / 256.0)), f4_ / t4 / 1e12);
        double s1 = timeit([&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, (float*)o, iters, 0.5f); });
        double s4 = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, (float*)o, iters, 0.5f); });
        printf("f32 16x16x4 grid=%4d: 1 acc %.2f TF/s   4 acc %.2f TF/s\n", grid, f1 / s1 / 1e12, f4_ / s4 / 1e12);
    }
    return 0;
}
