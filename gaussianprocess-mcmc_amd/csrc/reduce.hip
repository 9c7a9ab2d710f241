// reduce.hip -- O(N^2)/O(N) companions of the factorisation (8(b) packing).
#include "gpmc_common.h"

namespace gpmc {

// ----------------------------------------------------------------------- log-marginal reduction
// curG / propG of kcMCMC/sliceSample.py:122,147 from the factor: -(0.5 a.a + sum log L_ii + n/2 log 2pi)
// One workgroup, fixed summation order (bitwise reproducible), f64 accumulation for both types.
template <typename T>
__global__ void __launch_bounds__(1024)
logdet_quad_kernel(const T* __restrict__ W, int n, int ldw, const T* __restrict__ alpha,
                   const int* __restrict__ info, double* __restrict__ out, double* __restrict__ host_base,
                   const T* __restrict__ slot_dev, unsigned* __restrict__ seq, int ring_stride) {
    __shared__ double s_ld[1024], s_q[1024];
    const int tid = threadIdx.x;
    double ld = 0.0, qd = 0.0;
    for (int i = tid; i < n; i += 1024) {
        ld += log((double)W[(size_t)i * ldw + i]);
        const double a = (double)alpha[i];
        qd += a * a;
    }
    s_ld[tid] = ld;
    s_q[tid] = qd;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (tid < st) {
            s_ld[tid] += s_ld[tid + st];
            s_q[tid] += s_q[tid + st];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double LOG_2PI = 1.8378770664093454835606594728112;
        out[0] = -(0.5 * s_q[0] + s_ld[0] + 0.5 * (double)n * LOG_2PI);
        out[1] = s_ld[0];
        out[2] = s_q[0];
        out[3] = info ? (double)info[0] : 0.0;
        if (host_base != nullptr) {      // pinned, device-mapped: visible to the host once this kernel has completed
            // (with a parameter ring the slot index sits in block (*seq & 63) of the ring, and this evaluation -- the ring
            //  block's last reader -- moves the sequence number on)
            const unsigned k = seq ? *seq : 0u;
            const T* blk = slot_dev + (size_t)(k & 63u) * ring_stride;
            double* h = host_base + 4 * (int)blk[0];
            // (the block names the execution it was written for, right behind the slot index: a block of another execution
            //  means host and device lost step, and the host gets that instead of a value computed from the wrong parameters)
            if (seq && (unsigned)blk[1] != (k & 0xfffffu)) out[3] = (double)GPMC_RING_OUT_OF_STEP;
            for (int j = 0; j < 4; ++j) h[j] = out[j];
            __threadfence_system();
            if (seq) {
                *seq = k + 1u;
                // the execution's number behind its result slot, written last: the host may poll this word instead of
                // waiting on an event (gpmc_plan_result)
                host_base[4 * 64 + (int)blk[0]] = (double)(k & 0xfffffu);
                __threadfence_system();
            }
        }
    }
}

template <typename T>
int logdet_quad(const T* W, int n, int ldw, const T* alpha, const int* info_dev, double* out_dev, hipStream_t s,
                double* host_base, const T* slot_dev, unsigned* seq, int ring_stride) {
    GPMC_ARG(W && n > 0 && ldw >= n, 1);
    GPMC_ARG(alpha != nullptr, 4);
    GPMC_ARG(out_dev != nullptr, 6);
    hipLaunchKernelGGL((logdet_quad_kernel<T>), dim3(1), dim3(1024), 0, s, W, n, ldw, alpha, info_dev, out_dev,
                       (host_base != nullptr && slot_dev != nullptr) ? host_base : nullptr, slot_dev, seq, ring_stride);
    GPMC_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------- packing
template <typename T>
__global__ void pack_lower_kernel(const T* __restrict__ A, int n, int lda, T jitter, T* __restrict__ W, int np_,
                                  int ldw) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= np_) return;
    T v = (T)0;
    if (j <= i) {
        if (i < n) v = A[(size_t)i * lda + j] + ((i == j) ? jitter : (T)0);
        else v = (i == j) ? (T)1 : (T)0;
    }
    W[(size_t)i * ldw + j] = v;
}
template <typename T>
__global__ void pack_rows_kernel(const T* __restrict__ B, int nrhs, int n, int ldb, T* __restrict__ W, int np_,
                                 int ldw) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (j >= np_) return;
    W[(size_t)(np_ + r) * ldw + j] = (r < nrhs && j < n) ? B[(size_t)r * ldb + j] : (T)0;
}
template <typename T>
__global__ void unpack_lower_kernel(const T* __restrict__ W, int ldw, T* __restrict__ L, int n, int ldl) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    L[(size_t)i * ldl + j] = (j <= i) ? W[(size_t)i * ldw + j] : (T)0;
}
template <typename T>
__global__ void unpack_rows_kernel(const T* __restrict__ W, int np_, int ldw, T* __restrict__ B, int n, int ldb) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (j >= n) return;
    B[(size_t)r * ldb + j] = W[(size_t)(np_ + r) * ldw + j];
}

// rows (np_ x np_, ld) <- upper-trapezoidal s I: the diagonal becomes *diag_dev, everything right of it zero, the strict lower
// part is NOT touched.  The riding block of gpmc_potrf_tri comes back as P = s L^-T, zero left of the diagonal and never
// written there, so re-arming it for the next proposal only has to clear the upper triangle: half of the full memset
// (268 instead of 537 MB at N = 8192), one launch instead of a fill and a diagonal scatter, and the diagonal value comes
// from device memory, so the launch can sit in a recorded sequence.  One workgroup per 64-row x 256-column piece.
template <typename T>
__global__ void __launch_bounds__(256)
reset_tri_kernel(T* __restrict__ R, int np_, int ld, const T* __restrict__ diag_dev) {
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 256;
    if (c0 + 256 <= r0) return;                      // entirely left of the diagonal
    const T dv = *diag_dev;
    const int c = c0 + threadIdx.x;
    for (int r = r0; r < r0 + 64; ++r)
        if (c >= r && c < np_) R[(size_t)r * ld + c] = (c == r) ? dv : (T)0;
}
template <typename T>
int reset_tri(T* R, int np_, int ld, const T* diag_dev, hipStream_t s) {
    GPMC_ARG(R != nullptr && np_ > 0 && np_ % 64 == 0 && ld >= np_, 1);
    GPMC_ARG(diag_dev != nullptr, 4);
    hipLaunchKernelGGL((reset_tri_kernel<T>), dim3((np_ + 255) / 256, np_ / 64), dim3(256), 0, s, R, np_, ld, diag_dev);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template int reset_tri<double>(double*, int, int, const double*, hipStream_t);
template int reset_tri<float>(float*, int, int, const float*, hipStream_t);

// dst[0:n] = src[0:n] as a kernel (a recorded sequence then holds no memcpy node: kernels and memsets only, like the plan's)
template <typename T>
__global__ void __launch_bounds__(256) veccopy_kernel(T* __restrict__ dst, const T* __restrict__ src, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
template <typename T>
int veccopy(T* dst, const T* src, int n, hipStream_t s) {
    GPMC_ARG(dst != nullptr && src != nullptr && n > 0, 1);
    hipLaunchKernelGGL((veccopy_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template int veccopy<double>(double*, const double*, int, hipStream_t);
template int veccopy<float>(float*, const float*, int, hipStream_t);

// The results of one SDS proposal straight into pinned, device-mapped host memory (no copy node after the sequence):
// host[0..3] = out4 (G, log det, quadratic form, info of the first factorisation), host[4] = *info2, host[8 .. 8 + n) = vec.
template <typename T>
__global__ void __launch_bounds__(256)
publish_host_kernel(const double* __restrict__ out4, const int* __restrict__ info2, const T* __restrict__ vec, int n,
                    double* __restrict__ host) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) host[8 + i] = (double)vec[i];
    if (i < 4) host[i] = out4[i];
    if (i == 4) host[4] = (double)info2[0];
    __threadfence_system();
}
template <typename T>
int publish_host(const double* out4, const int* info2, const T* vec, int n, double* host_mapped, hipStream_t s) {
    GPMC_ARG(out4 != nullptr && info2 != nullptr && vec != nullptr && n > 0, 1);
    GPMC_ARG(host_mapped != nullptr, 5);
    hipLaunchKernelGGL((publish_host_kernel<T>), dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, s, out4, info2,
                       vec, n, host_mapped);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template int publish_host<double>(const double*, const int*, const double*, int, double*, hipStream_t);
template int publish_host<float>(const double*, const int*, const float*, int, double*, hipStream_t);

// The SDS proposal's small steps between R = -P P^T (gpmc_syrk_tri2) and chol(R + 1e-11 I) in ONE launch (they were four
// elementwise launches of ~5 us each, a fifth of a proposal at the reference's demo sizes): diag(R) += *shift_dev (the s of
// s I - P P^T, read on the device), diag_out = that diagonal (the jitter schedule's yardstick and what restores R), diag(R) +=
// jitter (the 1e-11 of sliceSample.py:205), and the riding row row_out = f - m (sliceSample.py:206).  The same additions in
// the same order as the launches it replaces: the same bits.
template <typename T>
__global__ void __launch_bounds__(256)
shift_diag_sub_kernel(T* __restrict__ R, int ldr, int n, const T* __restrict__ shift_dev, T jitter, T* __restrict__ diag_out,
                      const T* __restrict__ f, const T* __restrict__ m, T* __restrict__ row_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T d = R[(size_t)i * ldr + i] + shift_dev[0];
    diag_out[i] = d;
    R[(size_t)i * ldr + i] = d + jitter;
    row_out[i] = f[i] - m[i];
}
template <typename T>
int shift_diag_sub(T* R, int ldr, int n, const T* shift_dev, T jitter, T* diag_out, const T* f, const T* m, T* row_out,
                   hipStream_t s) {
    GPMC_ARG(R != nullptr && n > 0 && ldr >= n, 1);
    GPMC_ARG(shift_dev != nullptr && diag_out != nullptr, 4);
    GPMC_ARG(f != nullptr && m != nullptr && row_out != nullptr, 7);
    hipLaunchKernelGGL((shift_diag_sub_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, R, ldr, n, shift_dev, jitter,
                       diag_out, f, m, row_out);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template int shift_diag_sub<double>(double*, int, int, const double*, double, double*, const double*, const double*, double*, hipStream_t);
template int shift_diag_sub<float>(float*, int, int, const float*, float, float*, const float*, const float*, float*, hipStream_t);

template <typename T>
int pack_lower(const T* A, int n, int lda, T jitter, T* W, int np_, int ldw, hipStream_t s) {
    GPMC_ARG(A && n > 0 && lda >= n, 1);
    GPMC_ARG(W && np_ >= n && np_ % GPMC_PAD == 0 && ldw >= np_, 5);
    hipLaunchKernelGGL((pack_lower_kernel<T>), dim3((np_ + 255) / 256, np_), dim3(256), 0, s, A, n, lda, jitter, W,
                       np_, ldw);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int pack_rows(const T* B, int nrhs, int n, int ldb, T* W, int mp, int np_, int ldw, hipStream_t s) {
    GPMC_ARG(nrhs >= 0 && (nrhs == 0 || (B && ldb >= n)), 1);
    GPMC_ARG(W && mp >= np_ + nrhs && ldw >= np_, 5);
    if (mp == np_) return 0;
    hipLaunchKernelGGL((pack_rows_kernel<T>), dim3((np_ + 255) / 256, mp - np_), dim3(256), 0, s, B, nrhs, n, ldb,
                       W, np_, ldw);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int unpack_lower(const T* W, int np_, int ldw, T* L, int n, int ldl, hipStream_t s) {
    GPMC_ARG(W && L && n > 0 && n <= np_ && ldl >= n && ldw >= np_, 1);
    hipLaunchKernelGGL((unpack_lower_kernel<T>), dim3((n + 255) / 256, n), dim3(256), 0, s, W, ldw, L, n, ldl);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int unpack_rows(const T* W, int np_, int ldw, T* B, int nrhs, int n, int ldb, hipStream_t s) {
    GPMC_ARG(W && B && nrhs > 0 && n > 0 && n <= np_ && ldb >= n, 1);
    hipLaunchKernelGGL((unpack_rows_kernel<T>), dim3((n + 255) / 256, nrhs), dim3(256), 0, s, W, np_, ldw, B, n, ldb);
    GPMC_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------- matrix-vector products
// symv_lower: y = A x from the lower triangle only, in two launches with a fixed summation order:
//   1. row part     y[i]  = sum_{j <= i} A[i][j] x[j]   -- trmv_lower_kernel, one wave per row
//   2. column part  y[i] += sum_{j >  i} A[j][i] x[j]   -- symv_cols_kernel below: workgroup b owns columns
//      [64b, 64b+64) (so every y[i] has exactly one writer), 16 waves stride the rows below, lane = column (512-byte
//      coalesced row segments), 8 independent partial sums per lane.
// Every lower element is read twice in total (HBM-bound, 2 * s * N^2/2 bytes).  The first version did both parts in
// one 256-thread workgroup per 64 rows -- 512 waves on the whole chip, 0.65 TB/s (0.83 ms at N = 8192).
template <typename T>
__global__ void __launch_bounds__(1024)
symv_cols_kernel(const T* __restrict__ W, int n, int ldw, const T* __restrict__ x, T* __restrict__ y) {
    __shared__ double part[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.x * 64;
    const int i = i0 + lane;
    double s = 0.0;
    if (i < n) {
        double p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = 0.0;
        int j = i0 + 64 + wv;
        for (; j + 16 * 7 < n; j += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] += (double)W[(size_t)(j + 16 * u) * ldw + i] * (double)x[j + 16 * u];
        }
        for (; j < n; j += 16) p[0] += (double)W[(size_t)j * ldw + i] * (double)x[j];
        // rows of the diagonal block below the diagonal: j in (i, i0 + 64)
        for (int jj = i + 1 + wv; jj < min(i0 + 64, n); jj += 16) p[1] += (double)W[(size_t)jj * ldw + i] * (double)x[jj];
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
    part[wv][lane] = s;
    __syncthreads();
    if (tid < 64 && i0 + tid < n) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][tid];
        y[i0 + tid] = (T)((double)y[i0 + tid] + t);
    }
}

// trmv_lower: y = L x + y0 (y0 may be null), one wave per row.
template <typename T>
__global__ void __launch_bounds__(256)
trmv_lower_kernel(const T* __restrict__ W, int n, int ldw, const T* __restrict__ x, const T* __restrict__ y0,
                  T* __restrict__ y) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= n) return;
    double s = 0.0;
    for (int j = lane; j <= i; j += 64) s += (double)W[(size_t)i * ldw + j] * (double)x[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[i] = (T)(s + (y0 ? (double)y0[i] : 0.0));
}

// trmv_upper: y = y0 + alpha * (U x) for an upper-trapezoidal U stored by rows (row i holds columns >= i; whatever
// lies left of the diagonal is not read), one wave per row, fixed summation order.
template <typename T>
__global__ void __launch_bounds__(256)
trmv_upper_kernel(const T* __restrict__ W, int n, int ldw, const T* __restrict__ x, const T* __restrict__ y0,
                  double alpha, T* __restrict__ y) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= n) return;
    double s = 0.0;
    const int j0 = i & ~63;                                  // aligned start: 512-byte segments for f64
    for (int j = j0 + lane; j < n; j += 64)
        if (j >= i) s += (double)W[(size_t)i * ldw + j] * (double)x[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[i] = (T)((y0 ? (double)y0[i] : 0.0) + alpha * s);
}

template <typename T>
int trmv_upper(const T* W, int n, int ldw, const T* x, const T* y0, double alpha, T* y, hipStream_t s) {
    GPMC_ARG(W && n > 0 && ldw >= n, 1);
    GPMC_ARG(x && y && x != y, 4);
    hipLaunchKernelGGL((trmv_upper_kernel<T>), dim3((n + 3) / 4), dim3(256), 0, s, W, n, ldw, x, y0, alpha, y);
    GPMC_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int symv_lower(const T* W, int n, int ldw, const T* x, T* y, hipStream_t s) {
    GPMC_ARG(W && n > 0 && ldw >= n, 1);
    GPMC_ARG(x && y && x != y, 4);
    hipLaunchKernelGGL((trmv_lower_kernel<T>), dim3((n + 3) / 4), dim3(256), 0, s, W, n, ldw, x, (const T*)nullptr, y);
    GPMC_LAUNCH_CHECK();
    hipLaunchKernelGGL((symv_cols_kernel<T>), dim3((n + 63) / 64), dim3(1024), 0, s, W, n, ldw, x, y);
    GPMC_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int trmv_lower(const T* W, int n, int ldw, const T* x, const T* y0, T* y, hipStream_t s) {
    GPMC_ARG(W && n > 0 && ldw >= n, 1);
    GPMC_ARG(x && y && x != y, 4);
    hipLaunchKernelGGL((trmv_lower_kernel<T>), dim3((n + 3) / 4), dim3(256), 0, s, W, n, ldw, x, y0, y);
    GPMC_LAUNCH_CHECK();
    return 0;
}

#define INST(T)                                                                                              \
    template int logdet_quad<T>(const T*, int, int, const T*, const int*, double*, hipStream_t, double*, const T*, unsigned*, int);             \
    template int pack_lower<T>(const T*, int, int, T, T*, int, int, hipStream_t);                            \
    template int pack_rows<T>(const T*, int, int, int, T*, int, int, int, hipStream_t);                      \
    template int unpack_lower<T>(const T*, int, int, T*, int, int, hipStream_t);                             \
    template int unpack_rows<T>(const T*, int, int, T*, int, int, int, hipStream_t);                         \
    template int symv_lower<T>(const T*, int, int, const T*, T*, hipStream_t);                               \
    template int trmv_lower<T>(const T*, int, int, const T*, const T*, T*, hipStream_t);                     \
    template int trmv_upper<T>(const T*, int, int, const T*, const T*, double, T*, hipStream_t);
INST(double)
INST(float)

}  // namespace gpmc
