// kbuild.hip -- covariance-matrix build (SURVEY.md 8(a) a1 + a2).
//
// Replaces kcGP.covK.RBF(log_ell, log_sf).getCovMatrix(x, z, mode) -- call sites
// kcMCMC/sliceSample.py:104-105,136-137 -- and the dense diagonal S of aux_var_model
// (sliceSample.py:183-190), which is fused onto the diagonal here instead of being materialised.
// Arithmetic follows the pyGPs form the oracle restates: inputs divided by ell, squared distance by
// direct differences, sf2 * exp(-A/2).  HBM-write bound: s*N^2 bytes out, s*N*d in.
//
// Geometry: one 256-thread workgroup per 64 x 128 output tile.  Wave w owns rows w, w+4, ... of the
// tile; lane tx owns two adjacent columns, so every store instruction of a wave writes one
// contiguous 1 KiB (f64) row segment.  Scaled inputs are staged in LDS (row values broadcast, column
// values read as one 16-byte access per lane).
#include "gpmc_common.h"

namespace gpmc {

constexpr int KB_ROWS = 64, KB_COLS = 128, KB_DC = 8, KB_RPT = KB_ROWS / 4;

template <typename T>
__device__ inline T dev_exp(T x);
template <>
__device__ inline double dev_exp<double>(double x) { return exp(x); }
template <>
__device__ inline float dev_exp<float>(float x) { return expf(x); }
template <typename T>
__device__ inline T dev_sqrt(T x);
template <>
__device__ inline double dev_sqrt<double>(double x) { return sqrt(x); }
template <>
__device__ inline float dev_sqrt<float>(float x) { return sqrtf(x); }

template <typename T, int KERNEL>
__global__ void __launch_bounds__(256)
kbuild_kernel(const T* __restrict__ X, int n, int d, int ldx, const T* __restrict__ Z, int nz, int ldz,
              const T* __restrict__ hyp_in, T* __restrict__ K, int rows_out, int cols_out, int ldk, int flags,
              int same_xz, int vec_ok, KbuildExtras<T> ex) {
    const T* hyp = hyp_in;
    const int row0 = blockIdx.y * KB_ROWS;
    const int col0 = (blockIdx.x + ex.ctile0) * KB_COLS;
    // the log-marginal plan's per-evaluation housekeeping rides on workgroup (0, 0) instead of taking launches of its
    // own (a copy and a fill kernel of ~5 us each, 2 % of an N = 2048 evaluation): the right-hand side goes into its
    // riding row below the matrix, the factorisation's info word is cleared
    if (ex.vec_dst != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
        for (int e = threadIdx.x; e < ex.vec_n; e += 256) ex.vec_dst[e] = ex.vec_src[e];
        if (threadIdx.x == 0 && ex.zero_word != nullptr) *ex.zero_word = 0;
    }
    if (ex.zero_quads != 0 && blockIdx.x == 0) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (unsigned q = blockIdx.y * 256u + threadIdx.x; q < ex.zero_quads; q += gridDim.y * 256u) ex.zero_ptr[q] = z;
    }
    if ((flags & GPMC_KB_LOWER) && col0 > row0 + KB_ROWS - 1) return;  // tile strictly above the diagonal

    __shared__ T xi[KB_ROWS][KB_DC];
    __shared__ __attribute__((aligned(16))) T xj[KB_DC][KB_COLS];

    const int tid = threadIdx.x;
    const int tx = tid & 63;
    const int w = tid >> 6;
    const bool ard = (KERNEL == GPMC_KERNEL_MATERN52_ARD);
    // the parameters once into LDS: with a parameter ring (ex.seq) they sit in host memory, and every read of hyp[] in the
    // staging loops below would be a PCIe round trip
    __shared__ T hs[2 + 64];
    const int nh = 2 + (ard ? d : 1);
    const bool hyp_staged = nh <= 2 + 64;
    if (ex.seq != nullptr) hyp += (size_t)(*ex.seq & 63u) * ex.ring_stride;
    if (hyp_staged) {
        if (tid < nh) hs[tid] = hyp[tid];
        __syncthreads();
    }
    auto H = [&](int k) -> T { return hyp_staged ? hs[k] : hyp[k]; };
    const T sf2 = H(0);
    const T dadd = H(1);

    T acc0[KB_RPT], acc1[KB_RPT];
#pragma unroll
    for (int i = 0; i < KB_RPT; ++i) { acc0[i] = 0; acc1[i] = 0; }

    for (int k0 = 0; k0 < d; k0 += KB_DC) {
        const int kc = min(KB_DC, d - k0);
        __syncthreads();
        // stage scaled rows (x_i / ell) and columns (z_j / ell)
        for (int e = tid; e < KB_ROWS * KB_DC; e += 256) {
            const int r = e / KB_DC, kk = e % KB_DC;
            T v = 0;
            if (kk < kc && row0 + r < n) v = X[(size_t)(row0 + r) * ldx + k0 + kk] / H(2 + (ard ? k0 + kk : 0));
            xi[r][kk] = v;
        }
        for (int e = tid; e < KB_COLS * KB_DC; e += 256) {
            const int c = e / KB_DC, kk = e % KB_DC;
            T v = 0;
            if (kk < kc && col0 + c < nz) v = Z[(size_t)(col0 + c) * ldz + k0 + kk] / H(2 + (ard ? k0 + kk : 0));
            xj[kk][c] = v;
        }
        __syncthreads();
        for (int kk = 0; kk < kc; ++kk) {
            const T b0 = xj[kk][2 * tx], b1 = xj[kk][2 * tx + 1];
#pragma unroll
            for (int i = 0; i < KB_RPT; ++i) {
                const T a = xi[w + 4 * i][kk];
                const T t0 = a - b0, t1 = a - b1;
                acc0[i] += t0 * t0;
                acc1[i] += t1 * t1;
            }
        }
    }

    const int j0 = col0 + 2 * tx;
#pragma unroll
    for (int i = 0; i < KB_RPT; ++i) {
        const int r = row0 + w + 4 * i;
        if (r >= rows_out) continue;
        T v0, v1;
        if (KERNEL == GPMC_KERNEL_RBF) {
            v0 = sf2 * dev_exp<T>((T)-0.5 * acc0[i]);
            v1 = sf2 * dev_exp<T>((T)-0.5 * acc1[i]);
        } else {  // ARD Matern-5/2: sf2 (1 + t + 5 r^2/3) exp(-t), t = sqrt5 * r
            const T s5 = dev_sqrt<T>((T)5);
            const T t0 = s5 * dev_sqrt<T>(acc0[i]), t1 = s5 * dev_sqrt<T>(acc1[i]);
            v0 = sf2 * ((T)1 + t0 + ((T)5 / (T)3) * acc0[i]) * dev_exp<T>(-t0);
            v1 = sf2 * ((T)1 + t1 + ((T)5 / (T)3) * acc1[i]) * dev_exp<T>(-t1);
        }
        if (same_xz) {
            if (r == j0) v0 += dadd;
            if (r == j0 + 1) v1 += dadd;
        }
        if (r >= n || j0 >= nz) v0 = ((flags & GPMC_KB_PAD_IDENTITY) && r == j0) ? (T)1 : (T)0;
        if (r >= n || j0 + 1 >= nz) v1 = ((flags & GPMC_KB_PAD_IDENTITY) && r == j0 + 1) ? (T)1 : (T)0;
        T* dst = K + (size_t)r * ldk + j0;
        if (vec_ok && j0 + 1 < cols_out) {
            if (sizeof(T) == 8) {
                d2_t o; o[0] = v0; o[1] = v1;
                *reinterpret_cast<d2_t*>(dst) = o;
            } else {
                typedef float f2_t __attribute__((ext_vector_type(2)));
                f2_t o; o[0] = v0; o[1] = v1;
                *reinterpret_cast<f2_t*>(dst) = o;
            }
        } else {
            if (j0 < cols_out) dst[0] = v0;
            if (j0 + 1 < cols_out) dst[1] = v1;
        }
    }
}

template <typename T>
int kbuild(int kernel, const T* X, int n, int d, int ldx, const T* Z, int nz, int ldz, const T* hyp_dev, T* K,
           int rows_out, int cols_out, int ldk, int flags, hipStream_t s, KbuildExtras<T> ex) {
    GPMC_ARG(kernel == GPMC_KERNEL_RBF || kernel == GPMC_KERNEL_MATERN52_ARD, 1);
    GPMC_ARG(X != nullptr, 2);
    GPMC_ARG(n > 0 && d > 0 && ldx >= d, 3);
    const int same = (Z == nullptr || Z == X) ? 1 : 0;
    if (same) { Z = X; nz = n; ldz = ldx; }
    GPMC_ARG(nz > 0 && ldz >= d, 7);
    GPMC_ARG(hyp_dev != nullptr && K != nullptr, 10);
    GPMC_ARG(rows_out >= n && cols_out >= nz && ldk >= cols_out, 11);
    const int vec_ok = (ldk % 2 == 0) && ((reinterpret_cast<uintptr_t>(K) % (2 * sizeof(T))) == 0);
    const int ctiles = (cols_out + KB_COLS - 1) / KB_COLS;
    GPMC_ARG(ex.ctile0 >= 0 && ex.ctile1 >= 0 && ex.ctile1 <= ctiles && (ex.ctile1 == 0 || ex.ctile0 < ex.ctile1), 16);
    if (ex.ctile1 == 0) { ex.ctile0 = 0; ex.ctile1 = ctiles; }
    dim3 grid(ex.ctile1 - ex.ctile0, (rows_out + KB_ROWS - 1) / KB_ROWS);
    if (kernel == GPMC_KERNEL_RBF)
        hipLaunchKernelGGL((kbuild_kernel<T, GPMC_KERNEL_RBF>), grid, dim3(256), 0, s, X, n, d, ldx, Z, nz, ldz,
                           hyp_dev, K, rows_out, cols_out, ldk, flags, same, vec_ok, ex);
    else
        hipLaunchKernelGGL((kbuild_kernel<T, GPMC_KERNEL_MATERN52_ARD>), grid, dim3(256), 0, s, X, n, d, ldx, Z,
                           nz, ldz, hyp_dev, K, rows_out, cols_out, ldk, flags, same, vec_ok, ex);
    GPMC_LAUNCH_CHECK();
    return 0;
}

template int kbuild<double>(int, const double*, int, int, int, const double*, int, int, const double*, double*, int,
                            int, int, int, hipStream_t, KbuildExtras<double>);
template int kbuild<float>(int, const float*, int, int, int, const float*, int, int, const float*, float*, int, int,
                           int, int, hipStream_t, KbuildExtras<float>);

}  // namespace gpmc
