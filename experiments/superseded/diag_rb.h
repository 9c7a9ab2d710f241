// diag_rb.h -- 64x64 Cholesky by one 256-thread workgroup, matrix held in REGISTERS.
//
// Thread (bi, bj) = (tid >> 4, tid & 15) owns the 4x4 sub-block rows 4bi.., cols 4bj.. .  Column step c:
// the 16 owners of column c publish it (unscaled, 64 values incl. the pivot) into a double-buffered LDS
// line, ONE barrier, every thread reads its 4 row values + 4 column values + the pivot, forms
// w_i = a_ic / d_c and applies 16 FMAs to its registers.  Columns are scaled by 1/sqrt(d_c) at the end
// (dpotf2 also scales by a reciprocal).  ~64 barrier round trips per block instead of v1's LDS
// read-modify-write loops.
#pragma once
#include "gpmc_common.h"

namespace gpmc {

template <typename T>
__device__ inline T fast_rcp(T x) { return (T)1 / x; }
template <>
__device__ inline double fast_rcp<double>(double x) {   // v_rcp_f64 + 2 Newton steps: <= 1 ulp-ish, no div_fixup path
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

struct DiagRbShared {
    // col[2][64]: published column (double-buffered); dpiv[64]: pivots d_c; Ls: factor for write-out/inverse
    double col[2][64];
    double dpiv[64];
    double sdi[64];
    double Ls[64][66];
    int fail;
};

// Factor the block held as a[4][4] per thread.  On return sh.Ls holds L (row-major, strict upper zero),
// sh.sdi[j] = 1/L[j][j].  All 256 threads must call.  gcol: global column of the block (for info).
template <typename T>
__device__ inline void diag_rb_factor(T (&a)[4][4], DiagRbShared& sh, int tid, int gcol) {
    const int bi = tid >> 4, bj = tid & 15;
    if (tid == 0) sh.fail = 0;
#pragma unroll 1
    for (int tq = 0; tq < 16; ++tq) {
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
            const int c = 4 * tq + tc;
            double* cb = sh.col[c & 1];
            if (bj == tq && bi >= tq) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) cb[4 * bi + ii] = (double)a[ii][tc];
            }
            __syncthreads();
            if (bi >= tq && bj >= tq && bj <= bi) {
                const double d = cb[c];
                const double inv = fast_rcp<double>(d);
                double w[4], lj[4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) w[ii] = (4 * bi + ii > c) ? cb[4 * bi + ii] * inv : 0.0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) lj[jj] = (4 * bj + jj > c) ? cb[4 * bj + jj] : 0.0;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) a[ii][jj] = (T)((double)a[ii][jj] - w[ii] * lj[jj]);
                if (bi == tq && bj == tq) {   // the pivot's owner records it
                    sh.dpiv[c] = d;
                    if (!(d > 0.0) && sh.fail == 0) sh.fail = gcol + c + 1;
                }
            }
        }
    }
    __syncthreads();
    if (tid < 64) sh.sdi[tid] = 1.0 / sqrt(sh.dpiv[tid]);
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int i = 4 * bi + ii, j = 4 * bj + jj;
            double v = 0.0;
            if (j < i) v = (double)a[ii][jj] * sh.sdi[j];
            else if (j == i) v = sqrt(sh.dpiv[j]);
            sh.Ls[i][j] = v;
        }
    __syncthreads();
}

template <typename T>
__global__ void __launch_bounds__(256)
diag_factor_rb_kernel(T* __restrict__ Wd, int ld, T* __restrict__ dinv, int* __restrict__ info, int c0) {
    __shared__ DiagRbShared sh;
    const int tid = threadIdx.x, bi = tid >> 4, bj = tid & 15;
    T a[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            int i = 4 * bi + ii, j = 4 * bj + jj;
            if (j > i) { const int t = i; i = j; j = t; }
            a[ii][jj] = Wd[(size_t)i * ld + j];
        }
    diag_rb_factor<T>(a, sh, tid, c0);
    for (int e = tid; e < 64 * 64; e += 256) {
        const int i = e >> 6, j = e & 63;
        Wd[(size_t)i * ld + j] = (T)sh.Ls[i][j];
    }
    if (tid < 64) {  // inverse of diagonal tile t, column jj, by forward substitution
        const int t = tid >> 4, jj = tid & 15, o = 16 * t;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double s = (i == jj) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) s -= sh.Ls[o + i][o + k] * x[k];
            x[i] = s * sh.sdi[o + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) dinv[t * 256 + i * 16 + jj] = (T)x[i];
    }
    if (tid == 0 && sh.fail != 0) atomicCAS(info, 0, sh.fail);
}

}  // namespace gpmc
