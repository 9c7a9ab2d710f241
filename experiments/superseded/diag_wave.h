// diag_wave.h -- 64x64 Cholesky inside ONE wavefront, registers only (no LDS, no barriers).
//
// The symmetric block is held as ten 16x16 tiles of its UPPER triangle in the MFMA accumulator
// layout (lane (q,r): register e = element [crow(q,e)][r]); the factor is U = L^T.  Per tile row:
//   * 16 column steps on the diagonal tile and the tiles to its right: the pivot row is fetched
//     across lane-quarters with ds_bpermute, the pivot column within the quarter likewise; the
//     rank-1 update is 4 FMAs per tile; finished rows are scaled by 1/sqrt(d) in place;
//   * the tiles below are updated with MFMA: T[a'][b'] -= U[a][a']^T U[a][b'] -- the accumulator
//     layout of U[a][a'] IS the A-operand layout of its transpose and that of U[a][b'] the
//     B-operand layout, so no data moves between the VALU phase and the MFMA phase.
// inv(U_aa) (needed by the panel TRSM) is computed per lane-column by back substitution with the
// U entries broadcast through v_readlane.
#pragma once
#include "gpmc_common.h"

namespace gpmc {

template <typename T>
struct Xl;
template <>
struct Xl<double> {
    __device__ static inline double bperm(double v, int src_lane) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo);
        hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
        return __hiloint2double(hi, lo);
    }
    __device__ static inline double rdlane(double v, int lane) {  // lane must be wave-uniform
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
        return __hiloint2double(hi, lo);
    }
};
template <>
struct Xl<float> {
    __device__ static inline float bperm(float v, int src_lane) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
    }
    __device__ static inline float rdlane(float v, int lane) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
    }
};

// One column step of tile row `Tr[0..NT)` (Tr[0] is the diagonal tile).  EC is the register that holds
// the pivot row, qc the lane-quarter that holds it (wave-uniform, may be a runtime value).
template <typename T, int NT, int EC>
__device__ inline void wave_col_step(typename Mfma<T>::acc_t (&Tr)[NT], int qc, int q, int r, int gcol, int& fail) {
    using M = Mfma<T>;
    const int c = M::crow(qc, EC);
    const T d = Xl<T>::rdlane(Tr[0][EC], 16 * qc + c);
    if (!(d > (T)0) && fail == 0) fail = gcol + c + 1;
    const T inv = (T)1 / d;
    const T rs = (T)1 / sqrt(d);
    T uj[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) uj[b] = Xl<T>::bperm(Tr[b][EC], 16 * qc + r);   // pivot row, column r
    T w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const T ui = Xl<T>::bperm(Tr[0][e], 16 * q + c);                          // pivot column, row crow(q,e)
        w[e] = (M::crow(q, e) > c) ? ui * inv : (T)0;                             // only rows below the pivot
    }
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) Tr[b][e] -= w[e] * uj[b];
    const bool mine = (q == qc);
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        T val = Tr[b][EC] * rs;
        if (b == 0 && r < c) val = (T)0;
        Tr[b][EC] = mine ? val : Tr[b][EC];
    }
}

template <typename T, int NT>
__device__ inline void wave_factor_tile_row(typename Mfma<T>::acc_t (&Tr)[NT], int q, int r, int gcol, int& fail) {
    // visit pivots in increasing column order: c = crow(qc, ec)
    if constexpr (sizeof(T) == 8) {   // c = qc + 4 ec : ec outer (unrolled), qc inner
        for (int qc = 0; qc < 4; ++qc) wave_col_step<T, NT, 0>(Tr, qc, q, r, gcol, fail);
        for (int qc = 0; qc < 4; ++qc) wave_col_step<T, NT, 1>(Tr, qc, q, r, gcol, fail);
        for (int qc = 0; qc < 4; ++qc) wave_col_step<T, NT, 2>(Tr, qc, q, r, gcol, fail);
        for (int qc = 0; qc < 4; ++qc) wave_col_step<T, NT, 3>(Tr, qc, q, r, gcol, fail);
    } else {                          // c = 4 qc + ec : qc outer, ec inner (unrolled)
        for (int qc = 0; qc < 4; ++qc) {
            wave_col_step<T, NT, 0>(Tr, qc, q, r, gcol, fail);
            wave_col_step<T, NT, 1>(Tr, qc, q, r, gcol, fail);
            wave_col_step<T, NT, 2>(Tr, qc, q, r, gcol, fail);
            wave_col_step<T, NT, 3>(Tr, qc, q, r, gcol, fail);
        }
    }
}

// V = U^-1 for an upper-triangular 16x16 tile in accumulator layout; every lane-quarter computes
// column r redundantly, then keeps the rows its registers stand for.
template <typename T>
__device__ inline typename Mfma<T>::acc_t wave_inv_upper(const typename Mfma<T>::acc_t& U, int q, int r) {
    using M = Mfma<T>;
    typename M::acc_t dg;   // reciprocal of the diagonal entry this lane holds (if any)
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[e] = (T)1 / ((M::crow(q, e) == r) ? U[e] : (T)1);
    T v[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        T s = (i == r) ? (T)1 : (T)0;
#pragma unroll
        for (int k = i + 1; k < 16; ++k) s -= Xl<T>::rdlane(U[M::re(i)], 16 * M::rq(i) + k) * v[k];
        v[i] = s * Xl<T>::rdlane(dg[M::re(i)], 16 * M::rq(i) + i);
    }
    typename M::acc_t V;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        T x = v[M::crow(0, e)];
#pragma unroll
        for (int qq = 1; qq < 4; ++qq) x = (q == qq) ? v[M::crow(qq, e)] : x;
        V[e] = x;
    }
    return V;
}

// Full 64x64: Tm[a][b] (b >= a) in, U tiles out in place, V[a] = inv(U_aa).
template <typename T>
__device__ inline void wave_factor64(typename Mfma<T>::acc_t (&Tm)[4][4], typename Mfma<T>::acc_t (&V)[4], int q,
                                     int r, int gcol, int& fail) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    {
        acc_t row[4] = {Tm[0][0], Tm[0][1], Tm[0][2], Tm[0][3]};
        wave_factor_tile_row<T, 4>(row, q, r, gcol, fail);
        Tm[0][0] = row[0]; Tm[0][1] = row[1]; Tm[0][2] = row[2]; Tm[0][3] = row[3];
    }
#pragma unroll
    for (int a2 = 1; a2 < 4; ++a2)
#pragma unroll
        for (int b2 = a2; b2 < 4; ++b2)
#pragma unroll
            for (int s = 0; s < 4; ++s) Tm[a2][b2] = M::mma(-Tm[0][a2][s], Tm[0][b2][s], Tm[a2][b2]);
    V[0] = wave_inv_upper<T>(Tm[0][0], q, r);
    {
        acc_t row[3] = {Tm[1][1], Tm[1][2], Tm[1][3]};
        wave_factor_tile_row<T, 3>(row, q, r, gcol + 16, fail);
        Tm[1][1] = row[0]; Tm[1][2] = row[1]; Tm[1][3] = row[2];
    }
#pragma unroll
    for (int a2 = 2; a2 < 4; ++a2)
#pragma unroll
        for (int b2 = a2; b2 < 4; ++b2)
#pragma unroll
            for (int s = 0; s < 4; ++s) Tm[a2][b2] = M::mma(-Tm[1][a2][s], Tm[1][b2][s], Tm[a2][b2]);
    V[1] = wave_inv_upper<T>(Tm[1][1], q, r);
    {
        acc_t row[2] = {Tm[2][2], Tm[2][3]};
        wave_factor_tile_row<T, 2>(row, q, r, gcol + 32, fail);
        Tm[2][2] = row[0]; Tm[2][3] = row[1];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) Tm[3][3] = M::mma(-Tm[2][3][s], Tm[2][3][s], Tm[3][3]);
    V[2] = wave_inv_upper<T>(Tm[2][2], q, r);
    {
        acc_t row[1] = {Tm[3][3]};
        wave_factor_tile_row<T, 1>(row, q, r, gcol + 48, fail);
        Tm[3][3] = row[0];
    }
    V[3] = wave_inv_upper<T>(Tm[3][3], q, r);
}

// Standalone launch: one wave factors the block at Wd (lower triangle read), writes L = U^T (strict
// upper of the block zeroed) and dinv[t][r][k] = inv(L_tt)[r][k] = V_tt[k][r].
template <typename T>
__global__ void __launch_bounds__(64)
diag_factor_wave_kernel(T* __restrict__ Wd, int ld, T* __restrict__ dinv, int* __restrict__ info, int c0) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
    acc_t Tm[4][4], V[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int row = 16 * b + r, col = 16 * a + M::crow(q, e);   // U_ab[crow][r] = A[16b+r][16a+crow]
                if (col > row) { const int t = row; row = col; col = t; }   // diagonal tiles: mirror
                Tm[a][b][e] = Wd[(size_t)row * ld + col];
            }
    int fail = 0;
    wave_factor64<T>(Tm, V, q, r, c0, fail);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Wd[(size_t)(16 * b + r) * ld + 16 * a + M::crow(q, e)] = Tm[a][b][e];
                if (b > a) Wd[(size_t)(16 * a + r) * ld + 16 * b + M::crow(q, e)] = (T)0;
            }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) dinv[t * 256 + r * 16 + M::crow(q, e)] = V[t][e];
    if (lane == 0 && fail != 0) atomicCAS(info, 0, fail);
}

}  // namespace gpmc
