// panel_pair.h -- the FIRST 128 columns of an outer panel in one launch (f64 arithmetic, TS storage).
//
// Written to save the ~4.9 us the 64-column kernel (panel_fused.h) spends in every second block re-loading, in a new
// launch, what the same workgroup held a moment earlier.  Measured: alone on the chip it takes exactly as long as the
// two launches it replaces (43-45 vs 18.6 + 23.7 us at N = 2048: the redundant L21 solve and the extra LDS round trips
// cost what the re-load saved), but beside the trailing GEMM every panel LAUNCH waits ~40 us for CU slots, and one
// launch per 128 columns halves those waits: 6144: 3.96 -> 3.83 ms, 8192: 6.69 -> 6.50 ms.  So potrf() uses it under
// look-ahead only (gpmc_set_option("panel_pair")).  Workgroup b (strip rows c0 + 64 b ...) does, with no left-looking
// history (k0 == c0):
//     L11 = chol(D1)                       redundantly, as in panel_fused.h
//     L21 = S21 L11^-T                     redundantly (S21 = rows of the second diagonal block x columns of the first)
//     X1  = T1  L11^-T                     its own strip, first 64 columns                       (b >= 2)
//     D2 -= L21 L21^T ;  T2 -= X1 L21^T    rank-64 updates straight from LDS
//     L22 = chol(D2)                       redundantly
//     X2  = T2  L22^-T                     its own strip, second 64 columns                      (b >= 2)
// Workgroup 0 owns L11, workgroup 1 owns L21 and L22.  All three blocks are inputs of every other workgroup of the
// launch, so none may be overwritten in place: they are parked in scratch and the NEXT panel launch on the stream (or
// panel_flush_kernel at the end) copies them into W -- the protocol of panel_fused.h, extended to three blocks.
#pragma once
#include "panel_fused.h"

namespace gpmc {

// LDS: four 64 x 64 buffers, 152 KB -- a workgroup of this kernel needs a completely free CU.  A two-buffer variant
// (X1 kept in registers as the update's B operand, T2 loaded from global in register form; 79 KB, 212 VGPRs, able to
// slot in beside one GEMM workgroup like the 64-column kernel) was correct but slower in itself (N = 2048, alone:
// 0.968 vs 0.922 ms) and gave the look-ahead gain back (N = 8192: 6.69 vs 6.46 ms, 4096: 2.19 vs 2.09).  Whether a
// panel workgroup shares a CU or waits for an empty one is not what decides: padding the 64-column kernel's LDS
// request so that it, too, needs an empty CU (panel_lds_pad) loses 2-3 % (8192: 6.58 vs 6.41 ms).
struct PairShared {
    double A[64][MP_LD];      // Lout of factor 1, later of factor 2
    double S[64][MP_LD];      // S21 in;  L21 as [row][k] after its solve (in place)
    double B[64][MP_LD];      // T1 in;   X1 as [row][k] after its solve (in place)
    double C[64][MP_LD];      // T2 in;   X2 on its way out
    union {
        double Praw[64][MP_PLD];
        double Linv[16][4][4];
    };
    double rsd[64];
    int fail;
};
static_assert(sizeof(PairShared) <= 160 * 1024, "PairShared must fit the 160 KB of a CU");

template <typename TS>
__global__ void __launch_bounds__(256)
panel_pair_kernel(TS* __restrict__ W, int ld, int c0, int* __restrict__ info, double* __restrict__ lpark,
                  double* __restrict__ lpark21, int prev_c0, int prev2_c0, int prev21_c0, int hi_prio) {
    if (hi_prio) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char pair_smem[];
    PairShared& sh = *reinterpret_cast<PairShared*>(pair_smem);
    double (*Lout)[MP_LD] = sh.A;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int i = lane >> 4, c = lane & 15, ia = lane & 3;
    const int b = blockIdx.x;
    const size_t R0 = (size_t)c0 + 64 * (size_t)b;
    const size_t c1 = (size_t)c0 + 64;

    panel_flush_duties<TS>(W, ld, lpark, lpark21, prev_c0, prev2_c0, prev21_c0, tid);

    // ---------------- 0. inputs.  The strip blocks stay in LDS until their own solve: held in registers across the
    // first factor they pushed the kernel to 440 registers, with accumulators shuttled through AGPRs around every MFMA
    double D1[4][4], D2[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const size_t r = 4 * (w + 4 * m) + i;
            D1[m][h] = (h <= m) ? (double)W[((size_t)c0 + r) * ld + c0 + 16 * h + c] : 0.0;
            D2[m][h] = (h <= m && b != 0) ? (double)W[(c1 + r) * ld + c1 + 16 * h + c] : 0.0;
        }
    if (b != 0) {
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            *reinterpret_cast<d2_t*>(&sh.S[row][c2]) = ld2<TS>(&W[(c1 + row) * ld + c0 + c2]);
            if (b >= 2) {
                *reinterpret_cast<d2_t*>(&sh.B[row][c2]) = ld2<TS>(&W[(R0 + row) * ld + c0 + c2]);
                *reinterpret_cast<d2_t*>(&sh.C[row][c2]) = ld2<TS>(&W[(R0 + row) * ld + c1 + c2]);
            }
        }
    }

    auto make_inverses = [&]() {   // inverses of the sixteen 4x4 diagonal blocks of Lout (rsd = 1 / diag from the factor)
        if (tid < 16) {
            const int o = 4 * tid;
            const double l10 = Lout[o + 1][o];
            const double l20 = Lout[o + 2][o], l21 = Lout[o + 2][o + 1];
            const double l30 = Lout[o + 3][o], l31 = Lout[o + 3][o + 1], l32 = Lout[o + 3][o + 2];
            const double i00 = sh.rsd[o], i11 = sh.rsd[o + 1], i22 = sh.rsd[o + 2], i33 = sh.rsd[o + 3];
            const double i10 = -(l10 * i00) * i11;
            const double i21 = -(l21 * i11) * i22;
            const double i32 = -(l32 * i22) * i33;
            const double i20 = -(l20 * i00 + l21 * i10) * i22;
            const double i31 = -(l31 * i11 + l32 * i21) * i33;
            const double i30 = -(l30 * i00 + l31 * i10 + l32 * i20) * i33;
            double(*V)[4] = sh.Linv[tid];
            V[0][0] = i00; V[0][1] = 0.0; V[0][2] = 0.0; V[0][3] = 0.0;
            V[1][0] = i10; V[1][1] = i11; V[1][2] = 0.0; V[1][3] = 0.0;
            V[2][0] = i20; V[2][1] = i21; V[2][2] = i22; V[2][3] = 0.0;
            V[3][0] = i30; V[3][1] = i31; V[3][2] = i32; V[3][3] = i33;
        }
        __syncthreads();
    };
    // X <- X Lout^-T for a 64 x 64 block held in LDS as [row][col]: wave w owns rows 16w .. 16w+15; it reads them
    // transposed (slab rows = block columns, lanes = its rows), solves by 4-row slabs with the 4x4 inverses (an
    // accumulator slab is the next MFMA's B operand unchanged) and writes the SAME locations back -- in place
    auto solve_block = [&](double (*X)[MP_LD]) {
        double T[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) T[g] = X[16 * w + c][4 * g + i];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const double av = sh.Linv[p][ia][i];
            const double Y = __builtin_amdgcn_mfma_f64_4x4x4f64(av, T[p], 0.0, 0, 0, 0);
            T[p] = Y;
            const double Yn = -Y;     // one sign flip per step instead of one per multiplier
#pragma unroll
            for (int g = p + 1; g < 16; ++g) {
                const double al = Lout[4 * g + ia][4 * p + i];
                T[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(al, Yn, T[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) X[16 * w + c][4 * g + i] = T[g];
    };
    auto park = [&](double* dst, double (*X)[MP_LD]) {   // a 64 x 64 LDS block -> scratch
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            *reinterpret_cast<d2_t*>(&dst[row * 64 + c2]) = *reinterpret_cast<const d2_t*>(&X[row][c2]);
        }
    };

    // ---------------- 1. L11 (every workgroup)
    mp_factor64(D1, Lout, sh.Praw, &sh.fail, sh.rsd, tid, c0);     // opens and closes with a barrier
    const int f1 = sh.fail;
    if (b == 0) {
        park(lpark + (size_t)(c0 / 64) * 4096, Lout);
        if (tid == 0 && f1 != 0) atomicCAS(info, 0, f1);
        return;
    }
    make_inverses();

    // ---------------- 2. L21 = S21 L11^-T (every workgroup), X1 = T1 L11^-T (its own strip), both in place in LDS
    solve_block(sh.S);
    if (b >= 2) solve_block(sh.B);
    __syncthreads();
    if (b == 1) {   // owner of L21: park it (it is an input of every other workgroup)
        park(lpark21 + (size_t)(c0 / 64) * 4096, sh.S);
    } else {        // X1 is nobody else's input: straight to W, 512-byte row segments
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            st2<TS>(&W[(R0 + row) * ld + c0 + c2], *reinterpret_cast<const d2_t*>(&sh.B[row][c2]));
        }
    }

    // ---------------- 3. rank-64 updates from LDS: D2 -= L21 L21^T ; T2 -= (X1 L21^T)^T, T2 held transposed
    double T2[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) T2[g] = (b >= 2) ? sh.C[16 * w + c][4 * g + i] : 0.0;
    {
        double PD[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int h = 0; h < 4; ++h) PD[m][h] = 0.0;
#pragma unroll 2
        for (int k8 = 0; k8 < 8; ++k8) {
            const int ko = 8 * k8 + 2 * i;
            d2_t bD[4], aD[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) bD[h] = *reinterpret_cast<const d2_t*>(&sh.S[16 * h + c][ko]);
#pragma unroll
            for (int m = 0; m < 4; ++m) aD[m] = *reinterpret_cast<const d2_t*>(&sh.S[4 * (w + 4 * m) + ia][ko]);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int h = 0; h <= m; ++h)
                        PD[m][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(aD[m][e2], bD[h][e2], PD[m][h], 0, 0, 0);
            if (b >= 2) {
                const d2_t bT = -*reinterpret_cast<const d2_t*>(&sh.B[16 * w + c][ko]);   // the update's minus sign, once
#pragma unroll
                for (int gb = 0; gb < 16; gb += 8) {
                    d2_t aT[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) aT[g] = *reinterpret_cast<const d2_t*>(&sh.S[4 * (gb + g) + ia][ko]);
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                        for (int g = 0; g < 8; ++g)
                            T2[gb + g] = __builtin_amdgcn_mfma_f64_4x4x4f64(aT[g][e2], bT[e2], T2[gb + g], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int h = 0; h < 4; ++h) D2[m][h] -= PD[m][h];
    }

    // ---------------- 4. L22 (every remaining workgroup); mp_factor64's opening barrier also ends the reads of
    // Lout / Linv by the solves above
    mp_factor64(D2, Lout, sh.Praw, &sh.fail, sh.rsd, tid, c0 + 64);
    const int f2 = sh.fail;
    if (b == 1) {
        park(lpark + (size_t)(c0 / 64 + 1) * 4096, Lout);
        const int f = f1 ? f1 : f2;                   // the FIRST bad pivot of the pair
        if (tid == 0 && f != 0) atomicCAS(info, 0, f);
        return;
    }
    make_inverses();

    // ---------------- 5. X2 = T2 L22^-T (T2 is already in registers, transposed), out through C
    {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const double av = sh.Linv[p][ia][i];
            const double Y = __builtin_amdgcn_mfma_f64_4x4x4f64(av, T2[p], 0.0, 0, 0, 0);
            T2[p] = Y;
            const double Yn = -Y;
#pragma unroll
            for (int g = p + 1; g < 16; ++g) {
                const double al = Lout[4 * g + ia][4 * p + i];
                T2[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(al, Yn, T2[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) sh.C[16 * w + c][4 * g + i] = T2[g];
    }
    __syncthreads();
    for (int e = tid; e < 64 * 32; e += 256) {
        const int row = e >> 5, c2 = (e & 31) * 2;
        st2<TS>(&W[(R0 + row) * ld + c1 + c2], *reinterpret_cast<const d2_t*>(&sh.C[row][c2]));
    }
}

// end-of-factorisation flush of whatever the last panel launch parked (grid = 3 workgroups)
template <typename TS>
__global__ void __launch_bounds__(256)
panel_flush3_kernel(TS* __restrict__ W, int ld, const double* __restrict__ lpark, const double* __restrict__ lpark21,
                    int prev_c0, int prev2_c0, int prev21_c0) {
    panel_flush_duties<TS>(W, ld, lpark, lpark21, prev_c0, prev2_c0, prev21_c0, threadIdx.x);
}

}  // namespace gpmc
