// panel_fused.h -- one kernel per 64-column block of the outer panel (f64): left-looking update,
// diagonal factor and TRSM fused, everything on v_mfma_f64_4x4x4_4b_f64.
//
// Grid: one 256-thread workgroup per 64-row strip from the diagonal block down (strip 0 = the diagonal
// block itself).  Every workgroup
//   1. accumulates  P_D = L_d L_d^T  (lower slabs of the diagonal block, K = c0-k0 columns of this outer
//      panel already factored) and  P_T = L_d L_s^T  (its own strip, TRANSPOSED: slab rows = block
//      columns, lanes = strip rows) from 64-column chunks staged in LDS;
//   2. factors D = A_dd - P_D REDUNDANTLY (diag_mp.h) -- recomputing a 64x64 factor in every workgroup
//      costs no wall time and removes two kernel boundaries and the 16x16 inverse tiles;
//   3. solves its strip: Y = T^T-form forward substitution by 4-row slabs with the 4x4 diagonal
//      inverses, Y_p = inv(L_pp) T_p ; T_g -= L_gp Y_p -- an accumulator slab is the next MFMA's B
//      operand unchanged;
//   4. writes the strip back through an LDS transpose.  Strip 0 must NOT overwrite the diagonal block
//      in place -- slower workgroups of the same launch still read it as input -- so it parks L in
//      scratch and the NEXT launch on the stream (or panel_flush_kernel at the end) copies it into W.
#pragma once
#include "diag_mp.h"

namespace gpmc {

// storage <-> compute conversion: the panel arithmetic is always f64; TS (double or float) is what W holds
template <typename TS>
__device__ inline d2_t ld2(const TS* p);
template <>
__device__ inline d2_t ld2<double>(const double* p) { return *reinterpret_cast<const d2_t*>(p); }
template <>
__device__ inline d2_t ld2<float>(const float* p) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = *reinterpret_cast<const f2_t*>(p);
    d2_t o; o[0] = (double)v[0]; o[1] = (double)v[1];
    return o;
}
template <typename TS>
__device__ inline void st2(TS* p, d2_t v);
template <>
__device__ inline void st2<double>(double* p, d2_t v) { *reinterpret_cast<d2_t*>(p) = v; }
template <>
__device__ inline void st2<float>(float* p, d2_t v) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t o; o[0] = (float)v[0]; o[1] = (float)v[1];
    *reinterpret_cast<f2_t*>(p) = o;
}

// flush duties shared by both panel kernels: copy what the previous panel launch parked back into W -- up to two
// diagonal factors (duties 0, 1) and one L21 block (duty 2: rows c+64.., columns c..).  Duty k is done by workgroup
// min(k, gridDim.x - 1): the last launches of a workspace without riding rows have fewer than three workgroups.
template <typename TS>
__device__ inline void panel_flush_duties(TS* __restrict__ W, int ld, const double* __restrict__ lpark,
                                          const double* __restrict__ lpark21, int prev_c0, int prev2_c0, int prev21_c0,
                                          int tid) {
    const int b = blockIdx.x, last = (int)gridDim.x - 1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (b != (k < last ? k : last)) continue;
        const int pc = (k == 0) ? prev_c0 : (k == 1 ? prev2_c0 : prev21_c0);
        if (pc < 0) continue;
        const double* src = ((k == 2) ? lpark21 : lpark) + (size_t)(pc / 64) * 4096;
        const size_t r0 = (size_t)pc + (k == 2 ? 64 : 0), q0 = (size_t)pc;
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            st2<TS>(&W[(r0 + row) * ld + q0 + c2], *reinterpret_cast<const d2_t*>(&src[row * 64 + c2]));
        }
    }
}

struct PanelShared {
    // Ld: chunk of the diagonal strip's rows [row][k] (phase 1); the SAME storage then holds the diagonal
    // block's factor Lout (phases 2-3).  78,856 B in all.  LDS budget of the look-ahead: a CU has 163,840 B; the
    // big-tile GEMM asks for 73,728 B + a pad that lifts it above 81,920 B (so two of them never share a CU) yet
    // leaves room for one of these (potrf.hip, g_lds_pad_now) -- with 1,280-byte allocation granules 62 + 65 <= 128.
    // Before Linv was folded onto Praw the struct was 80,904 B (64 granules) and the two kernels never shared a CU:
    // every panel launch waited for GEMM workgroups to retire.
    double Ld[64][MP_LD];
    double Ls[64][MP_LD];     // chunk of this strip's rows; later the strip block in / out
    union {
        double Praw[64][MP_PLD];  // micro-panel staging of the factor (dead after mp_factor64's closing barrier)
        double Linv[16][4][4];    // inverses of the 4x4 diagonal blocks of Lout (written after that barrier)
    };
    double rsd[64];           // 1 / L[j][j] from the micro-panel wave
    int fail;
};
static_assert(sizeof(PanelShared) <= 62 * 1280, "PanelShared must stay within 62 LDS granules (see the budget above)");

// DBG = true is a diagnostic instantiation only (gpmc_debug_panel_stamps): wave 0 of workgroup 1 writes s_memtime
// stamps at the phase boundaries into dbg[]; the production launch uses DBG = false and no stamp executes.
#define PANEL_STAMP(k)                                                                         \
    do {                                                                                       \
        if (DBG && blockIdx.x == 1 && tid == 0) dbg[k] = __builtin_amdgcn_s_memtime();          \
    } while (0)

template <bool DBG, typename TS>
__global__ void __launch_bounds__(256)
panel_block_kernel(TS* __restrict__ W, int ld, int k0, int c0, int* __restrict__ info,
                   double* __restrict__ lpark, int prev_c0, unsigned long long* __restrict__ dbg, int hi_prio,
                   const double* __restrict__ lpark21, int prev2_c0, int prev21_c0) {
    // Beside the trailing GEMM (look-ahead) every SIMD also holds a GEMM wave issuing MFMAs back to back; this
    // kernel is a latency chain, so its waves ask for issue priority.
    if (hi_prio) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char panel_smem[];
    PanelShared& sh = *reinterpret_cast<PanelShared*>(panel_smem);
    double (*Lout)[MP_LD] = sh.Ld;   // aliased: valid once phase 1 is over
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int i = lane >> 4, c = lane & 15, ia = lane & 3;   // i doubles as the operands' k index (lane >> 4)
    const bool is_diag = (blockIdx.x == 0);
    const size_t R0 = (size_t)c0 + 64 * (size_t)blockIdx.x;
    PANEL_STAMP(0);

    // flush what the previous panel launch parked (nobody reads those blocks in this launch): one diagonal factor, or
    // after a 128-column launch (panel_pair.h) two diagonal factors and L21
    panel_flush_duties<TS>(W, ld, lpark, lpark21, prev_c0, prev2_c0, prev21_c0, tid);
    double PD[4][4];
    double PT[16];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 4; ++h) PD[m][h] = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) PT[g] = 0.0;

    // ---------------- 0. issue the loads nothing depends on yet: the diagonal block's slabs and this strip's block
    double Din[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 4; ++h)
            Din[m][h] = (h <= m) ? (double)W[((size_t)c0 + 4 * (w + 4 * m) + i) * ld + c0 + 16 * h + c] : 0.0;

    PANEL_STAMP(1);
    // ---------------- 1. left-looking products over the K = c0 - k0 finished columns of this outer panel;
    // the next 64-column chunk is fetched into registers while the current one is multiplied
    d2_t pd[8], ps[8];
    auto fetch_chunk = [&](int kc) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int e = tid + 256 * t, row = e >> 5, c2 = (e & 31) * 2;
            pd[t] = ld2<TS>(&W[((size_t)c0 + row) * ld + kc + c2]);
            if (!is_diag) ps[t] = ld2<TS>(&W[(R0 + row) * ld + kc + c2]);
        }
    };
    if (k0 < c0) fetch_chunk(k0);
    for (int kc = k0; kc < c0; kc += 64) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int e = tid + 256 * t, row = e >> 5, c2 = (e & 31) * 2;
            *reinterpret_cast<d2_t*>(&sh.Ld[row][c2]) = pd[t];
            if (!is_diag) *reinterpret_cast<d2_t*>(&sh.Ls[row][c2]) = ps[t];
        }
        __syncthreads();
        if (kc + 64 < c0) fetch_chunk(kc + 64);
#pragma unroll 2
        for (int k8 = 0; k8 < 8; ++k8) {
            const int ko = 8 * k8 + 2 * i;   // this lane's k pair
            d2_t bD[4], aD[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) bD[h] = *reinterpret_cast<const d2_t*>(&sh.Ld[16 * h + c][ko]);
#pragma unroll
            for (int m = 0; m < 4; ++m) aD[m] = *reinterpret_cast<const d2_t*>(&sh.Ld[4 * (w + 4 * m) + ia][ko]);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int h = 0; h <= m; ++h)
                        PD[m][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(aD[m][e2], bD[h][e2], PD[m][h], 0, 0, 0);
            if (!is_diag) {
                const d2_t bT = *reinterpret_cast<const d2_t*>(&sh.Ls[16 * w + c][ko]);
                // two batches of 8 row-groups: 32 operand registers live instead of 64 (at 256 VGPRs the compiler
                // parks accumulators in AGPRs and shuffles them around every MFMA)
#pragma unroll
                for (int gb = 0; gb < 16; gb += 8) {
                    d2_t aT[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) aT[g] = *reinterpret_cast<const d2_t*>(&sh.Ld[4 * (gb + g) + ia][ko]);
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                        for (int g = 0; g < 8; ++g)
                            PT[gb + g] = __builtin_amdgcn_mfma_f64_4x4x4f64(aT[g][e2], bT[e2], PT[gb + g], 0, 0, 0);
                }
            }
        }
    }

    PANEL_STAMP(2);
    // ---------------- D = A_dd - P_D
    double D[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 4; ++h) D[m][h] = Din[m][h] - PD[m][h];

    // ---------------- T = (A_sd - P)^T : the strip block goes through LDS to be read transposed
    double T[16];
    if (!is_diag) {
        __syncthreads();
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            *reinterpret_cast<d2_t*>(&sh.Ls[row][c2]) = ld2<TS>(&W[(R0 + row) * ld + c0 + c2]);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 16; ++g) T[g] = sh.Ls[16 * w + c][4 * g + i] - PT[g];
    }

    PANEL_STAMP(3);
    // ---------------- 2. factor the diagonal block (every workgroup); Ld is dead from here on
    __syncthreads();
    mp_factor64(D, Lout, sh.Praw, &sh.fail, sh.rsd, tid, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);

    PANEL_STAMP(4);
    if (is_diag) {
        double* dst = lpark + (size_t)(c0 / 64) * 4096;
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            *reinterpret_cast<d2_t*>(&dst[row * 64 + c2]) = *reinterpret_cast<const d2_t*>(&Lout[row][c2]);
        }
        if (tid == 0 && sh.fail != 0) atomicCAS(info, 0, sh.fail);
        return;
    }

    // ---------------- 2.5 inverses of the sixteen 4x4 diagonal blocks
    if (tid < 16) {
        const int o = 4 * tid;
        const double l00 = Lout[o][o], l10 = Lout[o + 1][o], l11 = Lout[o + 1][o + 1];
        const double l20 = Lout[o + 2][o], l21 = Lout[o + 2][o + 1], l22 = Lout[o + 2][o + 2];
        const double l30 = Lout[o + 3][o], l31 = Lout[o + 3][o + 1], l32 = Lout[o + 3][o + 2],
                     l33 = Lout[o + 3][o + 3];
        (void)l00; (void)l11; (void)l22; (void)l33;
        const double i00 = sh.rsd[o], i11 = sh.rsd[o + 1], i22 = sh.rsd[o + 2], i33 = sh.rsd[o + 3];   // no divisions
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

    PANEL_STAMP(5);
    // ---------------- 3. TRSM: wave w solves its 16 strip rows (lanes c), 4-row slabs p = 0..15
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const double av = sh.Linv[p][ia][i];                       // A[i][k] = inv(L_pp)[i][k], lane (k, blk, i)
        const double Y = __builtin_amdgcn_mfma_f64_4x4x4f64(av, T[p], 0.0, 0, 0, 0);
        T[p] = Y;
        const double Yn = -Y;     // one sign flip per step instead of one per multiplier
#pragma unroll
        for (int g = p + 1; g < 16; ++g) {
            const double al = Lout[4 * g + ia][4 * p + i];     // A[i][k] = -L[4g+i][4p+k]
            T[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(al, Yn, T[g], 0, 0, 0);
        }
    }

    PANEL_STAMP(6);
    // ---------------- 4. write the strip back (transpose through LDS, then 512-B row segments)
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 16; ++g) sh.Ls[16 * w + c][4 * g + i] = T[g];
    __syncthreads();
    for (int e = tid; e < 64 * 32; e += 256) {
        const int row = e >> 5, c2 = (e & 31) * 2;
        st2<TS>(&W[(R0 + row) * ld + c0 + c2], *reinterpret_cast<const d2_t*>(&sh.Ls[row][c2]));
    }
    PANEL_STAMP(7);
}

// Stand-alone TRSM of one 64-column block for rows that are NOT part of a factorisation in progress
// (gpmc_trsm_rows): X = S L_d^-T with L_d given (only its lower triangle is read).  Phases 2.5-4 of
// panel_block_kernel.
struct TrsmShared {
    double Lout[64][MP_LD];
    double Ls[64][MP_LD];
    double Linv[16][4][4];
};

__global__ void __launch_bounds__(256)
trsm_block_kernel(const double* __restrict__ Ld, int ld, double* __restrict__ R, int ldr) {
    extern __shared__ __attribute__((aligned(16))) char trsm_smem[];
    TrsmShared& sh = *reinterpret_cast<TrsmShared*>(trsm_smem);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int i = lane >> 4, c = lane & 15, ia = lane & 3;
    double* Rs = R + (size_t)64 * blockIdx.x * ldr;
    for (int e = tid; e < 64 * 32; e += 256) {
        const int row = e >> 5, c2 = (e & 31) * 2;
        *reinterpret_cast<d2_t*>(&sh.Lout[row][c2]) = *reinterpret_cast<const d2_t*>(&Ld[(size_t)row * ld + c2]);
        *reinterpret_cast<d2_t*>(&sh.Ls[row][c2]) = *reinterpret_cast<const d2_t*>(&Rs[(size_t)row * ldr + c2]);
    }
    __syncthreads();
    if (tid < 16) {
        const int o = 4 * tid;
        const double l00 = sh.Lout[o][o], l10 = sh.Lout[o + 1][o], l11 = sh.Lout[o + 1][o + 1];
        const double l20 = sh.Lout[o + 2][o], l21 = sh.Lout[o + 2][o + 1], l22 = sh.Lout[o + 2][o + 2];
        const double l30 = sh.Lout[o + 3][o], l31 = sh.Lout[o + 3][o + 1], l32 = sh.Lout[o + 3][o + 2],
                     l33 = sh.Lout[o + 3][o + 3];
        const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22, i33 = 1.0 / l33;
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
    double T[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) T[g] = sh.Ls[16 * w + c][4 * g + i];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const double av = sh.Linv[p][ia][i];
        const double Y = __builtin_amdgcn_mfma_f64_4x4x4f64(av, T[p], 0.0, 0, 0, 0);
        T[p] = Y;
        const double Yn = -Y;     // one sign flip per step instead of one per multiplier
#pragma unroll
        for (int g = p + 1; g < 16; ++g) {
            const double al = sh.Lout[4 * g + ia][4 * p + i];
            T[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(al, Yn, T[g], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) sh.Ls[16 * w + c][4 * g + i] = T[g];
    __syncthreads();
    for (int e = tid; e < 64 * 32; e += 256) {
        const int row = e >> 5, c2 = (e & 31) * 2;
        *reinterpret_cast<d2_t*>(&Rs[(size_t)row * ldr + c2]) = *reinterpret_cast<const d2_t*>(&sh.Ls[row][c2]);
    }
}

// Lt[i][j] = L[n-1-j][n-1-i] for j <= i, 0 above: the reverse-transpose J L^T J of a lower-triangular matrix is
// lower triangular, and rows * L^-1 = ((rows J) Lt^-T) J turns the backward substitution into a forward one.
__global__ void __launch_bounds__(256)
revt_lower_kernel(const double* __restrict__ L, int n, int ld, double* __restrict__ Lt, int ldo) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int bi = blockIdx.y, bj = blockIdx.x;               // output tile (rows 32bi.., cols 32bj..)
    if (bj > bi) return;
    // source tile: rows n-1-(32bj+..), cols n-1-(32bi+..)
    for (int r = ty; r < 32; r += 8) {
        const int sr = n - 1 - (32 * bj + r), sc = n - 1 - (32 * bi + tx);
        tile[r][tx] = (sc <= sr) ? L[(size_t)sr * ld + sc] : 0.0;     // tile[r][tx] = L[n-1-(32bj+r)][n-1-(32bi+tx)]
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int oi = 32 * bi + r, oj = 32 * bj + tx;
        Lt[(size_t)oi * ldo + oj] = (oj <= oi) ? tile[tx][r] : 0.0;   // = L[n-1-oj][n-1-oi]
    }
}

template <typename TS>
__global__ void __launch_bounds__(256)
panel_flush_kernel(TS* __restrict__ W, int ld, const double* __restrict__ lpark, int c0) {
    const double* src = lpark + (size_t)(c0 / 64) * 4096;
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int row = e >> 5, c2 = (e & 31) * 2;
        st2<TS>(&W[((size_t)c0 + row) * ld + c0 + c2], *reinterpret_cast<const d2_t*>(&src[row * 64 + c2]));
    }
}

}  // namespace gpmc
