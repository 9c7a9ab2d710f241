// diag_mp.h -- 64x64 Cholesky by one 256-thread workgroup: 8-column micro-panels + v_mfma_f64_4x4x4 updates.
//
// FP64 VALU instructions cost ~9 cycles per wave-instruction at one wave per SIMD, so any scheme that
// applies the rank-1 updates on the VALU spends ~1000 cycles per column.  Here the block lives in MFMA
// accumulators as 4x16 "slabs" (lane (i = l>>4, c = l&15) of slab (g,h) = element [4g+i][16h+c], which
// is both the D layout of v_mfma_f64_4x4x4_4b_f64 and one register of the 16x16x4 accumulator layout);
// wave w owns slab rows g = w, w+4, w+8, w+12 (10 lower slabs).  Per micro-step t (columns 8t..8t+7):
//   1. the owners copy those 8 columns (rows >= 8t) into LDS,
//   2. wave 0 factors the 64x8 micro-panel with lane = row: pivots by v_readlane, 1/sqrt by v_rsq_f64 +
//      two Goldschmidt steps, <= 7 FMAs per column; writes the columns of L into LDS,
//   3. all waves apply the rank-8 update to their slabs straight from those columns: 2 MFMAs (16 cycles
//      each) per slab.  MFMA operand maps (measured, scripts/mfma_probe.hip):
//      A[i][k] lane 16k+4blk+i, B[k][j] lane 16k+4blk+j, D[i][j] lane 16i+4blk+j.
// f64 only (the f32 path keeps the generic kernels).
// Measured and not kept (round 3): 4-column micro-panels -- half the serial VALU stream per 64 columns (16 x (4 rsq chains +
// 6 updates + 20 readlanes) instead of 8 x (8 + 28 + 72); the updates of columns 4..7 by 0..3 become one more MFMA per
// slab) for twice the barriers and extractions: correct (all kernel tests), N = 2048 / 4096 / 8192 0.587 / 1.203 / 4.262 ms
// against 0.573-0.583 / 1.186-1.193 / 4.234-4.27: the two barriers and the LDS round trip of a micro-step cost what the
// shorter stream saves.
#pragma once
#include "gpmc_common.h"

namespace gpmc {

constexpr int MP_LD = 72;   // row stride (doubles): 576 B = 16 banks mod 64 -> the 4 rows x 4 k-pairs of a replicated A read hit 16 distinct bank groups
constexpr int MP_PLD = 9;   // micro-panel row stride (doubles)

__device__ inline double mp_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d): v_rsq_f64 seed + Goldschmidt refinement (no division, no denormal fix-up path).  MP_RSQ_ITERS = 2
// is accurate to ~1 ulp; 1 iteration is selectable for measurement (-DMP_RSQ_ITERS=1).
#ifndef MP_RSQ_ITERS
#define MP_RSQ_ITERS 2
#endif
__device__ inline double mp_rsqrt(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);
#if MP_RSQ_ITERS >= 2
    g = __builtin_fma(g, r, g);
    r = __builtin_fma(-h, g, 0.5);
    const double h2 = h + h;                 // off the dependent chain: 2h(1 + r) in one fma
    return __builtin_fma(h2, r, h2);
#else
    return h + h;
#endif
}

// MP_FASTPIVOT = 1 (experiment, compiled out): form pivot k+1 ahead of the vector update, d' = A - B^2 / d with
// A = p[k+1](row k+1), B = p[k](row k+1) read BEFORE 1/sqrt(d) is known (both are final one column earlier); the
// multipliers are then the uniform B rs and no v_readlane sits between two pivots.  Measured: the factor wave's
// micro-step goes from ~2260 to 2104 ticks, the block's factor phase stays at ~24K and N = 2048 / 4096 / 8192 do not
// move (0.92 / 2.24 / 6.83 ms).  The ISA explains it: ~260 instructions per micro-step at ~8 cycles each -- FP64 VALU
// instructions issue every ~8 cycles at one wave per SIMD even when independent -- so this phase is bound by
// instruction issue, and the variant trades a shorter chain for more FP64 instructions (20 vs 13.5 per column).
#ifndef MP_FASTPIVOT
#define MP_FASTPIVOT 0
#endif

// Slabs of wave w: D[m][h], m = 0..3 (slab row g = w + 4m), h = 0..m.
// One micro-step, T8 compile-time.  All 256 threads call; two barriers inside.
// dbgp != nullptr only in the diagnostic instantiation (stamps of micro-step 3 by every wave's lane 0)
#define MP_STAMP(k)                                                                                     \
    do {                                                                                                \
        if (T8 == 3 && dbgp != nullptr && lane == 0) dbgp[8 + 8 * w + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

template <int T8>
__device__ inline void mp_micro_step(double (&D)[4][4], double (*Lout)[MP_LD], double (*Praw)[MP_PLD], int* failp,
                                     double* rsd, int w, int lane, int gcol, unsigned long long* dbgp = nullptr) {
    constexpr int HT = T8 >> 1;            // slab column that holds columns 8T8..8T8+7
    constexpr int CLO = 8 * (T8 & 1);      // first lane-column of the micro-panel inside that slab
    const int i = lane >> 4, c = lane & 15;
    MP_STAMP(0);
    // ---- 1. extract columns 8T8..8T8+7, rows >= 8T8
#pragma unroll
    for (int m = HT; m < 4; ++m) {
        const int g = w + 4 * m;
        if (4 * g + 3 >= 8 * T8 && (c >> 3) == (T8 & 1)) Praw[4 * g + i][c - CLO] = D[m][HT];
    }
    MP_STAMP(1);
    __syncthreads();
    MP_STAMP(2);
    // ---- 2. wave 0: factor the micro-panel, lane = row
    if (w == 0) {
        const int row = lane;
        double p[8], rsv[8];
        int bad = 0;                                         // wave-uniform: 1-based first non-positive pivot
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = Praw[row][k];
#if MP_FASTPIVOT
        double d = mp_readlane(p[0], 8 * T8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pr = 8 * T8 + k;                       // pivot row == pivot column
            // uniform pre-reads from the unscaled column k and from column k+1 (both final w.r.t. columns < k)
            double B[8];
#pragma unroll
            for (int k2 = k + 1; k2 < 8; ++k2) B[k2] = mp_readlane(p[k], 8 * T8 + k2);
            const double A = (k < 7) ? mp_readlane(p[(k < 7) ? k + 1 : k], pr + 1) : 0.0;
            bad = (bad == 0 && !(d > 0.0)) ? (pr + 1) : bad;
            const double rs = mp_rsqrt(d);
            rsv[k] = rs;
            if (k < 7) {
                const double b2 = B[(k < 7) ? k + 1 : k] * B[(k < 7) ? k + 1 : k];
                d = __builtin_fma(-b2, rs * rs, A);          // next pivot: the only thing the next rsq waits for
            }
            const double lk = p[k] * rs;                     // row pr: d * rs = sqrt(d) to ~1 ulp; rows above: junk, zeroed below
            p[k] = lk;
#pragma unroll
            for (int k2 = k + 1; k2 < 8; ++k2) {
                const double mlt = B[k2] * rs;               // L[8T8+k2][8T8+k], uniform
                p[k2] = __builtin_fma(-lk, mlt, p[k2]);
                asm volatile("" : "+v"(p[k2]));
            }
        }
#else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pr = 8 * T8 + k;                       // pivot row == pivot column
            const double d = mp_readlane(p[k], pr);
            bad = (bad == 0 && !(d > 0.0)) ? (pr + 1) : bad;
            const double rs = mp_rsqrt(d);
            rsv[k] = rs;
            const double lk = p[k] * rs;                     // row pr: d * rs = sqrt(d) to ~1 ulp; rows above: junk, zeroed below
            p[k] = lk;
            // right-looking: the 7-k independent rank-1 updates issue back to back; only p[k+1]'s is on the pivot
            // chain.  The empty asm pins this order (hipcc otherwise sinks all k updates of column k in front of
            // pivot k as one dependent FMA chain).
#pragma unroll
            for (int k2 = k + 1; k2 < 8; ++k2) {
                const double mlt = mp_readlane(lk, 8 * T8 + k2);   // L[8T8+k2][8T8+k]
                p[k2] = __builtin_fma(-lk, mlt, p[k2]);
                asm volatile("" : "+v"(p[k2]));
            }
        }
#endif
#pragma unroll
        for (int k = 0; k < 8; ++k) Lout[row][8 * T8 + k] = (row < 8 * T8 + k) ? 0.0 : p[k];
        if (lane < 8) {
            double rsel = rsv[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) rsel = (lane == k) ? rsv[k] : rsel;
            rsd[8 * T8 + lane] = rsel;                       // 1 / L[j][j], reused by the 4x4 inverses
        }
        if (lane == 0 && bad != 0 && *failp == 0) *failp = gcol + bad;
    }
    MP_STAMP(3);
    __syncthreads();
    MP_STAMP(4);
    // ---- 3. rank-8 update of the slabs below / right of the micro-panel
    if (T8 < 7) {
        const int ka = lane >> 4, ia = lane & 3;             // A operand: lane (k, blk, i)
        constexpr int HLO = (T8 + 1) >> 1;                   // first slab column with columns >= 8T8+8
#pragma unroll
        for (int m = HLO; m < 4; ++m) {
            const int g = w + 4 * m;
            if (g >= 2 * T8 + 2) {
                const double a0 = -Lout[4 * g + ia][8 * T8 + ka];
                const double a1 = -Lout[4 * g + ia][8 * T8 + 4 + ka];
#pragma unroll
                for (int h = HLO; h <= m; ++h) {
                    const bool live = 16 * h + c >= 8 * T8 + 8;           // finished columns must not move
                    const double b0 = live ? Lout[16 * h + c][8 * T8 + i] : 0.0;   // B operand: lane (k = l>>4, col)
                    const double b1 = live ? Lout[16 * h + c][8 * T8 + 4 + i] : 0.0;
                    D[m][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, D[m][h], 0, 0, 0);
                    D[m][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, D[m][h], 0, 0, 0);
                }
            }
        }
    }
    MP_STAMP(5);
}

// Factor the block held in slabs; on return Lout holds L (strict upper zero).  gcol = global column of the block.
__device__ inline void mp_factor64(double (&D)[4][4], double (*Lout)[MP_LD], double (*Praw)[MP_PLD], int* failp,
                                   double* rsd, int tid, int gcol, unsigned long long* dbgp = nullptr) {
    const int w = tid >> 6, lane = tid & 63;
    if (tid == 0) *failp = 0;
    __syncthreads();
    mp_micro_step<0>(D, Lout, Praw, failp, rsd, w, lane, gcol);
    mp_micro_step<1>(D, Lout, Praw, failp, rsd, w, lane, gcol);
    mp_micro_step<2>(D, Lout, Praw, failp, rsd, w, lane, gcol);
    mp_micro_step<3>(D, Lout, Praw, failp, rsd, w, lane, gcol, dbgp);
    mp_micro_step<4>(D, Lout, Praw, failp, rsd, w, lane, gcol);
    mp_micro_step<5>(D, Lout, Praw, failp, rsd, w, lane, gcol);
    mp_micro_step<6>(D, Lout, Praw, failp, rsd, w, lane, gcol);
    mp_micro_step<7>(D, Lout, Praw, failp, rsd, w, lane, gcol);
    __syncthreads();
}

// Standalone launch (drop-in for the other diag kernels): also emits the 16x16 inverse tiles trsm_strip uses.
__global__ void __launch_bounds__(256)
diag_factor_mp_kernel(double* __restrict__ Wd, int ld, double* __restrict__ dinv, int* __restrict__ info, int c0) {
    __shared__ double Lout[64][MP_LD];
    __shared__ double Praw[64][MP_PLD];
    __shared__ double rsd[64];
    __shared__ int sfail;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int i = lane >> 4, c = lane & 15;
    double D[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 4; ++h)
            D[m][h] = (h <= m) ? Wd[(size_t)(4 * (w + 4 * m) + i) * ld + 16 * h + c] : 0.0;
    mp_factor64(D, Lout, Praw, &sfail, rsd, tid, c0);
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e >> 6, cc = e & 63;
        Wd[(size_t)r * ld + cc] = Lout[r][cc];
    }
    if (tid < 64) {  // inverse of diagonal tile t, column jj, by forward substitution
        const int t = tid >> 4, jj = tid & 15, o = 16 * t;
        double x[16];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            double s = (ii == jj) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < ii; ++k) s -= Lout[o + ii][o + k] * x[k];
            x[ii] = s / Lout[o + ii][o + ii];
        }
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) dinv[t * 256 + ii * 16 + jj] = x[ii];
    }
    if (tid == 0 && sfail != 0) atomicCAS(info, 0, sfail);
}

}  // namespace gpmc
