// panel_fused2.h -- fused panel kernel with IN-BLOCK LOOK-AHEAD (f64 arithmetic, TS storage).
//
// Same job and data flow as panel_fused.h (left-looking products, redundant 64x64 factor per workgroup, strip TRSM)
// but 320 threads: waves 0-3 are workers (they own the D slabs and 16 strip rows each, exactly as before), wave 4
// does nothing but the serial micro-panel factor.  s_memtime stamps of the 4-wave kernel showed, per 8-column
// micro-step, ~2060 cycles of factor with three waves idle, then ~760 cycles of rank-8 update and ~460 of
// extraction/barriers with the factor wave idle.  Here, in micro-step t:
//     workers:  U_crit(t-1)  [only the slab column that holds micro-panel t]  -> extract(t) -> B1
//               -> U_rest(t-1) + strip-TRSM for the 8 columns of micro-panel t-1          -> B2
//     wave 4 :                                                                   B1 -> factor(t) -> B2
// so the rest of the update and the whole strip TRSM run in the shadow of the factor.
#pragma once
#include "panel_fused.h"

namespace gpmc {

struct Panel2Shared {
    double Ld[64][MP_LD];       // phase 1: chunk of the diagonal strip's rows; afterwards Lout (the factor)
    double Ls[64][MP_LD];       // chunk of this strip's rows; later the strip block in / out
    double Praw[64][MP_PLD];
    double LinvW[4][2][4][4];   // per worker wave: inverses of the two 4x4 diagonal blocks of the last micro-panel
    double rsd[64];
    int fail;
};

// rank-8 update of worker wave w's slabs with micro-panel TP; CRIT selects the slab column (TP+1)>>1 only
// (the one that contains micro-panel TP+1), otherwise every slab column to its right.
template <int TP, bool CRIT>
__device__ inline void mp2_update(double (&D)[4][4], double (*Lout)[MP_LD], int w, int lane) {
    const int ka = lane >> 4, ia = lane & 3, c = lane & 15;
    constexpr int HC = (TP + 1) >> 1;
#pragma unroll
    for (int m = HC; m < 4; ++m) {
        const int g = w + 4 * m;
        if (g >= 2 * TP + 2) {
            const double a0 = -Lout[4 * g + ia][8 * TP + ka];
            const double a1 = -Lout[4 * g + ia][8 * TP + 4 + ka];
#pragma unroll
            for (int h = HC; h <= m; ++h) {
                if ((h == HC) != CRIT) continue;
                const bool live = 16 * h + c >= 8 * TP + 8;              // finished columns must not move
                const double b0 = live ? Lout[16 * h + c][8 * TP + ka] : 0.0;
                const double b1 = live ? Lout[16 * h + c][8 * TP + 4 + ka] : 0.0;
                D[m][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, D[m][h], 0, 0, 0);
                D[m][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, D[m][h], 0, 0, 0);
            }
        }
    }
}

// strip TRSM for the two 4-row slabs p = 2TP, 2TP+1 (columns of micro-panel TP), by one worker wave
template <int TP>
__device__ inline void mp2_trsm(double (&T)[16], Panel2Shared& sh, double (*Lout)[MP_LD], int w, int lane) {
    const int i = lane >> 4, ia = lane & 3;
    if (lane < 2) {   // inverse of the 4x4 diagonal block 2TP + lane, no divisions (rsd = 1 / L_jj)
        const int o = 4 * (2 * TP + lane);
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
        double(*V)[4] = sh.LinvW[w][lane];
        V[0][0] = i00; V[0][1] = 0.0; V[0][2] = 0.0; V[0][3] = 0.0;
        V[1][0] = i10; V[1][1] = i11; V[1][2] = 0.0; V[1][3] = 0.0;
        V[2][0] = i20; V[2][1] = i21; V[2][2] = i22; V[2][3] = 0.0;
        V[3][0] = i30; V[3][1] = i31; V[3][2] = i32; V[3][3] = i33;
    }
    // same wave writes and reads LinvW[w]: LDS operations of one wave complete in order; the fence only stops the
    // compiler from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        constexpr int P0 = 2 * TP;
        const int p = P0 + q;
        const double av = sh.LinvW[w][q][ia][i];                       // A[i][k] = inv(L_pp)[i][k], lane (k, blk, i)
        const double Y = __builtin_amdgcn_mfma_f64_4x4x4f64(av, T[P0 + q], 0.0, 0, 0, 0);
        T[P0 + q] = Y;
#pragma unroll
        for (int g = P0 + q + 1; g < 16; ++g) {
            const double al = -Lout[4 * g + ia][4 * p + i];            // A[i][k] = -L[4g+i][4p+k]
            T[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(al, Y, T[g], 0, 0, 0);
        }
    }
}

#define MP2_STAMP(k)                                                                                     \
    do {                                                                                                 \
        if (T8 == 3 && dbgp != nullptr && lane == 0) dbgp[8 + 8 * w + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

template <int T8>
__device__ inline void mp2_step(double (&D)[4][4], double (&T)[16], Panel2Shared& sh, double (*Lout)[MP_LD], int w,
                                int lane, bool is_diag, int gcol, unsigned long long* dbgp) {
    constexpr int HT = T8 >> 1;
    constexpr int CLO = 8 * (T8 & 1);
    const int i = lane >> 4, c = lane & 15;
    MP2_STAMP(0);
    if (w < 4) {
        if constexpr (T8 >= 1) mp2_update<T8 - 1, true>(D, Lout, w, lane);
#pragma unroll
        for (int m = HT; m < 4; ++m) {          // extract columns 8T8..8T8+7, rows >= 8T8
            const int g = w + 4 * m;
            if (4 * g + 3 >= 8 * T8 && (c >> 3) == (T8 & 1)) sh.Praw[4 * g + i][c - CLO] = D[m][HT];
        }
    }
    MP2_STAMP(1);
    __syncthreads();   // B1: micro-panel T8 is in Praw
    MP2_STAMP(2);
    if (w == 4) {
        const int row = lane;
        double p[8], rsv[8];
        int bad = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = sh.Praw[row][k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pr = 8 * T8 + k;
            const double d = mp_readlane(p[k], pr);
            bad = (bad == 0 && !(d > 0.0)) ? (pr + 1) : bad;
            const double rs = mp_rsqrt(d);
            rsv[k] = rs;
            const double lk = p[k] * rs;
            p[k] = lk;
#pragma unroll
            for (int k2 = k + 1; k2 < 8; ++k2) {
                const double mlt = mp_readlane(lk, 8 * T8 + k2);
                p[k2] = __builtin_fma(-lk, mlt, p[k2]);
                asm volatile("" : "+v"(p[k2]));
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) Lout[row][8 * T8 + k] = (row < 8 * T8 + k) ? 0.0 : p[k];
        if (lane < 8) {
            double rsel = rsv[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) rsel = (lane == k) ? rsv[k] : rsel;
            sh.rsd[8 * T8 + lane] = rsel;
        }
        if (lane == 0 && bad != 0 && sh.fail == 0) sh.fail = gcol + bad;
    } else {
        if constexpr (T8 >= 1) {
            mp2_update<T8 - 1, false>(D, Lout, w, lane);
            if (!is_diag) mp2_trsm<T8 - 1>(T, sh, Lout, w, lane);
        }
    }
    MP2_STAMP(3);
    __syncthreads();   // B2: columns 8T8..8T8+7 of L are in Lout
    MP2_STAMP(4);
}

template <bool DBG, typename TS>
__global__ void __launch_bounds__(320)
panel_block2_kernel(TS* __restrict__ W, int ld, int k0, int c0, int* __restrict__ info, double* __restrict__ lpark,
                    int prev_c0, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char panel2_smem[];
    Panel2Shared& sh = *reinterpret_cast<Panel2Shared*>(panel2_smem);
    double (*Lout)[MP_LD] = sh.Ld;   // aliased: valid once phase 1 is over
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int i = lane >> 4, c = lane & 15, ia = lane & 3;
    const bool worker = (w < 4);
    const bool is_diag = (blockIdx.x == 0);
    const size_t R0 = (size_t)c0 + 64 * (size_t)blockIdx.x;
    PANEL_STAMP(0);
    if (tid == 0) sh.fail = 0;

    if (is_diag && prev_c0 >= 0 && worker) {   // flush the previous block's factor
        const double* src = lpark + (size_t)(prev_c0 / 64) * 4096;
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            st2<TS>(&W[((size_t)prev_c0 + row) * ld + prev_c0 + c2], *reinterpret_cast<const d2_t*>(&src[row * 64 + c2]));
        }
    }
    double PD[4][4], PT[16], Din[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 4; ++h) { PD[m][h] = 0.0; Din[m][h] = 0.0; }
#pragma unroll
    for (int g = 0; g < 16; ++g) PT[g] = 0.0;
    if (worker) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int h = 0; h < 4; ++h)
                if (h <= m) Din[m][h] = (double)W[((size_t)c0 + 4 * (w + 4 * m) + i) * ld + c0 + 16 * h + c];
    }
    PANEL_STAMP(1);

    // ---------------- 1. left-looking products (workers), next chunk prefetched into registers
    d2_t pd[8], ps[8];
    auto fetch_chunk = [&](int kc) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int e = tid + 256 * t, row = e >> 5, c2 = (e & 31) * 2;
            pd[t] = ld2<TS>(&W[((size_t)c0 + row) * ld + kc + c2]);
            if (!is_diag) ps[t] = ld2<TS>(&W[(R0 + row) * ld + kc + c2]);
        }
    };
    if (k0 < c0 && worker) fetch_chunk(k0);
    for (int kc = k0; kc < c0; kc += 64) {
        __syncthreads();
        if (worker) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int e = tid + 256 * t, row = e >> 5, c2 = (e & 31) * 2;
                *reinterpret_cast<d2_t*>(&sh.Ld[row][c2]) = pd[t];
                if (!is_diag) *reinterpret_cast<d2_t*>(&sh.Ls[row][c2]) = ps[t];
            }
        }
        __syncthreads();
        if (worker) {
            if (kc + 64 < c0) fetch_chunk(kc + 64);
#pragma unroll 2
            for (int k8 = 0; k8 < 8; ++k8) {
                const int ko = 8 * k8 + 2 * i;
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
    }
    PANEL_STAMP(2);

    double D[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 4; ++h) D[m][h] = Din[m][h] - PD[m][h];
    double T[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) T[g] = 0.0;
    if (!is_diag) {
        __syncthreads();
        if (worker)
            for (int e = tid; e < 64 * 32; e += 256) {
                const int row = e >> 5, c2 = (e & 31) * 2;
                *reinterpret_cast<d2_t*>(&sh.Ls[row][c2]) = ld2<TS>(&W[(R0 + row) * ld + c0 + c2]);
            }
        __syncthreads();
        if (worker) {
#pragma unroll
            for (int g = 0; g < 16; ++g) T[g] = sh.Ls[16 * w + c][4 * g + i] - PT[g];
        }
    }
    PANEL_STAMP(3);
    __syncthreads();   // Ld is dead from here on (becomes Lout)

    // ---------------- 2+3. factor with in-block look-ahead; strip TRSM in the factor's shadow.
    // Five waves on four SIMDs: the factor wave is the youngest wave of its SIMD and would lose VALU issue
    // arbitration (priority, then age) to the worker it shares with -- give it static priority.
    if (__builtin_amdgcn_readfirstlane(w) == 4) __builtin_amdgcn_s_setprio(3);
    mp2_step<0>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    mp2_step<1>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    mp2_step<2>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    mp2_step<3>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    mp2_step<4>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    mp2_step<5>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    mp2_step<6>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    mp2_step<7>(D, T, sh, Lout, w, lane, is_diag, c0, (DBG && blockIdx.x == 1) ? dbg : nullptr);
    PANEL_STAMP(4);

    if (is_diag) {
        if (worker) {
            double* dst = lpark + (size_t)(c0 / 64) * 4096;
            for (int e = tid; e < 64 * 32; e += 256) {
                const int row = e >> 5, c2 = (e & 31) * 2;
                *reinterpret_cast<d2_t*>(&dst[row * 64 + c2]) = *reinterpret_cast<const d2_t*>(&Lout[row][c2]);
            }
        }
        if (tid == 0 && sh.fail != 0) atomicCAS(info, 0, sh.fail);
        return;
    }
    if (worker) mp2_trsm<7>(T, sh, Lout, w, lane);   // the last two slabs
    PANEL_STAMP(5);
    PANEL_STAMP(6);
    // ---------------- 4. write the strip back (transpose through LDS)
    __syncthreads();
    if (worker) {
#pragma unroll
        for (int g = 0; g < 16; ++g) sh.Ls[16 * w + c][4 * g + i] = T[g];
    }
    __syncthreads();
    if (worker)
        for (int e = tid; e < 64 * 32; e += 256) {
            const int row = e >> 5, c2 = (e & 31) * 2;
            st2<TS>(&W[(R0 + row) * ld + c0 + c2], *reinterpret_cast<const d2_t*>(&sh.Ls[row][c2]));
        }
    PANEL_STAMP(7);
}

}  // namespace gpmc
