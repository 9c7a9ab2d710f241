// gemm4p.h -- PERSISTENT pipelined variant of the FP64 slab GEMM (gemm4.h) for the trailing SYRK update.
//
// Measurement behind it: at K = 128 a full-size trailing update spends 0.15 ms in its K loops and
// ~0.075 ms reading/writing the C tiles, and nothing overlaps the two -- every workgroup loads its C tile, runs a
// short K loop, stores.  Here one 512-thread workgroup per CU (2 waves per SIMD, wave tile 32 x 64 = 32 slabs = 64
// accumulator VGPRs) walks a list of tiles; while tile t is in its K loop the C values of tile t+1 are already
// streaming into a SECOND accumulator set, tile t's results are stored right after its last stage, and the A/B
// staging pipeline (global -> registers -> LDS, double-buffered) runs across tile boundaries without a bubble.
#pragma once
#include "gpmc_common.h"

namespace gpmc {

constexpr int GP_ROWB = 144;
constexpr int GP_BM = 128, GP_BN = 128, GP_NT = 512, GP_RG = 8, GP_CG = 4;

struct TileXY { int row0, col0; };

__device__ inline TileXY gp_tile(int t, int ntc, int tri) {
    int ti, tj;
    if (tri) {
        const int T3 = ntc * (ntc + 1) / 2;
        if (t < T3) {
            ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            tj = t - ti * (ti + 1) / 2;
        } else {
            const int bb = t - T3;
            ti = ntc + bb / ntc;
            tj = bb % ntc;
        }
    } else {
        ti = t / ntc;
        tj = t % ntc;
    }
    return TileXY{ti * GP_BM, tj * GP_BN};
}

template <bool EPI>
__global__ void __launch_bounds__(GP_NT, 1)
gemm4p_nt_kernel(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, const double* Cin,
                 int ldcin, double* C, int ldc, int K, int ntc, int tri, int ntiles, double alpha, int n_true,
                 double diag_add) {
    extern __shared__ __attribute__((aligned(16))) char smemp[];
    char* As = smemp;                          // [2][BM][ROWB]
    char* Bs = smemp + 2 * GP_BM * GP_ROWB;    // [2][BN][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;       // 4 x 2 waves, wave tile 32 x 64
    const int i = lane >> 4, c = lane & 15, ia = lane & 3;
    const int nk = K / 16;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    double accA[GP_RG][GP_CG], accB[GP_RG][GP_CG];
    d2_t pa[2], pb[2];

    auto c_load = [&](double (&acc)[GP_RG][GP_CG], TileXY t) {
        const int crow = t.row0 + wm * 32 + i, ccol = t.col0 + wn * 64 + c;
#pragma unroll
        for (int rg = 0; rg < GP_RG; ++rg)
#pragma unroll
            for (int cg = 0; cg < GP_CG; ++cg) acc[rg][cg] = Cin[(size_t)(crow + 4 * rg) * ldcin + ccol + 16 * cg];
    };
    auto c_store = [&](double (&acc)[GP_RG][GP_CG], TileXY t) {
        const int crow = t.row0 + wm * 32 + i, ccol = t.col0 + wn * 64 + c;
#pragma unroll
        for (int rg = 0; rg < GP_RG; ++rg)
#pragma unroll
            for (int cg = 0; cg < GP_CG; ++cg) {
                const int gr = crow + 4 * rg, gc = ccol + 16 * cg;
                double v = acc[rg][cg];
                if (EPI) {
                    if (gr == gc) v += diag_add;
                    if (gr >= n_true || gc >= n_true) v = (gr == gc) ? 1.0 : 0.0;
                }
                C[(size_t)gr * ldc + gc] = v;
            }
    };
    auto stage_load = [&](TileXY t, int kt) {   // global -> registers for stage kt of tile t
        const double* Ag = A + (size_t)t.row0 * lda + kt * 16;
        const double* Bg = B + (size_t)t.col0 * ldb + kt * 16;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = tid + GP_NT * u;
            pa[u] = *reinterpret_cast<const d2_t*>(Ag + (size_t)(ch >> 3) * lda + (ch & 7) * 2);
            pb[u] = *reinterpret_cast<const d2_t*>(Bg + (size_t)(ch >> 3) * ldb + (ch & 7) * 2);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = tid + GP_NT * u;
            d2_t v = pa[u];
            v[0] *= alpha; v[1] *= alpha;
            *reinterpret_cast<d2_t*>(As + ((size_t)buf * GP_BM + (ch >> 3)) * GP_ROWB + (ch & 7) * 16) = v;
            *reinterpret_cast<d2_t*>(Bs + ((size_t)buf * GP_BN + (ch >> 3)) * GP_ROWB + (ch & 7) * 16) = pb[u];
        }
    };
    auto compute = [&](double (&acc)[GP_RG][GP_CG], int buf) {
        const char* Ab = As + ((size_t)buf * GP_BM + wm * 32 + ia) * GP_ROWB + i * 16;
        const char* Bb = Bs + ((size_t)buf * GP_BN + wn * 64 + c) * GP_ROWB + i * 16;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            d2_t bf[GP_CG];
#pragma unroll
            for (int cg = 0; cg < GP_CG; ++cg) bf[cg] = *reinterpret_cast<const d2_t*>(Bb + cg * 16 * GP_ROWB + kc * 64);
#pragma unroll
            for (int rg = 0; rg < GP_RG; ++rg) {
                const d2_t af = *reinterpret_cast<const d2_t*>(Ab + rg * 4 * GP_ROWB + kc * 64);
#pragma unroll
                for (int cg = 0; cg < GP_CG; ++cg)
                    acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[0], bf[cg][0], acc[rg][cg], 0, 0, 0);
#pragma unroll
                for (int cg = 0; cg < GP_CG; ++cg)
                    acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[1], bf[cg][1], acc[rg][cg], 0, 0, 0);
            }
        }
    };

    // one tile: `cur` is computed and stored, `nxt` receives the next tile's C meanwhile; stage (tile, 0) is already
    // in LDS buffer `buf0` on entry, and on exit stage (next tile, 0) is in buffer (buf0 + nk) & 1.
    auto run_tile = [&](double (&cur)[GP_RG][GP_CG], double (&nxt)[GP_RG][GP_CG], TileXY t, bool has_next, TileXY tn,
                        int buf0) {
        if (has_next) c_load(nxt, tn);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = (buf0 + kt) & 1;
            const bool more = (kt + 1 < nk) || has_next;
            if (more) {
                if (kt + 1 < nk) stage_load(t, kt + 1);
                else stage_load(tn, 0);
            }
            compute(cur, buf);
            if (more) stage_store(buf ^ 1);
            __syncthreads();
        }
        c_store(cur, t);
    };

    TileXY t0 = gp_tile(blockIdx.x, ntc, tri);
    c_load(accA, t0);
    stage_load(t0, 0);
    stage_store(0);
    __syncthreads();
    int buf0 = 0;
    for (int n = 0; n < my_tiles; n += 2) {
        const int ta = (int)blockIdx.x + n * (int)gridDim.x;
        const bool hb = (n + 1 < my_tiles);
        const TileXY txa = gp_tile(ta, ntc, tri);
        const TileXY txb = hb ? gp_tile(ta + (int)gridDim.x, ntc, tri) : txa;
        run_tile(accA, accB, txa, hb, txb, buf0);
        buf0 = (buf0 + nk) & 1;
        if (!hb) break;
        const bool hc = (n + 2 < my_tiles);
        const TileXY txc = hc ? gp_tile(ta + 2 * (int)gridDim.x, ntc, tri) : txb;
        run_tile(accB, accA, txb, hc, txc, buf0);
        buf0 = (buf0 + nk) & 1;
    }
}

}  // namespace gpmc
