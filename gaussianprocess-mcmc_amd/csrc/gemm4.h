// gemm4.h -- FP64 NT GEMM / SYRK on v_mfma_f64_4x4x4_4b_f64.
//
// Measured on MI355X (scripts/mfma_peak.hip): v_mfma_f64_16x16x4_f64 saturates at ~49 TF/s, the 4-block
// v_mfma_f64_4x4x4_4b_f64 at ~74 TF/s (the 78.6 TF/s FP64 spec rate).  One 4x4x4 instruction multiplies four
// independent (4x4)(4x4) block pairs; feeding all four blocks the SAME A block (lane 16k+4blk+i reads A[i][k]
// whatever blk is -- an LDS broadcast) and four adjacent B blocks turns it into a (4x4)(4x16) product whose
// result, lane 16i + c -> C[i][c], is a 4-row x 16-column slab: the same register image as one register of the
// 16x16x4 accumulator, with 128-byte row segments for the C read-modify-write.
//   wave tile  = RG row-groups (4 rows) x CG column-groups (16 columns): RG*CG independent accumulators
//   per 8 k    = RG + CG 16-byte LDS reads for 2*RG*CG MFMAs (16 cycles each)
// Same interface and tile decode as gemm_nt_kernel (potrf.hip).
// Measured and removed in round 3 (DESIGN.md section 4 keeps the numbers): an XCD-aware 8x8 super-block tile order (halves
// the panel re-reads that miss the XCD L2s, 253 vs 372 MB fetched per K = 256 launch, and is 12 % slower from the
// super-block imbalance), folding C in after the K loop on alternate dispatch waves, start-up phase shifts of co-resident
// workgroups, and non-temporal C accesses: all +-0 or worse.
#pragma once
#include "gpmc_common.h"

namespace gpmc {

constexpr int G4_ROWB = 144;  // LDS bytes per tile row at KCH = 2: 128 data + 16 pad (KCH = 4: 256 + 16)

// KCH = number of 8-k chunks per LDS stage (2 -> BK = 16, 4 -> BK = 32).  The small-tile configuration uses 4: its
// workgroups are latency-bound (one global round trip per stage for ~0.3 us of MFMA work), so fewer, fatter stages.
template <int WGM, int WGN, int RG, int CG, bool EPI, int KCH = 2>
__global__ void __launch_bounds__(WGM * WGN * 64, (WGM * WGN == 4) ? 2 : 1)
gemm4_nt_kernel(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, const double* Cin,
                int ldcin, double* C, int ldc, int K, int ntc, int tri, double alpha, int beta, int n_true,
                double diag_add, int ktri, int prio_cols, unsigned* __restrict__ gflag, double* C2) {
    constexpr int BM = WGM * RG * 4, BN = WGN * CG * 16;
    constexpr int NT = WGM * WGN * 64;                     // threads per workgroup (256 or 512)
    constexpr int BKE = 8 * KCH;                           // k elements per stage
    constexpr int CPR = 4 * KCH;                           // 16-byte chunks per tile row
    constexpr int ROWB = 64 * KCH + 16;                    // LDS bytes per tile row
    constexpr int NA = BM * CPR / NT, NBI = BN * CPR / NT; // 16-byte chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem4[];
    char* As = smem4;                       // [2][BM][G4_ROWB]
    char* Bs = smem4 + 2 * BM * ROWB;       // [2][BN][ROWB]

    int ti, tj;
    bool is_prio = false;
    int prio_slot = 0;
    if (tri && prio_cols > 0) {
        // Look-ahead hand-over inside the launch (potrf.hip, la_dag): the first prio_cols tile columns -- the columns of
        // the panel after next -- are processed FIRST (blocks are dispatched in index order) and every such tile
        // publishes a flag when it is in memory, so the tile-DAG launch that factors that panel starts on them while the
        // rest of this update is still running.  The remaining tiles follow in the usual order on the shifted trapezoid.
        const int b = blockIdx.x;
        const int T3 = ntc * (ntc + 1) / 2;
        const int mt = ntc + ((int)gridDim.x - T3) / ntc;          // tile rows of the whole region
        int P = 0;
        for (int j = 0; j < prio_cols; ++j) P += mt - j;
        if (b < P) {
            tj = 0;
            int q = b;
            while (q >= mt - tj) { q -= mt - tj; ++tj; }
            ti = tj + q;
            is_prio = true;
            prio_slot = tj * mt + ti;
        } else {
            const int bb = b - P, n2 = ntc - prio_cols;
            const int T32 = n2 * (n2 + 1) / 2;
            int t2, j2;
            if (bb < T32) {
                t2 = (int)((sqrt(8.0 * (double)bb + 1.0) - 1.0) * 0.5);
                while (t2 * (t2 + 1) / 2 > bb) --t2;
                while ((t2 + 1) * (t2 + 2) / 2 <= bb) ++t2;
                j2 = bb - t2 * (t2 + 1) / 2;
            } else {
                const int b3 = bb - T32;
                t2 = n2 + b3 / n2;
                j2 = b3 % n2;
            }
            ti = t2 + prio_cols;
            tj = j2 + prio_cols;
        }
    } else {
        const int b = blockIdx.x;
        if (tri) {
            const int T3 = ntc * (ntc + 1) / 2;
            if (b < T3) {
                ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
                while (ti * (ti + 1) / 2 > b) --ti;
                while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
                tj = b - ti * (ti + 1) / 2;
            } else {
                const int bb = b - T3;
                ti = ntc + bb / ntc;
                tj = bb % ntc;
            }
        } else {
            ti = b / ntc;
            tj = b % ntc;
        }
    }
    const int row0 = ti * BM, col0 = tj * BN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WGN, wn = wv % WGN;
    const int i = lane >> 4, c = lane & 15, ia = lane & 3;   // i doubles as the operands' k index

    double acc[RG][CG];
    const int crow = row0 + wm * RG * 4 + i;        // + 4*rg
    const int ccol = col0 + wn * CG * 16 + c;       // + 16*cg
    if (beta) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const double* cp = &Cin[(size_t)(crow + 4 * rg) * ldcin + ccol + 16 * cg];
                acc[rg][cg] = *cp;
            }
    } else {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) acc[rg][cg] = 0.0;
    }

    // ktri: both operands are upper-trapezoidal (row i is zero left of column i, e.g. L^-T): for a lower tile the
    // products start at k = row0 (>= col0); skipping the leading zeros makes a full SYRK cost N^3/3 instead of N^3
    const int kbeg = ktri ? row0 : 0;
    const double* Ag = A + (size_t)row0 * lda + kbeg;
    const double* Bg = B + (size_t)col0 * ldb + kbeg;
    d2_t pa[NA], pb[NBI];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int ch = tid + NT * t;
            pa[t] = *reinterpret_cast<const d2_t*>(Ag + (size_t)(ch / CPR) * lda + k0 + (ch % CPR) * 2);
        }
#pragma unroll
        for (int t = 0; t < NBI; ++t) {
            const int ch = tid + NT * t;
            pb[t] = *reinterpret_cast<const d2_t*>(Bg + (size_t)(ch / CPR) * ldb + k0 + (ch % CPR) * 2);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int ch = tid + NT * t;
            d2_t v = pa[t];
            v[0] *= alpha; v[1] *= alpha;
            *reinterpret_cast<d2_t*>(As + ((size_t)buf * BM + (ch / CPR)) * ROWB + (ch % CPR) * 16) = v;
        }
#pragma unroll
        for (int t = 0; t < NBI; ++t) {
            const int ch = tid + NT * t;
            *reinterpret_cast<d2_t*>(Bs + ((size_t)buf * BN + (ch / CPR)) * ROWB + (ch % CPR) * 16) = pb[t];
        }
    };

    const int nk = (K - kbeg) / BKE;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BKE);
        // A: lane (k = l>>4, blk, i = l&3) reads row 4rg+i ; B: lane (k = l>>4, col = l&15) reads row 16cg+col
        const char* Ab = As + ((size_t)cur * BM + wm * RG * 4 + ia) * ROWB + i * 16;
        const char* Bb = Bs + ((size_t)cur * BN + wn * CG * 16 + c) * ROWB + i * 16;
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            d2_t bf[CG];
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) bf[cg] = *reinterpret_cast<const d2_t*>(Bb + cg * 16 * ROWB + kc * 64);
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const d2_t af = *reinterpret_cast<const d2_t*>(Ab + rg * 4 * ROWB + kc * 64);
                // the two k-halves of one accumulator are issued CG instructions apart (no back-to-back dependence)
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
                    acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[0], bf[cg][0], acc[rg][cg], 0, 0, 0);
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
                    acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[1], bf[cg][1], acc[rg][cg], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const int gr = crow + 4 * rg, gc = ccol + 16 * cg;
            double v = acc[rg][cg];
            if (EPI) {
                if (gr == gc) v += diag_add;
                if (gr >= n_true || gc >= n_true) v = (gr == gc) ? 1.0 : 0.0;
            }
            C[(size_t)gr * ldc + gc] = v;
            if (EPI) if (C2) C2[(size_t)gr * ldc + gc] = v;      // second copy of the result (gpmc_syrk_tri2), same ld
        }
    if (is_prio) {
        // publish: plain stores -> every storing wave drains -> workgroup barrier -> one lane's agent-scope release
        // (L2 write-back), drained -> relaxed agent flag store (CDNA4 guide, inter-workgroup visibility, valid forms)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(gflag + prio_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace gpmc
