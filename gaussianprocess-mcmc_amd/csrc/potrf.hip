// potrf.hip -- blocked right-looking Cholesky on MFMA.
//
// Replaces kcGP.tools.jitchol (kcMCMC/sliceSample.py:196,205: LAPACK dpotrf) and, through extra rows
// riding below the matrix, np.linalg.solve(L, .) (sliceSample.py:108,197): factoring the leading np_
// columns of the (mp x np_) lower trapezoid [A; B^T] leaves L in the triangle and B^T L^-T = (L^-1 B)^T
// in the extra rows, so the forward substitution of the log-marginal (alpha = L^-1 g) and the N-RHS
// solve V = L^-1 K cost no extra kernels.
//
// Structure (row-major, lower):
//   outer block NB (256): right-looking;  trailing update C -= P P^T on v_mfma_f64_16x16x4_f64,
//                         128x128 tiles, K = NB  (the MFMA-bound kernel of the metric)
//   inner block 64:       left-looking inside the outer panel:
//        panel_update (MFMA GEMM, 64x64 tiles, K = 64*j)
//        diag_factor  (one workgroup, 64x64 block in LDS, one barrier per column, + the inverses of
//                      its four 16x16 diagonal tiles)
//        trsm_strip   (per wave 16 rows: X^T = L^-1 B^T by 16-blocks, a pure MFMA chain -- the f64
//                      accumulator layout is exactly the B-operand layout, so no data movement)
// With look-ahead the panel of step k+1 is factored on a second stream while the rest of the
// trailing update of step k runs.
#include <vector>

#include "gpmc_common.h"
#include "diag_mp.h"
#include "panel_fused.h"
#include "panel_pair.h"
#include "gemm4.h"
#include "tail.h"
// Superseded variants, kept for the measurements that led to the defaults: single-wave and
// register-blocked diagonal factors, the 5-wave panel kernel with in-block look-ahead, the persistent GEMM with C
// prefetch.  `make EXPERIMENTS=1` compiles them in; the default library does not carry them and their options
// (diag_impl 1/2, panel_impl 2, gemm_persistent) are rejected by gpmc_set_option.
#ifdef GPMC_EXPERIMENTS
#include "../../experiments/superseded/diag_wave.h"
#include "../../experiments/superseded/diag_rb.h"
#include "../../experiments/superseded/panel_fused2.h"
#include "../../experiments/superseded/gemm4p.h"
#endif

namespace gpmc {

Options g_opt;
thread_local char g_err[512] = {0};

bool g_ordered_handover = false;
unsigned long long* g_tail_dbg = nullptr;    // diagnostic only (gpmc_debug_tail_stamps): stamp buffer of the next launches
static size_t potrf_scratch_base(int np_) { return (size_t)(np_ / NB_IN + 1) * (1024 + 4096 + 4096) * sizeof(double); }
size_t potrf_scratch_bytes(int np_) { return potrf_scratch_base(np_) + TAIL_SYNC_BYTES + 64; }

// ======================================================================= zeroing of sync / info words
// A kernel, not hipMemsetAsync: in a replayed launch sequence that holds several memset nodes, HIP runtime 7.0.51831
// fills with a wrong value from the second replay on (the words came back as 0xa5a5a5a5: the tile-DAG kernel then saw a
// raised abort word and left at once -- the SDS proposal sequence with its two factorisations, round 3, DESIGN.md section 4;
// the log-marginal plan's sequences hold one memset node and never showed it).  One launch also clears the info word.
__global__ void __launch_bounds__(256) zero_words_kernel(unsigned* __restrict__ p, unsigned nquads, int* __restrict__ info) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < nquads) reinterpret_cast<u4*>(p)[i] = u4{0u, 0u, 0u, 0u};
    if (info != nullptr && i == 0) *info = 0;
}
static int zero_words(unsigned* p, size_t bytes, int* info, hipStream_t s) {       // p 16-byte aligned, bytes a multiple of 16
    const unsigned nquads = (unsigned)(bytes / 16);
    const unsigned grid = nquads ? (nquads + 255u) / 256u : 1u;
    hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(256), 0, s, p, nquads, info);
    GPMC_LAUNCH_CHECK();
    return 0;
}

// ======================================================================= diagonal block
// Factor the 64x64 block at Wd (lower triangle read) in LDS and write L back (strict upper = 0).
// Column step c applies a[i][j] -= a[i][c] a[j][c] / d_c directly from the unscaled column, so only
// one barrier per column is needed; columns are scaled by 1/sqrt(d_c) at the end (dpotf2 scales by
// the reciprocal as well).  Also emits inv(L_tt) for the four 16x16 diagonal tiles (for trsm_strip).
constexpr int DG_LD = NB_IN + 1;

template <typename T>
__global__ void __launch_bounds__(256)
diag_factor_kernel(T* __restrict__ Wd, int ld, T* __restrict__ dinv, int* __restrict__ info, int c0) {
    __shared__ T S[NB_IN][DG_LD];
    __shared__ T sd[NB_IN], sdi[NB_IN];
    __shared__ int sfail;
    const int tid = threadIdx.x;
    if (tid == 0) sfail = 0;
    for (int e = tid; e < NB_IN * NB_IN; e += 256) {
        const int i = e >> 6, j = e & 63;
        S[i][j] = (j <= i) ? Wd[(size_t)i * ld + j] : (T)0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    for (int c = 0; c < NB_IN; ++c) {
        const T dc = S[c][c];
        if (tid == 0 && !(dc > (T)0) && sfail == 0) sfail = c0 + c + 1;
        const T inv = (T)1 / dc;
        for (int i = c + 1 + ty; i < NB_IN; i += 16) {
            const T lic = S[i][c] * inv;
            for (int j = c + 1 + tx; j <= i; j += 16) S[i][j] -= lic * S[j][c];
        }
        __syncthreads();
    }
    if (tid < NB_IN) {
        const T s = sqrt(S[tid][tid]);
        sd[tid] = s;
        sdi[tid] = (T)1 / s;
    }
    __syncthreads();
    for (int e = tid; e < NB_IN * NB_IN; e += 256) {
        const int i = e >> 6, j = e & 63;
        T v = (T)0;
        if (j < i) v = S[i][j] * sdi[j];
        else if (j == i) v = sd[i];
        S[i][j] = v;
    }
    __syncthreads();
    for (int e = tid; e < NB_IN * NB_IN; e += 256) {
        const int i = e >> 6, j = e & 63;
        Wd[(size_t)i * ld + j] = S[i][j];
    }
    if (tid < NB_IN) {  // inverse of diagonal tile t, column jj, by forward substitution
        const int t = tid >> 4, jj = tid & 15, o = 16 * t;
        T x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            T s = (i == jj) ? (T)1 : (T)0;
#pragma unroll
            for (int k = 0; k < i; ++k) s -= S[o + i][o + k] * x[k];
            x[i] = s * sdi[o + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) dinv[t * 256 + i * 16 + jj] = x[i];
    }
    if (tid == 0 && sfail != 0) atomicCAS(info, 0, sfail);
}

// ======================================================================= TRSM strip
// Rows r0.. (64 per workgroup, 16 per wave) of the panel: X = B L^-T, computed transposed,
// Y = X^T (64 x 16 per wave) by 16-row blocks:  Y_jb = inv(L_jb,jb) (B^T_jb - sum_{p<jb} L_jb,p Y_p).
// An accumulator tile is used as the next MFMA's B operand unchanged (register s <-> k-step s).
constexpr int TS_LD = NB_IN + 2;

template <typename T>
__global__ void __launch_bounds__(256)
trsm_strip_kernel(T* __restrict__ Wp, int ld, const T* __restrict__ Ld, const T* __restrict__ dinv) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    __shared__ T Ls[NB_IN][TS_LD];
    __shared__ T Di[4][16][16 + 2];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB_IN * NB_IN; e += 256) {
        const int i = e >> 6, j = e & 63;
        Ls[i][j] = Ld[(size_t)i * ld + j];
    }
    for (int e = tid; e < 4 * 256; e += 256) Di[e >> 8][(e >> 4) & 15][e & 15] = dinv[e];
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    T* rowp = Wp + ((size_t)blockIdx.x * 64 + wv * 16 + r) * ld;  // this lane's matrix row

    acc_t Y[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        acc_t acc;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = rowp[16 * jb + M::crow(q, e)];
#pragma unroll
        for (int p = 0; p < jb; ++p) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const T a = -Ls[16 * jb + r][16 * p + M::crow(q, s)];
                acc = M::mma(a, Y[p][s], acc);
            }
        }
        acc_t out = acc_zero<T>();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const T a = Di[jb][r][M::crow(q, s)];
            out = M::mma(a, acc[s], out);
        }
        Y[jb] = out;
#pragma unroll
        for (int e = 0; e < 4; ++e) rowp[16 * jb + M::crow(q, e)] = out[e];
    }
}

// ======================================================================= GEMM / SYRK  (NT form)
//   C[i][j] = (beta ? Cin[i][j] : 0) + alpha * sum_k A[i][k] B[j][k]        row-major, k contiguous
// Tiles BM x BN = (WGM*TM*16) x (WGN*TN*16), 256 threads, K in steps of 128 bytes per row.
// tri != 0: only tiles with tj <= ti are launched (BM == BN), rows beyond the ntc-th tile row form a
// full rectangle (the riding right-hand sides).  EPI: add diag_add on the global diagonal and write
// identity where the global row index >= n_true (gpmc_syrk_diff).
constexpr int GM_ROWB = 144;  // LDS bytes per tile row: 128 data + 16 pad

template <typename T, int WGM, int WGN, int TM, int TN, bool EPI>
__global__ void __launch_bounds__(256, 2)
gemm_nt_kernel(const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, const T* Cin, int ldcin,
               T* C, int ldc, int K, int ntc, int tri, T alpha, int beta, int n_true, T diag_add, int ktri, T* C2) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    using frag_t = typename M::frag_t;
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    constexpr int EPL = M::EPL;
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr int NA = BM * 8 / 256, NBI = BN * 8 / 256;  // 16-byte chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                       // [2][BM][GM_ROWB]
    char* Bs = smem + 2 * BM * GM_ROWB;    // [2][BN][GM_ROWB]

    // ---- tile decode
    int ti, tj;
    {
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
    const int r = lane & 15, q = lane >> 4;

    acc_t acc[TM][TN];
    if (beta) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[mi][ni][e] = Cin[(size_t)(row0 + (wm * TM + mi) * 16 + M::crow(q, e)) * ldcin + col0 +
                                         (wn * TN + ni) * 16 + r];
    } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = acc_zero<T>();
    }

    const int kbeg = ktri ? row0 : 0;      // upper-trapezoidal operands: see gemm4.h
    const T* Ag = A + (size_t)row0 * lda + kbeg;
    const T* Bg = B + (size_t)col0 * ldb + kbeg;
    frag_t pa[NA], pb[NBI];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i;
            pa[i] = *reinterpret_cast<const frag_t*>(Ag + (size_t)(c >> 3) * lda + k0 + (c & 7) * EPL);
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int c = tid + 256 * i;
            pb[i] = *reinterpret_cast<const frag_t*>(Bg + (size_t)(c >> 3) * ldb + k0 + (c & 7) * EPL);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i;
            frag_t v = pa[i];
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] *= alpha;
            *reinterpret_cast<frag_t*>(As + ((size_t)buf * BM + (c >> 3)) * GM_ROWB + (c & 7) * 16) = v;
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int c = tid + 256 * i;
            *reinterpret_cast<frag_t*>(Bs + ((size_t)buf * BN + (c >> 3)) * GM_ROWB + (c & 7) * 16) = pb[i];
        }
    };

    const int nk = (K - kbeg) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const char* Ab = As + ((size_t)cur * BM + wm * TM * 16 + r) * GM_ROWB + q * 16;
        const char* Bb = Bs + ((size_t)cur * BN + wn * TN * 16 + r) * GM_ROWB + q * 16;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            frag_t af[TM], bf[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                af[mi] = *reinterpret_cast<const frag_t*>(Ab + mi * 16 * GM_ROWB + kc * 64);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                bf[ni] = *reinterpret_cast<const frag_t*>(Bb + ni * 16 * GM_ROWB + kc * 64);
#pragma unroll
            for (int e = 0; e < EPL; ++e)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = M::mma(af[mi][e], bf[ni][e], acc[mi][ni]);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gr = row0 + (wm * TM + mi) * 16 + M::crow(q, e);
                const int gc = col0 + (wn * TN + ni) * 16 + r;
                T v = acc[mi][ni][e];
                if (EPI) {
                    if (gr == gc) v += diag_add;
                    if (gr >= n_true || gc >= n_true) v = (gr == gc) ? (T)1 : (T)0;
                }
                C[(size_t)gr * ldc + gc] = v;
                if (EPI) if (C2) C2[(size_t)gr * ldc + gc] = v;
            }
}

template <typename T, int WGM, int WGN, int TM, int TN, bool EPI>
static int launch_gemm(const T* A, int lda, const T* B, int ldb, const T* Cin, int ldcin, T* C, int ldc, int mrows,
                       int ncols, int K, int tri, T alpha, int beta, int n_true, T diag_add, hipStream_t s,
                       int ktri = 0, int lds_pad = 0, int prio_cols = 0, unsigned* gflag = nullptr, T* C2 = nullptr) {
    constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    constexpr int BK = 128 / (int)sizeof(T);
    if (mrows <= 0 || ncols <= 0) return 0;
    if (mrows % BM || ncols % BN || K % BK || K <= 0)
        return fail(-1, "launch_gemm: shape %dx%dx%d not tileable by %dx%dx%d", mrows, ncols, K, BM, BN, BK);
    const int mt = mrows / BM, ntc = ncols / BN;
    int nblk;
    if (tri) {
        if (BM != BN || mt < ntc) return fail(-1, "launch_gemm: bad triangular shape");
        nblk = ntc * (ntc + 1) / 2 + (mt - ntc) * ntc;
    } else {
        nblk = mt * ntc;
    }
    const size_t smem = (size_t)2 * (BM + BN) * GM_ROWB;
    if constexpr (sizeof(T) == 8) {
        if (g_opt.gemm_impl == 1) {   // v_mfma_f64_4x4x4 slabs: same tile sizes, RG = 4 TM row-groups, CG = TN
#ifdef GPMC_EXPERIMENTS
            if (BM == 128 && beta && !ktri && g_opt.gemm_persistent && lds_pad == 0 && nblk >= 64 && !gflag) {
                // persistent pipelined kernel: one 512-thread workgroup per CU walks the tile list, next C tile
                // prefetched under the current K loop
                auto kp = gemm4p_nt_kernel<EPI>;
                static bool attrp_done = false;
                if (!attrp_done) {
                    GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kp),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attrp_done = true;
                }
                const int gridp = nblk < 256 ? nblk : 256;
                hipLaunchKernelGGL(kp, dim3(gridp), dim3(512), smem + 16384, s, A, lda, B, ldb, Cin, ldcin, C, ldc, K, ntc,
                                   tri, nblk, alpha, n_true, diag_add);
                GPMC_LAUNCH_CHECK();
                return 0;
            }
#endif
            if (BM == 128 && lds_pad > 0 && g_opt.gemm_8wave && !gflag) {
                // look-ahead phase: one workgroup per CU -> make it 8 waves (2 per SIMD), wave tile 32 x 64
                auto k8 = gemm4_nt_kernel<4, 2, 8, 4, EPI>;
                static bool attr8_done = false;
                if (!attr8_done) {
                    GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k8),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attr8_done = true;
                }
                hipLaunchKernelGGL(k8, dim3(nblk), dim3(512), smem + (size_t)lds_pad, s, A, lda, B, ldb, Cin, ldcin,
                                   C, ldc, K, ntc, tri, alpha, beta, n_true, diag_add, ktri, 0, (unsigned*)nullptr, C2);
                GPMC_LAUNCH_CHECK();
                return 0;
            }
            if constexpr (BM == 64) if (g_opt.gemm_small_bk32 && K % 32 == 0) {   // small tiles: BK = 32 (half the stages)
                auto ks = gemm4_nt_kernel<WGM, WGN, 4 * TM, TN, EPI, 4>;
                const size_t smems = (size_t)2 * (BM + BN) * (64 * 4 + 16);
                static bool attrs_done = false;
                if (!attrs_done) {
                    GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ks),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attrs_done = true;
                }
                hipLaunchKernelGGL(ks, dim3(nblk), dim3(256), smems, s, A, lda, B, ldb, Cin, ldcin, C, ldc, K, ntc, tri,
                                   alpha, beta, n_true, diag_add, ktri, 0, (unsigned*)nullptr, C2);
                GPMC_LAUNCH_CHECK();
                return 0;
            }
            auto k4 = gemm4_nt_kernel<WGM, WGN, 4 * TM, TN, EPI>;
            static bool attr4_done = false;
            if (!attr4_done) {
                GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k4),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr4_done = true;
            }
            const size_t smem4 = smem + ((BM == 128) ? (size_t)lds_pad : 0);
            hipLaunchKernelGGL(k4, dim3(nblk), dim3(256), smem4, s, A, lda, B, ldb, Cin, ldcin, C, ldc, K, ntc, tri,
                               alpha, beta, n_true, diag_add, ktri, gflag ? prio_cols : 0, gflag, C2);
            GPMC_LAUNCH_CHECK();
            return 0;
        }
    }
    auto kern = gemm_nt_kernel<T, WGM, WGN, TM, TN, EPI>;
    static bool attr_done = false;  // per template instantiation
    if (!attr_done) {
        GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), smem, s, A, lda, B, ldb, Cin, ldcin, C, ldc, K, ntc, tri, alpha,
                       beta, n_true, diag_add, ktri, C2);
    GPMC_LAUNCH_CHECK();
    return 0;
}

// ---- optional instrumentation of the dominant kernel (trailing SYRK update): HIP event pairs recorded on the
// launch stream around every launch, summed by gpmc_profile_read().  Used by bench.py for the roofline line.
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t used = 0;
    double flops = 0.0;
    long launches = 0;
};
static Prof g_prof;       // trailing-update GEMM launches
static Prof g_prof_dag;   // tile-DAG launches (panels and tail)
static int prof_mark2(hipStream_t s) {
    if (g_prof_dag.used == g_prof_dag.ev.size()) {
        hipEvent_t e;
        GPMC_HIP(hipEventCreate(&e));
        g_prof_dag.ev.push_back(e);
    }
    GPMC_HIP(hipEventRecord(g_prof_dag.ev[g_prof_dag.used++], s));
    return 0;
}
static int prof_mark(hipStream_t s) {
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t e;
        GPMC_HIP(hipEventCreate(&e));
        g_prof.ev.push_back(e);
    }
    GPMC_HIP(hipEventRecord(g_prof.ev[g_prof.used++], s));
    return 0;
}
int profile_begin() {
    g_prof.on = true;
    g_prof.used = 0;
    g_prof.flops = 0.0;
    g_prof.launches = 0;
    g_prof_dag.used = 0;
    g_prof_dag.flops = 0.0;
    g_prof_dag.launches = 0;
    return 0;
}
int profile_read(double* out) {   // out: [0] total ms in the trailing-update kernel, [1] launches, [2] algorithmic flops;
                                  // [3..5] the same for the tile-DAG kernel (events on the stream each launch went to)
    g_prof.on = false;
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        GPMC_HIP(hipEventSynchronize(g_prof.ev[i + 1]));
        float t = 0.f;
        GPMC_HIP(hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]));
        ms += t;
    }
    out[0] = ms;
    out[1] = (double)g_prof.launches;
    out[2] = g_prof.flops;
    double ms2 = 0.0;
    for (size_t i = 0; i + 1 < g_prof_dag.used; i += 2) {
        GPMC_HIP(hipEventSynchronize(g_prof_dag.ev[i + 1]));
        float t = 0.f;
        GPMC_HIP(hipEventElapsedTime(&t, g_prof_dag.ev[i], g_prof_dag.ev[i + 1]));
        ms2 += t;
    }
    out[3] = ms2;
    out[4] = (double)g_prof_dag.launches;
    out[5] = g_prof_dag.flops;
    return 0;
}

// C(mrows x ncols region) -= A A_j^T, big (128x128) or small (64x64) tiles by problem size
template <typename T>
static int trailing_update(T* W, int ld, int k0, int nbk, int mp, int np_, hipStream_t s, int col_lo, int col_hi,
                           bool latency_bound = false, int lds_pad = 0, int prio_cols = 0, unsigned* gflag = nullptr) {
    // update columns [col_lo, col_hi) of the trailing matrix (rows >= col_lo), K = nbk panel at k0
    const int rows = mp - col_lo;
    const int cols = col_hi - col_lo;
    const T* P = W + (size_t)col_lo * ld + k0;
    T* C = W + (size_t)col_lo * ld + col_lo;
    if (g_prof.on) {
        int rc = prof_mark(s);
        if (rc) return rc;
        // algorithmic flops: the lower triangle (incl. diagonal) of the cols x cols square part plus the
        // (np_ - col_hi) x cols block below it inside the matrix; riding right-hand-side rows are not counted
        const double sq = 0.5 * (double)cols * ((double)cols + 1.0);
        const double below = (double)(np_ - col_hi) * (double)cols;
        g_prof.flops += 2.0 * (double)nbk * (sq + below);
        g_prof.launches++;
    }
    int rc;
    if (rows > g_opt.small_tile_m && !latency_bound)
        rc = launch_gemm<T, 2, 2, 4, 4, false>(P, ld, P, ld, C, ld, C, ld, rows, cols, nbk, 1, (T)-1, 1, 0, (T)0, s, 0, lds_pad,
                                               prio_cols, gflag);
    else if (gflag)
        return fail(-1, "trailing_update: in-launch hand-over needs the big-tile GEMM");
    else
        rc = launch_gemm<T, 2, 2, 2, 2, false>(P, ld, P, ld, C, ld, C, ld, rows, cols, nbk, 1, (T)-1, 1, 0, (T)0, s);
    if (rc) return rc;
    if (g_prof.on) return prof_mark(s);
    return 0;
}

// factor the outer panel at k0 (nbk columns, rows k0..mp): inner left-looking loop of 64-column blocks
// what the last fused panel launch left parked in scratch, to be flushed into W by the next one (or by the final flush)
struct Parked {
    int d0 = -1, d1 = -1, l21 = -1;   // diagonal factors at columns d0, d1; L21 of the 128-column launch at l21
};

template <typename T>
static int factor_panel(T* W, int ld, int k0, int nbk, int mp, int* info_dev, T* dinv, double* lpark, hipStream_t s,
                        Parked& pk, double* lpark21, bool beside_gemm) {
#ifdef GPMC_EXPERIMENTS
    if (g_opt.panel_impl == 2) {   // fused kernel with in-block look-ahead: 5 waves, wave 4 = serial micro-panel factor
        static bool attr2_done = false;   // per T instantiation
        if (!attr2_done) {
            GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_block2_kernel<false, T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Panel2Shared)));
            attr2_done = true;
        }
        for (int c0 = k0; c0 < k0 + nbk; c0 += NB_IN) {
            hipLaunchKernelGGL((panel_block2_kernel<false, T>), dim3((mp - c0) / NB_IN), dim3(320), sizeof(Panel2Shared),
                               s, W, ld, k0, c0, info_dev, lpark, c0 - NB_IN, (unsigned long long*)nullptr);
            GPMC_LAUNCH_CHECK();
        }
        return 0;
    }
#endif
    if (g_opt.panel_impl == 1) {   // fused update + factor + TRSM, one launch per 64 columns (f64 arithmetic; W may be f32)
        static bool attr_done = false;   // per T instantiation
        if (!attr_done) {
            GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_block_kernel<false, T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_done = true;
        }
        static bool attrp_done = false;   // per T instantiation
        if (!attrp_done) {
            GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_pair_kernel<T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PairShared)));
            attrp_done = true;
        }
        for (int c0 = k0; c0 < k0 + nbk;) {
            // panel_pair = 1: only beside the trailing GEMM (look-ahead), where every panel LAUNCH pays a CU-slot wait of
            // ~40 us and one launch for 128 columns halves that; alone on the chip the pair launch takes exactly as long
            // as the two launches it replaces (43-45 vs 18.6 + 23.7 us).  2: everywhere.
            if ((g_opt.panel_pair >= 2 || (g_opt.panel_pair == 1 && beside_gemm)) && c0 == k0 && nbk >= 2 * NB_IN) {
                // the first 128 columns of the outer panel in one launch (panel_pair.h)
                hipLaunchKernelGGL((panel_pair_kernel<T>), dim3((mp - c0) / NB_IN), dim3(256), sizeof(PairShared), s, W, ld,
                                   c0, info_dev, lpark, lpark21, pk.d0, pk.d1, pk.l21, g_opt.panel_prio);
                GPMC_LAUNCH_CHECK();
                pk.d0 = c0 + NB_IN; pk.d1 = c0; pk.l21 = c0;
                c0 += 2 * NB_IN;
            } else {
                // beside the GEMM an LDS request above 90,112 B (= 163,840 - one 73,728 B GEMM workgroup) makes the
                // panel workgroup wait for an EMPTY CU instead of sharing one (panel_lds_pad, see gpmc_common.h)
                const size_t psm = sizeof(PanelShared) + (beside_gemm ? (size_t)g_opt.panel_lds_pad : 0);
                hipLaunchKernelGGL((panel_block_kernel<false, T>), dim3((mp - c0) / NB_IN), dim3(256), psm,
                                   s, W, ld, k0, c0, info_dev, lpark, pk.d0, (unsigned long long*)nullptr,
                                   g_opt.panel_prio, lpark21, pk.d1, pk.l21);
                GPMC_LAUNCH_CHECK();
                pk.d0 = c0; pk.d1 = -1; pk.l21 = -1;
                c0 += NB_IN;
            }
        }
        return 0;
    }
    for (int c0 = k0; c0 < k0 + nbk; c0 += NB_IN) {
        if (c0 > k0) {  // rows [c0, mp) x cols [c0, c0+64) -= L[rows, k0:c0] L[c0:c0+64, k0:c0]^T
            const T* Ap = W + (size_t)c0 * ld + k0;
            T* Cp = W + (size_t)c0 * ld + c0;
            int rc = launch_gemm<T, 2, 2, 2, 2, false>(Ap, ld, Ap, ld, Cp, ld, Cp, ld, mp - c0, NB_IN, c0 - k0, 0,
                                                       (T)-1, 1, 0, (T)0, s);
            if (rc) return rc;
        }
        T* Wd = W + (size_t)c0 * ld + c0;
        if (g_opt.diag_impl == 0 || (sizeof(T) != 8 && g_opt.diag_impl == 3))
            hipLaunchKernelGGL((diag_factor_kernel<T>), dim3(1), dim3(256), 0, s, Wd, ld, dinv, info_dev, c0);
#ifdef GPMC_EXPERIMENTS
        else if (g_opt.diag_impl == 1)
            hipLaunchKernelGGL((diag_factor_wave_kernel<T>), dim3(1), dim3(64), 0, s, Wd, ld, dinv, info_dev, c0);
        else if (g_opt.diag_impl == 2)
            hipLaunchKernelGGL((diag_factor_rb_kernel<T>), dim3(1), dim3(256), 0, s, Wd, ld, dinv, info_dev, c0);
#endif
        else if constexpr (sizeof(T) == 8)
            hipLaunchKernelGGL(diag_factor_mp_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<double*>(Wd), ld,
                               reinterpret_cast<double*>(dinv), info_dev, c0);
        else
            return fail(-1, "potrf: diag_impl %d is not available for this element type", g_opt.diag_impl);
        GPMC_LAUNCH_CHECK();
        const int below = mp - c0 - NB_IN;
        if (below > 0) {
            hipLaunchKernelGGL((trsm_strip_kernel<T>), dim3(below / 64), dim3(256), 0, s,
                               W + (size_t)(c0 + NB_IN) * ld + c0, ld, Wd, dinv);
            GPMC_LAUNCH_CHECK();
        }
        dinv += 4 * 256;  // each diagonal block keeps its own inverse tiles (no WAR hazard across streams)
    }
    return 0;
}

template <typename T>
int potrf(T* W, int mp, int np_, int ldw, int* info_dev, void* scratch, hipStream_t s, int ntri, Sched* sc, int rhs_rows) {
    GPMC_ARG(W != nullptr, 1);
    GPMC_ARG(np_ > 0 && np_ % GPMC_PAD == 0, 3);
    GPMC_ARG(mp >= np_ && mp % GPMC_PAD == 0, 2);
    GPMC_ARG(ldw >= np_ && ldw % (16 / (int)sizeof(T)) == 0, 4);
    GPMC_ARG(info_dev != nullptr, 5);
    GPMC_ARG(scratch != nullptr, 6);
    GPMC_ARG(reinterpret_cast<uintptr_t>(W) % 16 == 0, 1);
    GPMC_ARG(ntri >= 0 && ntri % GPMC_PAD == 0 && ntri <= mp - np_, 8);
    GPMC_ARG(sc == nullptr || sc->main == s, 9);
    GPMC_ARG(rhs_rows < 0 || rhs_rows <= mp - np_ - ntri, 10);
    // rhs_rows = meaningful rows of the riding block right below the matrix (-1: unknown, all of it).  The block is padded
    // to 128 rows; with at most 64 of them in use -- the log-marginal's g, the sampler's f - m: ONE row -- its second
    // 64-row half is zero before and after, and the tile-DAG launches leave it alone (2 of the ~90 tile rows of an
    // N = 8192 tail launch are riding rows: 2.4 of its 100 Gflop went into that half)
    const int dead_row64 = ((g_opt.tail_skip & 1) && rhs_rows >= 0 && rhs_rows <= NB_IN && mp - np_ - ntri >= 2 * NB_IN) ? np_ / NB_IN + 1 : -1;
    // ... and of the first half only the waves (16 rows each) that hold a right-hand side issue MFMAs
    const int ride_waves = ((g_opt.tail_skip & 2) && rhs_rows >= 1 && rhs_rows <= NB_IN && mp - np_ - ntri >= NB_IN) ? (rhs_rows + 15) / 16 : 4;
    // The last ntri riding rows are upper-trapezoidal on entry (riding row mp - ntri + i is zero left of column i, e.g.
    // a scaled identity, which leaves s L^-T behind).  Row i then stays zero in every column < i, so while the panel
    // ending at column kend is factored or applied only the first kend of those rows can hold anything: the launches
    // stop at row mp_at(kend) and the riding work drops from N^3 to N^3/3.
    auto mp_at = [&](int kend) { return ntri > 0 ? mp - ntri + (kend < ntri ? kend : ntri) : mp; };
    T* dinv = reinterpret_cast<T*>(scratch);
    double* lpark = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + (size_t)(np_ / NB_IN + 1) * 1024 * sizeof(double));
    double* lpark21 = lpark + (size_t)(np_ / NB_IN + 1) * 4096;
    const bool fused = (g_opt.panel_impl >= 1);
    Parked pk;
    auto final_flush = [&]() -> int {
        if (!fused) return 0;
        if (g_opt.panel_impl == 1 && pk.d0 < 0 && pk.d1 < 0 && pk.l21 < 0) return 0;   // nothing parked
        if (g_opt.panel_impl == 1) {   // whatever the last launch parked: one or two diagonal factors and L21
            hipLaunchKernelGGL((panel_flush3_kernel<T>), dim3(3), dim3(256), 0, s, W, ldw, lpark, lpark21, pk.d0, pk.d1, pk.l21);
        } else {   // panel_impl 2 (experiments build)
            hipLaunchKernelGGL((panel_flush_kernel<T>), dim3(1), dim3(256), 0, s, W, ldw, lpark, np_ - NB_IN);
        }
        GPMC_LAUNCH_CHECK();
        return 0;
    };
    // schedule: serial with NB = 128 for small problems; look-ahead (panel k+1 beside the rest of update k) for
    // np_ >= la_min_np, f64 and f32 alike, outer block by remaining columns (la_nb below)
    bool la = (g_opt.lookahead < 0) ? (np_ >= g_opt.la_min_np && fused) : (g_opt.lookahead != 0);
    int NB = g_opt.outer_nb;
    if (NB <= 0) NB = la ? 256 : ((sizeof(T) == 4 && np_ >= 4096) ? g_opt.f32_nb : 128);
    if (NB < 128 || NB % 128) NB = 256;
    // Look-ahead outer block by remaining columns (auto): the panel chain (serial, ~20 us per 64 columns alone) hides
    // under the update as long as the update is long enough, and a deeper update is a better GEMM (fewer C passes).
    // Measured (scripts/one_eval.py, ms/eval at N = 16384 / 24576 / 32768): NB 256: 33.8 / 100.5 / 226.6,
    // 384: 33.4 / 97.5 / 218.5, 512: 34.1 / 96.4 / 215.5.
    auto la_nb = [&](int rem) {
        if (g_opt.outer_nb > 0) return NB;
        if (rem < g_opt.la_nb128_m) return 128;
        if (sizeof(T) == 4) return rem >= 20480 ? 384 : 256;   // f32 (ms/eval at 8192 / 16384 / 24576): serial 5.82 / 22.1 / 60.3,
                                                                // look-ahead NB 256: 5.32 / 20.6 / 56.7, NB 384: 5.71 / 20.9 / 56.0
        return rem >= 20480 ? 512 : (rem >= 14336 ? 384 : 256);
    };
    // Under look-ahead a padded LDS request holds the big-tile GEMM to one workgroup per CU, so that retiring
    // workgroups free whole CUs for the panel kernel at a finer grain; that pays below np_ = 8192 only (6144: 3.95 vs
    // 4.03 ms) -- from 8192 on two GEMM workgroups per CU win (8192: 6.73 vs 6.79, 16384: 33.8 vs 36.2, 32768: 227 vs 253).
    const int la_pad = (np_ >= 8192) ? 0 : 16000;
    const int pad_serial = (g_opt.gemm_lds_pad < 0) ? 0 : g_opt.gemm_lds_pad;
    const int pad_la = (g_opt.gemm_lds_pad < 0) ? la_pad : g_opt.gemm_lds_pad;
    // *info is cleared on s in front of the first launch that can write it: together with the sync words where a fill launch
    // for those follows on s anyway (one launch instead of two), else by a launch of its own (flush_info)
    bool info_pending = !(sc && sc->info_cleared);
    auto flush_info = [&]() -> int {
        if (!info_pending) return 0;
        info_pending = false;
        return zero_words(nullptr, 0, info_dev, s);
    };

    // the last columns as one tile-DAG launch (tail.h): f64
    // (float storage: the kernel converts on load / store, its arithmetic is f64 as in the panel kernels; the f32 trailing
    // GEMM is twice as fast as the f64 one, so the extents where the launch pays are smaller: tail_m_f32, tail_whole_m_f32)
    const int tail_m_T = sizeof(T) == 8 ? ((ntri > 0 && g_opt.tail_m_tri > 0) ? g_opt.tail_m_tri : g_opt.tail_m) : g_opt.tail_m_f32;
    const int tail_whole_T = sizeof(T) == 8 ? ((ntri > 0 && g_opt.tail_whole_m_tri >= 0) ? g_opt.tail_whole_m_tri : g_opt.tail_whole_m) : g_opt.tail_whole_m_f32;
    const bool tail_ok = tail_m_T > 0 && fused && g_opt.panel_impl == 1;
    auto use_tail = [&](int k0) {
        return tail_ok && np_ - k0 <= ((k0 == 0 && tail_whole_T > tail_m_T) ? tail_whole_T : tail_m_T) &&
               (np_ - k0) / NB_IN <= TAIL_MAXT &&
               tail_sync_words((np_ - k0) / NB_IN, (mp - k0) / NB_IN) * sizeof(unsigned) <= TAIL_SYNC_BYTES;
    };
    unsigned* const dag_sync = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(scratch) + potrf_scratch_base(np_));
    // one tile-DAG launch on stream st: block columns [k0, k0 + 64 Tt) of rows k0 .. mp_rows, accumulating first over
    // the Hh finished block columns to their left.  sync_next = word offset of this launch's sync region; with
    // sync_zeroed the whole range was zeroed by ONE memset at the start of the factorisation (see tail.h), else the
    // region is zeroed right here.
    size_t sync_next = 0;
    bool sync_zeroed = false;
    auto dag_launch = [&](int k0, int Tt, int mp_rows, int Hh, hipStream_t st, int workers,
                          const unsigned* gflag = nullptr, int gmt = 0) -> int {
        const int Rt = (mp_rows - k0) / NB_IN;
        const size_t words = tail_sync_words(Tt, Rt);
        if (!sync_zeroed) sync_next = 0;
        if ((sync_next + words) * sizeof(unsigned) > TAIL_SYNC_BYTES) return fail(-1, "potrf: sync region overflow");
        unsigned* sync = dag_sync + sync_next;
        if (!sync_zeroed) {     // (sync_next is 0 here: an aligned start; the scratch reserve has 64 bytes of slack for the round-up)
            int* const info_too = (st == s && info_pending) ? info_dev : nullptr;
            if (info_too) info_pending = false;
            int rcz = zero_words(sync, (words * sizeof(unsigned) + 15) & ~(size_t)15, info_too, st);
            if (rcz) return rcz;
        }
        sync_next += words;
        static bool attr_done = false;   // per T instantiation
        if (!attr_done) {
            const void* fns[] = {reinterpret_cast<const void*>(tail_kernel<T, 0, false>),
                                 reinterpret_cast<const void*>(tail_kernel<T, 4, false>),
                                 sizeof(T) == 8 ? reinterpret_cast<const void*>(tail_kernel<double, 4, true>) : nullptr};
            for (const void* fn : fns)
                if (fn) GPMC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TailShared)));
            attr_done = true;
        }
        // grid = the Tt pairs + workers for the off-diagonal tiles: at most `workers` of them
        const int ntl = tail_tiles(Tt, Rt);
        const int grid = Tt + (ntl < workers ? ntl : workers);
        if (g_prof.on) {
            int rcp = prof_mark2(st);
            if (rcp) return rcp;
            // algorithmic flops of the launch: the H history columns applied to the trapezoid, then the trapezoid's own
            // factorisation -- m^3/3 for the m x m triangle + m^2 per row below it (riding rows not counted)
            const double m = 64.0 * Tt, below = (double)(np_ - k0) - m, K = 64.0 * Hh;
            g_prof_dag.flops += 2.0 * K * (0.5 * m * (m + 1.0) + below * m) + m * m * m / 3.0 + below * m * m;
            g_prof_dag.launches++;
        }
        const int tflags = (g_opt.tail_park ? 1 : 0) | (g_opt.tail_warm ? 4 : 0) | (g_opt.tail_roles_by_index ? 8 : 0);
#define GPMC_TAIL_LAUNCH(A16, PRE)                                                                                       \
        hipLaunchKernelGGL((tail_kernel<T, A16, PRE>), dim3(grid), dim3(256), sizeof(TailShared), st, W, ldw, k0, Tt, Rt, Hh, \
                           (mp_rows - ntri - k0) / NB_IN, sync, info_dev, g_tail_dbg, st == s ? 0 : g_opt.dag_wprio,          \
                           (unsigned)g_opt.tail_spin_limit, gflag, gmt, tflags,                                            \
                           (dead_row64 >= k0 / NB_IN && dead_row64 < k0 / NB_IN + Rt) ? dead_row64 - k0 / NB_IN : -1,       \
                           (ride_waves < 4 && np_ / NB_IN < k0 / NB_IN + Rt) ? (np_ - k0) / NB_IN : -1, ride_waves,          \
                           (g_opt.tail_xcd && !g_opt.shared_card && !g_ordered_handover && st == s && workers >= 256) ? 8 : 1)
        // variants (tail.h): the 4x4x4 accumulation (cross-checks only); the 16x16x4 accumulation; the same with the next
        // column's operands prefetched -- for launches whose time is their bulk, not their pivot chain
        const bool pre = g_opt.tail_acc16 && sizeof(T) == 8 && (st == s ? (g_opt.tail_prefetch_t > 0 && Tt >= g_opt.tail_prefetch_t)
                                                      : g_opt.dag_prefetch != 0);
        if (!g_opt.tail_acc16) GPMC_TAIL_LAUNCH(0, false);
        else if (pre) {      // (float storage: 8-byte loads, no gain measured -- f32 N = 8192 3.92 with, 3.86 without)
            if constexpr (sizeof(T) == 8) GPMC_TAIL_LAUNCH(4, true);
        }
        else GPMC_TAIL_LAUNCH(4, false);
#undef GPMC_TAIL_LAUNCH
        GPMC_LAUNCH_CHECK();
        if (g_prof.on) return prof_mark2(st);
        return 0;
    };
    // Worker workgroups of a tile-DAG launch that has the chip to itself (tail_workers > 0 overrides).  A launch whose time
    // is its pivot chain does best when every pair has a CU to itself -- a co-resident worker's MFMA stream slows the
    // pair's factor wave even with the parking (tail.h) -- i.e. pairs + workers <= 256 workgroups; a launch whose time is
    // its bulk wants every CU slot filled (2 per CU).  ms/eval over workers (f64, whole problem):
    // N = 2048: 128 0.575, 192 0.577, 256 0.606, 480 0.614;  3072: 128 0.859, 192 0.845, 256 0.894, 480 0.899;
    // 4096: 192 1.193, 256 1.176, 480 1.186;  5120: 256 1.70, 384 1.64, 480 1.56;  6144: 384 2.36, 480 2.15.
    // Decided by the launch's bulk time (its flops at the ~33 TF/s the workers reach) against its chain (~17.5 us per tile
    // column): the sizes above sit at 0.15 / 0.35 / 0.62 / 0.97 / 1.4.
    auto alone_workers = [&](int k0, int Tt, int mp_rows, int Hh) {
        if (g_opt.tail_workers > 0) return g_opt.tail_workers;
        const double m = 64.0 * Tt, below = (double)(mp_rows - k0) - m, K = 64.0 * Hh;
        const double flops = 2.0 * K * (0.5 * m * m + below * m) + m * m * m / 3.0 + below * m * m;
        const double ratio = (flops / 33e12) / (Tt * 17.5e-6);
        if (ratio < 0.42 && Tt <= 160) return 256 - Tt < 96 ? 96 : 256 - Tt;
        if (ratio < 0.85) return 256;
        return 480;
    };
    auto tail_from = [&](int k0) -> int {
        int rc = final_flush();     // whatever the last panel launch parked must be in W before tiles are read
        if (rc) return rc;
        pk = Parked();
        return dag_launch(k0, (np_ - k0) / NB_IN, mp, 0, s, alone_workers(k0, (np_ - k0) / NB_IN, mp, 0));
    };
    if (g_opt.lookahead < 0 && use_tail(0)) la = false;      // the whole problem is one tile-DAG launch
    // the caller's matrix builder (Sched::build_cols): everything now, unless the schedule starts with a tile-DAG panel on
    // the second stream -- then that panel's columns first and the rest beside its factorisation (below)
    const bool dag_la = la && np_ > NB && tail_ok && g_opt.la_dag && sizeof(T) == 8;
    const bool have_builder = sc && sc->build_cols;
    auto build_cols = [&](int c0, int c1, void* zp = nullptr, size_t zb = 0) -> int {
        if (!have_builder || c0 >= c1) return 0;
        return sc->build_cols(s, c0, c1, zp, zb);
    };
    if (!dag_la || !g_opt.kbuild_split) {
        // (a problem that is ONE tile-DAG launch: its sync words are cleared by the builder's launch, not by a fill launch)
        size_t zb = 0;
        if (have_builder && !la && use_tail(0))
            zb = (tail_sync_words(np_ / NB_IN, mp / NB_IN) * sizeof(unsigned) + 15) & ~(size_t)15;
        int rcb = build_cols(0, np_, zb ? dag_sync : nullptr, zb);
        if (rcb) return rcb;
        if (zb) sync_zeroed = true;
    }
    // look-ahead hands over to the tail once the tile-DAG launch can take the rest (else: once < la_min_m columns remain)
    auto hand_over = [&](int k1) { return tail_ok ? use_tail(k1) : (np_ - k1 < g_opt.la_min_m); };
    // serial schedule from column kstart on (the panel at kstart is factored first), outer block nbs
    auto serial_from = [&](int kstart, int nbs) -> int {
        for (int k0 = kstart; k0 < np_; k0 += nbs) {
            if (use_tail(k0)) return tail_from(k0);
            const int nbk = (np_ - k0 < nbs) ? np_ - k0 : nbs;
            int rc = flush_info();
            if (rc) return rc;
            rc = factor_panel<T>(W, ldw, k0, nbk, mp_at(k0 + nbk), info_dev, dinv + (size_t)(k0 / NB_IN) * 1024, lpark, s, pk, lpark21, false);
            if (rc) return rc;
            const int k1 = k0 + nbk;
            if (k1 < np_) {
                rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k1, np_, false, pad_serial);
                if (rc) return rc;
            }
        }
        return 0;
    };
    if (!la || np_ <= NB) {
        int rc = serial_from(0, NB);
        if (rc) return rc;
        return final_flush();
    }

    // ---- look-ahead: after panel k, update only the next panel's columns first (on s), then factor
    // panel k+1 on the high-priority panel stream while the remaining columns are updated on s.  The cross-stream
    // edges go through the schedule object (eager events, or cuts of the plan's recorded sequence: gpmc_common.h).
    if (!sc) {
        sc = eager_sched(s);
        if (!sc) return fail(-1, "potrf: cannot create the look-ahead stream");
    }
    hipStream_t ps = sc->panel;
    if (tail_ok && g_opt.la_dag && sizeof(T) == 8) {   // (float storage: the f32 GEMM outruns tile-DAG panels; 16384: 23.7 vs 20.3 ms)
        // ---- look-ahead with tile-DAG panels.  With c_k the panel starts:
        //   panel stream : D_0 | D_1 | D_2 | ...   D_{k+1} = ONE tile-DAG launch that factors panel k+1 after accumulating
        //                                          panel k's update of those columns as left-looking history (no separate
        //                                          next-panel update, no per-64-column launches waiting for CU slots)
        //   main stream  : G_0 | G_1 | G_2 | ...   G_k = columns [c_{k+2}, np_) -= panel k (big tiles), back to back
        // G_k waits for D_k; D_{k+1} waits for G_{k-1} (which brought its columns up to panel k-1).  The hand-over tail is
        // one more tile-DAG launch with panel k as history.
        // Outer block and panel workers by remaining columns (swept with the in-launch hand-over in place, ms/eval,
        // scripts/dag_sweep.sh): a panel launch no longer has to fit under ONE GEMM, so deeper panels (better GEMMs: 43 TF/s
        // at K = 256, 53 at 512) pay from the start, and the fewer CU slots the panel holds the more the GEMM keeps -- as
        // long as the panel chain still keeps up:  N = 8192: NB 256 4.77, 384 4.67, 512 4.70;  9216: 6.35 / 6.21 / 6.17;
        // 10240 NB 512: 128 workers 8.63, 160 8.21, 224 8.01, 256 8.09 (NB 256: 8.56);  12288 NB 512: 96 workers 15.07,
        // 128 13.69, 160 12.95, 192 13.44, 256 13.81 (NB 256: 13.95);  16384 NB 512: 96 31.5, 128 28.5, 160 29.1, 256 30.5
        // (NB 256: 30.6; NB 1024 / 224 workers 28.3).
        auto la_nb_dag = [&](int rem) {
            if (g_opt.outer_nb > 0) return NB;
            (void)rem;                       // by problem size: switching 512 -> 384 below 9216 remaining columns
                                             // measured worse than staying (10240: 8.31 vs 8.01, 12288: 13.43 vs 12.95)
            // 512 from N = 8192 on (with the 16x16x4 accumulation: NB 384 / 192 workers 4.36 ms, NB 512 with 192 / 224 /
            // 256 / 288 workers 4.31 / 4.31 / 4.28 / 4.29).  Round 2 held this back because two PROCESSES on one card ran
            // that schedule into the spin limit (14 of 24 runs): a gated panel launch of 8 + 256 workgroups spins on flags
            // of a GEMM that may not be resident yet, and two of them (528 workgroups) fill all 512 CU slots so that
            // neither GEMM can place a block.  That wait is now structurally absent when the card is shared (option
            // shared_card, below: stream-ordered hand-over), so the outer block no longer has to guard against it.
            return np_ >= 8192 ? 512 : 384;
        };
        auto panel_workers = [&](int nb, int rem) {
            if (g_opt.dag_workers > 0) return g_opt.dag_workers;
            (void)rem;   // by problem size too: more workers as the remaining columns shrink loses (12288: 13.54 vs 12.95)
            if (nb >= 512) return np_ >= 14336 ? 128 : (np_ >= 11264 ? 160 : (np_ >= 9216 ? 224 : 256));
            return nb / 2;
        };
        hipEvent_t pe, ge = nullptr;
        int nbk = la_nb_dag(np_);
        // (a narrower FIRST panel -- the only one that runs alone on the chip, its chain exposed -- measured +-0 at N = 8192,
        //  9216 and 12288 with widths 128 / 256, and mixed widths are not something the in-launch hand-over supports)
        if (nbk > np_) nbk = np_;
        // In-launch hand-over (dag_ahead): G_k -- the update of the columns beyond panel k+1 by panel k -- processes the
        // tiles of panel k+2's columns first and flags each one (gemm4.h), and D_{k+2} gates its tiles on those flags
        // instead of on G_k's end: the panel chain runs up to a step ahead of the GEMM chain, which then runs back to
        // back.  Words of G_k's flags: (tile columns of panel k+2) x (tile rows of its region), 0 when not used.
        auto next_nb = [&](int kk) { return (np_ - kk < la_nb_dag(np_ - kk)) ? np_ - kk : la_nb_dag(np_ - kk); };
        // Not when other processes use the card too (shared_card): the gated launch's workgroups hold CU slots while they
        // wait for a producer that may not have been dispatched, and with another process's launches competing for the
        // same slots that wait can outlast the spin limit.  The hand-over is then the event edge ge (stream order): no
        // workgroup of a panel launch ever waits for another kernel.
        auto handover_words = [&](int k1, int k2) -> size_t {
            if (sizeof(T) != 8 || !g_opt.dag_ahead || g_opt.shared_card || g_ordered_handover || !g_opt.dag_hist || g_opt.gemm_impl != 1 ||
                k2 >= np_ || hand_over(k2))
                return 0;
            const int rows = mp_at(k1) - k2;
            if (rows <= g_opt.small_tile_m || rows % 128 || (np_ - k2) % 128) return 0;
            return (size_t)(next_nb(k2) / 128) * (size_t)(rows / 128);
        };
        const unsigned* gf_prev = nullptr;     // hand-over flags the NEXT panel launch gates on (of the last GEMM launched)
        int gmt_prev = 0;
        {   // sync words of every launch of this schedule, zeroed once, here, on s before the fork
            size_t total = 0;
            for (int k0 = 0, nb = nbk;;) {
                total += tail_sync_words(nb / NB_IN, (mp - k0) / NB_IN);
                const int k1 = k0 + nb;
                if (k1 >= np_) break;
                if (hand_over(k1)) { total += tail_sync_words((np_ - k1) / NB_IN, (mp - k1) / NB_IN); break; }
                nb = (np_ - k1 < la_nb_dag(np_ - k1)) ? np_ - k1 : la_nb_dag(np_ - k1);
                total += handover_words(k1, k1 + nb);
                k0 = k1;
            }
            const bool fits = total * sizeof(unsigned) <= TAIL_SYNC_BYTES;
            const size_t zb = fits ? ((total * sizeof(unsigned) + 15) & ~(size_t)15) : 0;
            if (have_builder && g_opt.kbuild_split) {      // the first panel's columns, and the sync words with them
                int rcb = build_cols(0, nbk, zb ? dag_sync : nullptr, zb);
                if (rcb) return rcb;
            } else if (fits) {
                int* const info_too = info_pending ? info_dev : nullptr;
                info_pending = false;
                int rcz = zero_words(dag_sync, zb, info_too, s);
                if (rcz) return rcz;
            }
            sync_zeroed = fits;
            int rci = flush_info();
            if (rci) return rci;
        }
        {
            hipEvent_t ef;
            int rc = sc->edge_record(s, &ef);
            if (rc) return rc;
            rc = sc->edge_wait(ps, ef);
            if (rc) return rc;
            rc = dag_launch(0, nbk / NB_IN, mp, 0, ps, alone_workers(0, nbk / NB_IN, mp, 0));
            if (rc) return rc;
            rc = sc->edge_record(ps, &pe);
            if (rc) return rc;
            if (g_opt.kbuild_split && have_builder && nbk < np_) {      // the rest of the caller's matrix, on s, beside D_0
                rc = build_cols(nbk, np_);
                if (rc) return rc;
                // D_1 reads its own columns' start values: it is enqueued behind D_0 on ps and, through this edge, behind
                // the rest of the build on s (every later panel launch waits for a GEMM that follows the build on s)
                rc = sc->edge_record(s, &ge);
                if (rc) return rc;
            }
        }
        for (int k0 = 0;;) {
            const int k1 = k0 + nbk;
            int rc = 0;
            if (k1 >= np_) return sc->edge_wait(s, pe);
            if (hand_over(k1)) {
                rc = sc->edge_wait(s, pe);
                if (rc) return rc;
                return dag_launch(k1, (np_ - k1) / NB_IN, mp, nbk / NB_IN, s, alone_workers(k1, (np_ - k1) / NB_IN, mp, nbk / NB_IN));
            }
            const int nbn = (np_ - k1 < la_nb_dag(np_ - k1)) ? np_ - k1 : la_nb_dag(np_ - k1);
            const int k2 = k1 + nbn;
            if (ge && gf_prev == nullptr) rc = sc->edge_wait(ps, ge);
            if (rc) return rc;
            int hist = nbk / NB_IN;
            if (!g_opt.dag_hist) {   // experiment: panel k's update of the next panel's columns by the GEMM kernel instead
                rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, ps, k1, k2, true, pad_la);
                if (rc) return rc;
                hist = 0;
            }
            rc = dag_launch(k1, nbn / NB_IN, mp, hist, ps, panel_workers(nbn, np_ - k1), gf_prev, gmt_prev);
            if (rc) return rc;
            hipEvent_t pe_next;
            rc = sc->edge_record(ps, &pe_next);
            if (rc) return rc;
            rc = sc->edge_wait(s, pe);
            if (rc) return rc;
            gf_prev = nullptr;
            if (k2 < np_) {
                const size_t hw = sync_zeroed ? handover_words(k1, k2) : 0;
                unsigned* gf = nullptr;
                if (hw) {
                    gf = dag_sync + sync_next;
                    sync_next += hw;
                    gf_prev = gf;
                    gmt_prev = (mp_at(k1) - k2) / 128;
                }
                rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k2, np_, false, pad_la,
                                        hw ? next_nb(k2) / 128 : 0, gf);
                if (rc) return rc;
                rc = sc->edge_record(s, &ge);
                if (rc) return rc;
            }
            pe = pe_next;
            k0 = k1;
            nbk = nbn;
        }
    }
    if (g_opt.la_depth == 2) {
        // ---- depth-2 look-ahead.  With c_k = k NB:
        //   panel stream : P_0 | U_1 P_1 | U_2 P_2 | ...   U_{k+1} = columns [c_{k+1}, c_{k+2}) -= panel k (small tiles),
        //                                                  P_{k+1} = factor those columns
        //   main stream  : G_0 | G_1 | G_2 | ...           G_k = columns [c_{k+2}, np_) -= panel k (big tiles)
        // G_k waits for P_k; U_{k+1} waits for G_{k-1} (which brought its columns up to panel k-1).  The main stream
        // then carries nothing but the big GEMMs back to back: the latency-bound next-panel update and one
        // cross-stream hand-over per step leave its critical path.
        hipEvent_t pe_prev, ge_prev = nullptr;
        {
            hipEvent_t ef;   // fork: the panel stream starts behind whatever built the matrix on s
            int rc = flush_info();
            if (rc) return rc;
            rc = sc->edge_record(s, &ef);
            if (rc) return rc;
            rc = sc->edge_wait(ps, ef);
            if (rc) return rc;
            const int nbk = (np_ < NB) ? np_ : NB;
            rc = factor_panel<T>(W, ldw, 0, nbk, mp_at(nbk), info_dev, dinv, lpark, ps, pk, lpark21, false);
            if (rc) return rc;
            rc = sc->edge_record(ps, &pe_prev);
            if (rc) return rc;
        }
        for (int k0 = 0; k0 < np_; k0 += NB) {
            const int nbk = (np_ - k0 < NB) ? np_ - k0 : NB;
            const int k1 = k0 + nbk;
            if (k1 >= np_) {   // panel k was the last one: join
                int rc = sc->edge_wait(s, pe_prev);
                if (rc) return rc;
                break;
            }
            if (hand_over(k1)) {
                // tail: join, bring the whole remaining matrix up to panel k, continue serially with NB = 128
                int rc = sc->edge_wait(s, pe_prev);
                if (rc) return rc;
                rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k1, np_, false, pad_serial);
                if (rc) return rc;
                rc = serial_from(k1, 128);
                if (rc) return rc;
                return final_flush();
            }
            const int nbn = (np_ - k1 < NB) ? np_ - k1 : NB;
            const int k2 = k1 + nbn;
            // panel stream: U_{k+1}, P_{k+1}
            int rc = 0;
            if (ge_prev) rc = sc->edge_wait(ps, ge_prev);
            if (rc) return rc;
            rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, ps, k1, k2, true, pad_la);
            if (rc) return rc;
            rc = factor_panel<T>(W, ldw, k1, nbn, mp_at(k2), info_dev, dinv + (size_t)(k1 / NB_IN) * 1024, lpark, ps, pk, lpark21, true);
            if (rc) return rc;
            hipEvent_t pe_next;
            rc = sc->edge_record(ps, &pe_next);
            if (rc) return rc;
            // main stream: G_k
            rc = sc->edge_wait(s, pe_prev);
            if (rc) return rc;
            if (k2 < np_) {
                rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k2, np_, false, pad_la);
                if (rc) return rc;
                rc = sc->edge_record(s, &ge_prev);
                if (rc) return rc;
            }
            pe_prev = pe_next;
        }
        return final_flush();
    }
    // panel 0 on the main stream
    int nbk = la_nb(np_);
    if (nbk > np_) nbk = np_;
    {
        int rc = flush_info();
        if (rc) return rc;
        rc = factor_panel<T>(W, ldw, 0, nbk, mp_at(nbk), info_dev, dinv, lpark, s, pk, lpark21, false);
        if (rc) return rc;
    }
    for (int k0 = 0, nb_next = 0; k0 < np_; k0 += nbk, nbk = nb_next) {
        const int k1 = k0 + nbk;
        if (k1 >= np_) break;
        if (hand_over(k1)) {
            // tail: the GEMM no longer hides the (contention-stretched) panel chain -> serial, NB = 128, 2 WGs/CU
            int rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k1, np_, false, pad_serial);
            if (rc) return rc;
            rc = serial_from(k1, 128);
            if (rc) return rc;
            return final_flush();
        }
        const int nbn = (np_ - k1 < la_nb(np_ - k1)) ? np_ - k1 : la_nb(np_ - k1);  // next panel width
        nb_next = nbn;
        // (1) next panel's columns, on s
        int rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k1, k1 + nbn, true, pad_la);
        if (rc) return rc;
        hipEvent_t e1, e2;
        rc = sc->edge_record(s, &e1);
        if (rc) return rc;
        // la_order = 1 issues (3) before (2) (enqueue-order experiment, measured slower)
        if (g_opt.la_order == 1 && k1 + nbn < np_) {
            rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k1 + nbn, np_, false, pad_la);
            if (rc) return rc;
        }
        // (2) factor next panel on ps
        rc = sc->edge_wait(ps, e1);
        if (rc) return rc;
        rc = factor_panel<T>(W, ldw, k1, nbn, mp_at(k1 + nbn), info_dev, dinv + (size_t)(k1 / NB_IN) * 1024, lpark, ps, pk, lpark21, true);
        if (rc) return rc;
        rc = sc->edge_record(ps, &e2);
        if (rc) return rc;
        // (3) the rest of the trailing matrix, on s, concurrently with (2)
        if (g_opt.la_order != 1 && k1 + nbn < np_) {
            rc = trailing_update<T>(W, ldw, k0, nbk, mp_at(k1), np_, s, k1 + nbn, np_, false, pad_la);
            if (rc) return rc;
        }
        rc = sc->edge_wait(s, e2);
        if (rc) return rc;
    }
    return final_flush();
}

template <typename T>
int syrk_diff(const T* D, int ldd, const T* P, int kp, int ldp, T diag_add, T* C, int n, int np_, int ldc,
              hipStream_t s) {
    GPMC_ARG(D && P && C, 1);
    GPMC_ARG(kp > 0 && kp % GPMC_PAD == 0 && ldp >= kp, 3);
    GPMC_ARG(n > 0 && n <= np_ && np_ % GPMC_PAD == 0, 8);
    GPMC_ARG(ldc >= np_ && ldd >= np_, 10);
    if (np_ > g_opt.small_tile_m)
        return launch_gemm<T, 2, 2, 4, 4, true>(P, ldp, P, ldp, D, ldd, C, ldc, np_, np_, kp, 1, (T)-1, 1, n, diag_add, s);
    return launch_gemm<T, 2, 2, 2, 2, true>(P, ldp, P, ldp, D, ldd, C, ldc, np_, np_, kp, 1, (T)-1, 1, n, diag_add, s);
}

// C = diag_add I - P P^T for an upper-trapezoidal P (row i zero left of column i; n x np_ inside an np_ x np_ padded
// square whose padding becomes the identity): the products of a lower tile start at its first row -> N^3/3 flops.
template <typename T>
int syrk_tri(const T* P, int ldp, T diag_add, T* C, int n, int np_, int ldc, hipStream_t s, T* C2) {
    GPMC_ARG(P && C, 1);
    GPMC_ARG(n > 0 && n <= np_ && np_ % GPMC_PAD == 0, 5);
    GPMC_ARG(ldc >= np_ && ldp >= np_, 7);
    if (np_ > g_opt.small_tile_m)
        return launch_gemm<T, 2, 2, 4, 4, true>(P, ldp, P, ldp, nullptr, 0, C, ldc, np_, np_, np_, 1, (T)-1, 0, n, diag_add, s, 1, 0, 0, nullptr, C2);
    return launch_gemm<T, 2, 2, 2, 2, true>(P, ldp, P, ldp, nullptr, 0, C, ldc, np_, np_, np_, 1, (T)-1, 0, n, diag_add, s, 1, 0, 0, nullptr, C2);
}

// rows (nrows x np_, ldr) <- rows L^-T with L the lower triangle of W (f64): 64-column blocks, MFMA TRSM on the
// block then a rank-64 GEMM update of the columns to its right.
int trsm_rows_f64(const double* W, int np_, int ldw, double* R, int nrows, int ldr, hipStream_t s) {
    GPMC_ARG(W != nullptr && np_ > 0 && np_ % NB_IN == 0 && ldw >= np_, 1);
    GPMC_ARG(R != nullptr && nrows > 0 && nrows % NB_IN == 0 && ldr >= np_ && ldr % 2 == 0, 4);
    static bool attr_done = false;
    if (!attr_done) {
        GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_block_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TrsmShared)));
        attr_done = true;
    }
    for (int c0 = 0; c0 < np_; c0 += NB_IN) {
        hipLaunchKernelGGL(trsm_block_kernel, dim3(nrows / NB_IN), dim3(256), sizeof(TrsmShared), s,
                           W + (size_t)c0 * ldw + c0, ldw, R + c0, ldr);
        GPMC_LAUNCH_CHECK();
        const int rest = np_ - c0 - NB_IN;
        if (rest > 0) {
            int rc = launch_gemm<double, 2, 2, 2, 2, false>(R + c0, ldr, W + (size_t)(c0 + NB_IN) * ldw + c0, ldw,
                                                            R + c0 + NB_IN, ldr, R + c0 + NB_IN, ldr, nrows, rest,
                                                            NB_IN, 0, -1.0, 1, 0, 0.0, s);
            if (rc) return rc;
        }
    }
    return 0;
}

// C (M x N, ldc) = A (M x K, lda) * B (N x K, ldb)^T, f64, M and N multiples of 64, K multiple of 16
int gemm_nt_f64(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int M, int N, int K,
                hipStream_t s) {
    GPMC_ARG(A && B && C, 1);
    GPMC_ARG(M > 0 && N > 0 && K > 0 && M % 64 == 0 && N % 64 == 0 && K % 16 == 0, 7);
    GPMC_ARG(lda >= K && ldb >= K && ldc >= N && lda % 2 == 0 && ldb % 2 == 0, 2);
    return launch_gemm<double, 2, 2, 2, 2, false>(A, lda, B, ldb, C, ldc, C, ldc, M, N, K, 0, 1.0, 0, 0, 0.0, s);
}

// diagnostic: run the DBG instantiation of the panel kernel once on (W, k0, c0) and return 8 s_memtime stamps
int debug_panel_stamps(double* W, int ld, int mp, int k0, int c0, int* info_dev, double* lpark,
                       unsigned long long* dbg_dev, hipStream_t s) {
#ifdef GPMC_EXPERIMENTS
    if (g_opt.panel_impl == 2) {
        GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_block2_kernel<true, double>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Panel2Shared)));
        hipLaunchKernelGGL((panel_block2_kernel<true, double>), dim3((mp - c0) / NB_IN), dim3(320), sizeof(Panel2Shared),
                           s, W, ld, k0, c0, info_dev, lpark, -1, dbg_dev);
        GPMC_LAUNCH_CHECK();
        return 0;
    }
#endif
    GPMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_block_kernel<true, double>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PanelShared)));
    hipLaunchKernelGGL((panel_block_kernel<true, double>), dim3((mp - c0) / NB_IN), dim3(256), sizeof(PanelShared), s, W,
                       ld, k0, c0, info_dev, lpark, -1, dbg_dev, 0, (const double*)nullptr, -1, -1);
    GPMC_LAUNCH_CHECK();
    return 0;
}

int revt_lower_f64(const double* L, int n, int ld, double* Lt, int ldo, hipStream_t s) {
    GPMC_ARG(L != nullptr && n > 0 && n % 32 == 0 && ld >= n, 1);
    GPMC_ARG(Lt != nullptr && ldo >= n && Lt != L, 4);
    hipLaunchKernelGGL(revt_lower_kernel, dim3(n / 32, n / 32), dim3(256), 0, s, L, n, ld, Lt, ldo);
    GPMC_LAUNCH_CHECK();
    return 0;
}

template int potrf<double>(double*, int, int, int, int*, void*, hipStream_t, int, Sched*, int);
template int potrf<float>(float*, int, int, int, int*, void*, hipStream_t, int, Sched*, int);
template int syrk_tri<double>(const double*, int, double, double*, int, int, int, hipStream_t, double*);
template int syrk_tri<float>(const float*, int, float, float*, int, int, int, hipStream_t, float*);
template int syrk_diff<double>(const double*, int, const double*, int, int, double, double*, int, int, int, hipStream_t);
template int syrk_diff<float>(const float*, int, const float*, int, int, float, float*, int, int, int, hipStream_t);

}  // namespace gpmc
