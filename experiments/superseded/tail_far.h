// tail_far.h -- NOT COMPILED.  Round 3's "GEMM-shaped far worker" for the tile-DAG kernel (csrc/tail.h), kept with its
// measurements because the review of round 2 asked for exactly this design and it lost.
//
// What it is: tiles far below the diagonal worked on four at a time -- one workgroup per 128 x 128 super-tile, one wave per
// 64 x 64 tile (the trailing GEMM's wave tile on v_mfma_f64_4x4x4: 20 operand reads per 128 MFMAs instead of 17 per 16,
// half the L2 traffic per flop), operands staged 16 k at a time through a double-buffered LDS image, the two tile columns
// finished one after the other (solve, publish, one more K-loop column, solve, publish), tickets ordered block by block
// so that a ticket only waits for lower tickets.  It was wired in as tail_kernel<..., FAR = true> with the options
// tail_far / tail_near, as a non-inlined device function (inlined, its 128 accumulators and the pair's code did not fit
// 256 registers: 42 spills), with every argument made scalar on entry (arguments of a non-inlined function arrive in
// VGPRs; buffer descriptors built from them were wrapped in waterfall loops until then).
//
// What was measured (MI355X, one process, ms per log-marginal evaluation, scripts/optsweep.py):
//   * bits: identical to the 64 x 64 path for every tail_far / tail_near combination tried
//     (test_tile_dag_schedule_knobs_do_not_change_the_bits at 4200 and 7300 with 9 far variants) -- the body is correct;
//   * time: N = 6144 (one launch) 2.14 without, 3.05 with (near = 8), 2.86 / 2.56 / 2.41 / 2.29 with near = 24 / 40 / 56 / 72;
//     N = 8192 4.27 without, 5.49 with, 5.21 / 4.89 / 4.64 / 4.58: monotone in the number of super-tiles, never a gain;
//   * s_memtime stamps of 990 super-tiles at N = 6144 (scripts/far_stamps.py at that commit): a K-loop column takes
//     27 K cycles = 12.9 us per super-tile with waits excluded (16.4 K would be the MFMA time of one wave alone on its
//     SIMD), 29 % of the K loop is spent looking at or waiting for flags (56 % for the early blocks), the two halves take
//     92 K cycles.  A super-tile therefore advances barely faster than the pivot chain (17 us per column): a worker that
//     starts late cannot catch up, and the last super-tiles' K loops (80 columns = 1 ms each, unsplittable) end the launch.
//     As four 64 x 64 tickets the same work is done by four workgroups in parallel, 3-5 us per column;
//   * PMC (rocprofv3, scripts/sq_counters.sh, N = 6144): MFMA busy 35 % with, 51 % without; wave time parked at
//     s_waitcnt / barriers 56 % vs 37 %.  And the premise was wrong: SQ_VALU_MFMA_BUSY_CYCLES is the same for both builds
//     (2579 M per launch) -- a v_mfma_f64_16x16x4 holds the pipe for 64 cycles per 2048 flop exactly as four 4x4x4 do; its
//     96-cycle issue interval only limits ONE wave issuing back to back (the 49 TF/s of profiles/r01_mfma_peak.txt), and the
//     tile workers run two waves per SIMD, which fill each other's gaps.  The 64 x 64 worker is not at its instruction's
//     ceiling; it is held by operand round trips and flag waits, which a coarser work unit makes worse.
// Kept: the LDS chunk swizzle of the operand tiles (tail.h, tail_swz), found on the way.
#if 0
// ---------------------------------------------------------------------------------------------------------------
// Far super-tiles (round 3).  PMC of round 2's tail launch (profiles/r03_sq_counters_before.txt): the MFMA pipe is busy
// 56 % of the launch with v_mfma_f64_16x16x4, an instruction that cannot be issued more densely than 2/3 (64 busy
// cycles per 96): the 64 x 64 worker sits at 83 % of ITS instruction's ceiling, and every column step of a tile re-reads
// 64 KB of operands from L2 for 0.5 Mflop (7.25 GB fetched per tail launch against 0.16 GB of matrix).  The full-rate
// instruction, v_mfma_f64_4x4x4, needs its operands reused from registers -- a GEMM-shaped wave tile -- so tiles far
// below the diagonal are worked on four at a time: a SUPER-TILE = the 2 x 2 tiles (2I, 2I+1) x (2J, 2J+1), one
// workgroup, one wave per tile (64 x 64 accumulators in the wave's registers, the trailing GEMM's wave tile: per 8 k
// 16 + 4 operand reads for 128 MFMAs where the 64 x 64 worker makes 17 for 16), operands staged 16 k at a time through
// a double-buffered LDS image of the two operand row blocks (2 x 128 rows x 144 B, exactly the bytes of Ld + Ls), half
// the L2 traffic per flop.  Every tile still sums its updates column by column, 8 k at a time, even k then odd k in one
// instruction each -- the same order as the 64 x 64 paths, so the bits do not depend on who works on a tile.
//   K loop      : columns -H .. 2J-1, all four tiles;
//   first half  : wait L(2J,2J); the two waves of tile column 2J solve theirs; both tiles are published;
//   column 2J   : one more K-loop column for the two waves of tile column 2J+1 (A operand: the pair tile (2J+1,2J),
//                 B operand: the two tiles just published, re-read);
//   second half : wait L(2J+1,2J+1); solve; publish.
// Which tiles: column block J (tile columns 2J, 2J+1 both inside the launch) hands the tile rows from 2J + 2 + near up
// to the last even row pair to super-tiles; the `near` rows under the diagonal stay 64 x 64 tiles (the chain needs them
// within a few column periods), as do an odd last row and everything in launches with trapezoidal riding rows.
// Ticket order: block by block -- near tiles of column 2J, near tiles of column 2J+1, super-tiles of block J -- so a
// ticket only ever waits for pairs and for LOWER tickets (operands of block J's tiles lie in earlier columns = earlier
// blocks, tile (i,2J+1) needs (i,2J) of the same block and earlier in it, a super-tile its own block's pair tile).
__host__ __device__ inline int tail_far_lo(int J, int near) { return 2 * J + 2 + near; }
__host__ __device__ inline int tail_far_hi(int R) { return R & ~1; }
__host__ __device__ inline bool tail_far_block(int J, int T, int R, int near) {
    return near >= 0 && 2 * J + 1 < T && tail_far_lo(J, near) < tail_far_hi(R);
}
__host__ __device__ inline int tail_far_count(int J, int T, int R, int near) {
    return tail_far_block(J, T, R, near) ? (tail_far_hi(R) - tail_far_lo(J, near)) / 2 : 0;
}
__host__ __device__ inline int tail_near_count(int l, int T, int R, int near) {
    const int fr = tail_first_row(l, T);
    if (!tail_far_block(l >> 1, T, R, near)) return R - fr > 0 ? R - fr : 0;
    const int lo = tail_far_lo(l >> 1, near);
    return (lo - fr > 0 ? lo - fr : 0) + (R - tail_far_hi(R));
}
// tickets of a launch (near = -1: no super-tiles, the count of tail_tiles)
__host__ __device__ inline int tail_tickets(int T, int R, int near) {
    int n = 0;
    for (int l = 0; l < T; ++l) n += tail_near_count(l, T, R, near);
    for (int J = 0; 2 * J + 1 < T; ++J) n += tail_far_count(J, T, R, near);
    return n;
}
// ticket q -> (far, ti, tl): far = false: the 64 x 64 tile (ti, tl); far = true: super-tile, ti = I, tl = J
__host__ __device__ inline void tail_ticket_decode(int q, int T, int R, int near, bool* far, int* ti, int* tl) {
    for (int J = 0;; ++J) {
        for (int e = 0; e < 2; ++e) {
            const int l = 2 * J + e;
            if (l >= T) continue;
            const int n = tail_near_count(l, T, R, near);
            if (q < n) {
                const int fr = tail_first_row(l, T);
                int row = fr + q;
                if (tail_far_block(J, T, R, near)) {
                    const int lo = tail_far_lo(J, near), c1 = lo - fr > 0 ? lo - fr : 0;
                    if (q >= c1) row = tail_far_hi(R) + (q - c1);
                }
                *far = false; *ti = row; *tl = l;
                return;
            }
            q -= n;
        }
        const int nf = tail_far_count(J, T, R, near);
        if (q < nf) { *far = true; *ti = tail_far_lo(J, near) / 2 + q; *tl = J; return; }
        q -= nf;
        if (2 * J >= T) { *far = false; *ti = R; *tl = T; return; }      // (not reached for q < tail_tickets)
    }
}

constexpr int TAIL_DBG_FAR_BASE = 16 * 128, TAIL_DBG_FAR_MAX = 8192;      // diagnostic stamp buffer: pairs, then 8 words per ticket
constexpr int FAR_ROWB = 144;      // LDS bytes per operand row and 16-k stage: 128 data + 16 pad (the trailing GEMM's image)
constexpr int FAR_SLD = 66;        // row stride (doubles) of the transposing tile stage: 132 dwords = 4 banks per row, so the
                                   // 8-byte accesses of lanes (c = row, i = column) hit 64 different banks
static_assert(2 * 128 * FAR_ROWB * 2 <= 2 * 64 * MP_LD * (int)sizeof(double), "operand stages overlay Ld + Ls");
static_assert(64 * FAR_SLD <= 64 * MP_LD, "tile stage overlays Ld or Ls");

template <typename FP, typename AP>
__device__ inline bool tail_wait4(const FP* f1, const FP* f2, const FP* f3, const FP* f4,
                                  AP* abortw, TailShared& sh, int tid, unsigned spin_limit, int* info, int why) {
    if (tid == 0) {
        int ok = 1;
        for (unsigned spins = 0;; ++spins) {
            if (tail_ld32(f1) != 0u && tail_ld32(f2) != 0u && tail_ld32(f3) != 0u && tail_ld32(f4) != 0u) break;
            if ((spins & 255u) == 255u && tail_ld32(abortw) != 0u) { ok = 0; break; }
            if (spins >= spin_limit) {
                atomicCAS(info, 0, why);
                __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        sh.ok = ok;
    }
    __syncthreads();
    const bool ok = __builtin_amdgcn_readfirstlane(sh.ok) != 0;
    __syncthreads();
    return ok;
}

// One super-tile (I, J) of the launch whose tile (0,0) is W0; all 256 threads; false = the launch is being aborted.
// sync words as in tail_kernel (flag[col][row] at sync + 4 + col * R + row).
__device__ __noinline__ bool tail_far_tile(double* __restrict__ W0_, int ld_, int FI_, int FJ_, int H_, int R_, int gb_,
                                           unsigned* __restrict__ sync_, int* __restrict__ info_, unsigned spin_limit_,
                                           const unsigned* __restrict__ gflag_, int gmt_, unsigned* busy_,
                                           unsigned long long* __restrict__ stamps_) {
    // Not inlined: with 128 accumulators of its own this body and the pair's do not fit one 256-register allocation
    // (42 spilled registers inlined; as a function it has the 256 to itself and its callee-saved registers go to the stack
    // once per super-tile).
    // LDS is reached through the kernel's dynamic segment, global words through address_space(1) pointers, tiles through
    // buffer descriptors -- nothing generic, no flat_ instruction.
    // Every argument is uniform over the workgroup but arrives in vector registers: back into scalars first.
    double* const W0 = tail_uniform_ptr(W0_);
    const int ld = __builtin_amdgcn_readfirstlane(ld_), FI = __builtin_amdgcn_readfirstlane(FI_),
              FJ = __builtin_amdgcn_readfirstlane(FJ_), H = __builtin_amdgcn_readfirstlane(H_),
              R = __builtin_amdgcn_readfirstlane(R_), gb = __builtin_amdgcn_readfirstlane(gb_),
              gmt = __builtin_amdgcn_readfirstlane(gmt_);
    const unsigned spin_limit = (unsigned)__builtin_amdgcn_readfirstlane((int)spin_limit_);
    int* const info = tail_uniform_ptr(info_);
    TailShared& sh = *reinterpret_cast<TailShared*>(tail_smem);
    tail_gu32* const sync = (tail_gu32*)tail_uniform_ptr(sync_);
    const tail_gu32* const gflag = (const tail_gu32*)tail_uniform_ptr(gflag_);
    (void)busy_;      // (no stepping aside here: a blocking look at the CU word per column cost more than it saved)
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane >> 4, c = lane & 15, ia = lane & 3;
    const int h = w & 1, r2 = w >> 1;              // this wave's tile: (2 FI + r2, 2 FJ + h)
    // diagnostic (gpmc_debug_far_stamps): [0] start [1] tiles loaded [2] K loop done [3] end [4] cycles of the K loop spent
    // looking at / waiting for flags [5] its columns [6] cycles waiting in the two halves [7] (CU id)
    unsigned long long* const stamps = tail_uniform_ptr(stamps_);
    unsigned long long waited = 0;
#define FAR_STAMP(k) do { if (stamps != nullptr && tid == 0) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define FAR_T0() const unsigned long long t0__ = stamps != nullptr ? __builtin_amdgcn_s_memtime() : 0ull
#define FAR_T1() do { if (stamps != nullptr) waited += __builtin_amdgcn_s_memtime() - t0__; } while (0)
    FAR_STAMP(0);
    tail_gu32* const abortw = sync;
    auto flag = [&](int col, int row) { return sync + 4 + col * R + row; };
    auto tile = [&](int ti, int tl) { return W0 + (size_t)(64 * ti) * ld + 64 * tl; };
    char* const As = reinterpret_cast<char*>(&sh.Ld[0][0]);        // [2][128][FAR_ROWB]: rows of tiles (2FJ, col), (2FJ+1, col)
    char* const Bs = As + 2 * 128 * FAR_ROWB;                      // [2][128][FAR_ROWB]: rows of tiles (2FI, col), (2FI+1, col)
    double (*S0)[FAR_SLD] = reinterpret_cast<double (*)[FAR_SLD]>(&sh.Ld[0][0]);     // transposing stages, one tile each
    double (*S1)[FAR_SLD] = reinterpret_cast<double (*)[FAR_SLD]>(&sh.Ls[0][0]);
    const int ra = 2 * FJ, rb = 2 * FI;

    // ---- the four tiles' start values may still be under way (look-ahead panel launch beside the trailing GEMM): the
    // super-tile is exactly one of its 128 x 128 tiles
    if (gflag != nullptr && FI < gmt) {
        if (!tail_wait2(gflag + FJ * gmt + FI, gflag + FJ * gmt + FI, abortw, sh, tid, spin_limit, info,
                        tail_timeout_code(TAIL_WAIT_GEMM, gb + ra), true))
            return false;
    }
    // ---- tiles into registers, transposed: lane (i, c) of acc[rg][cg] = element [16 cg + c][4 rg + i] of the wave's tile
    double acc[16][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const __amdgpu_buffer_rsrc_t src = tail_rsrc(tile(rb + rr, ra + hh));
            double (*S)[FAR_SLD] = rr ? S1 : S0;
            const int voff = (int)(((tid >> 5) * ld + (tid & 31) * 2) * sizeof(double)), step = (int)(8 * ld * sizeof(double));
            tail_u4 v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b128(src, voff + t * step, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                *reinterpret_cast<d2_t*>(&S[8 * t + (tid >> 5)][(tid & 31) * 2]) = __builtin_bit_cast(d2_t, v[t]);
        }
        __syncthreads();
        if (h == hh) {
            double (*S)[FAR_SLD] = r2 ? S1 : S0;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) acc[rg][cg] = S[16 * cg + c][4 * rg + i];
        }
        __syncthreads();
    }

    FAR_STAMP(1);
    // ---- K loop: operand rows staged 16 k at a time, double-buffered, the next stage's loads in flight under this
    // stage's MFMAs.  A column of this launch is loaded only once its four operand tiles are published.
    tail_u4 pa[4], pb[4];
    const int ld_voff = (int)(((tid >> 3) * ld + (tid & 7) * 2) * sizeof(double));     // lane's 16 bytes in rows 0 .. 31 of a stage
    const int ld_step = (int)(32 * ld * sizeof(double));
    auto issue = [&](int col, int ch) {        // rows (tid >> 3) + 32 t: t = 0, 1 of the first tile, 2, 3 of the second
        const __amdgpu_buffer_rsrc_t a0 = tail_rsrc(tile(ra, col)), a1 = tail_rsrc(tile(ra + 1, col));
        const __amdgpu_buffer_rsrc_t b0 = tail_rsrc(tile(rb, col)), b1 = tail_rsrc(tile(rb + 1, col));
        const int so = ch * 128;
        pa[0] = __builtin_amdgcn_raw_buffer_load_b128(a0, ld_voff, so, 16);
        pa[1] = __builtin_amdgcn_raw_buffer_load_b128(a0, ld_voff + ld_step, so, 16);
        pa[2] = __builtin_amdgcn_raw_buffer_load_b128(a1, ld_voff, so, 16);
        pa[3] = __builtin_amdgcn_raw_buffer_load_b128(a1, ld_voff + ld_step, so, 16);
        pb[0] = __builtin_amdgcn_raw_buffer_load_b128(b0, ld_voff, so, 16);
        pb[1] = __builtin_amdgcn_raw_buffer_load_b128(b0, ld_voff + ld_step, so, 16);
        pb[2] = __builtin_amdgcn_raw_buffer_load_b128(b1, ld_voff, so, 16);
        pb[3] = __builtin_amdgcn_raw_buffer_load_b128(b1, ld_voff + ld_step, so, 16);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o = (buf * 128 + (tid >> 3) + 32 * t) * FAR_ROWB + (tid & 7) * 16;
            *reinterpret_cast<tail_u4*>(As + o) = pa[t];
            *reinterpret_cast<tail_u4*>(Bs + o) = pb[t];
        }
    };
    // MFMAs of one stage; only_h: -1 all waves, else only the waves of tile column 2FJ + only_h
    auto compute = [&](int buf, int only_h) {
        if (only_h >= 0 && h != only_h) return;
        const char* Ab = As + (buf * 128 + 64 * h + ia) * FAR_ROWB + i * 16;
        const char* Bb = Bs + (buf * 128 + 64 * r2 + c) * FAR_ROWB + i * 16;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            d2_t bf[4];
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) bf[cg] = -*reinterpret_cast<const d2_t*>(Bb + cg * 16 * FAR_ROWB + kc * 64);
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const d2_t af = *reinterpret_cast<const d2_t*>(Ab + rg * 4 * FAR_ROWB + kc * 64);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[0], bf[cg][0], acc[rg][cg], 0, 0, 0);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[1], bf[cg][1], acc[rg][cg], 0, 0, 0);
            }
        }
    };
    int ready_upto = 0;
    // all threads; true if column cc's four operand tiles are published (looks at up to 64 columns at once)
    auto peek = [&](int cc, int cend) {
        if (cc < ready_upto) return true;
        FAR_T0();
        if (tid < 64) {
            const int x = cc + tid;
            const bool ok = x < cend && tail_ld32(flag(x, ra)) != 0u && tail_ld32(flag(x, ra + 1)) != 0u &&
                            tail_ld32(flag(x, rb)) != 0u && tail_ld32(flag(x, rb + 1)) != 0u;
            const unsigned long long m = __ballot(ok);
            if (tid == 0) sh.ok = (int)__builtin_ctzll(~m);
        }
        __syncthreads();
        const int n = __builtin_amdgcn_readfirstlane(sh.ok);
        __syncthreads();
        ready_upto = cc + n;
        FAR_T1();
        return n > 0;
    };
    auto wait_col = [&](int cc) {
        FAR_T0();
        const bool ok = tail_wait4(flag(cc, ra), flag(cc, ra + 1), flag(cc, rb), flag(cc, rb + 1), abortw, sh, tid, spin_limit,
                                   info, tail_timeout_code(TAIL_WAIT_TILE, gb + cc));
        ready_upto = cc + 1;
        FAR_T1();
        return ok;
    };
    // columns [c0, cend) through the pipeline
    auto kloop = [&](int c0, int cend, int only_h) -> bool {
        if (c0 >= cend) return true;
        if (c0 >= 0 && !peek(c0, cend) && !wait_col(c0)) return false;
        issue(c0, 0);
        stash(0);
        __syncthreads();
        int col = c0, ch = 0, cur = 0;
        for (;;) {
            int ncol = col, nch = ch + 1;
            if (nch == 4) { nch = 0; ++ncol; }
            const bool more = ncol < cend;
            bool pre = more;
            if (more && nch == 0 && ncol >= 0) pre = peek(ncol, cend);
            if (pre) issue(ncol, nch);
            compute(cur, only_h);
            if (pre) stash(cur ^ 1);
            __syncthreads();
            if (!more) break;
            if (!pre) {      // caught up with the producers: wait, then refill the pipeline
                if (!wait_col(ncol)) return false;
                issue(ncol, 0);
                stash(cur ^ 1);
                __syncthreads();
            }
            col = ncol; ch = nch; cur ^= 1;
        }
        return true;
    };
    if (!kloop(H > 0 ? -H : 0, ra, -1)) return false;
    FAR_STAMP(2);
    if (stamps != nullptr && tid == 0) { stamps[4] = waited; stamps[5] = (unsigned long long)(ra + (H > 0 ? H : 0)); stamps[7] = __smid(); }
    waited = 0;

    // ---- the two halves: tile column 2FJ + hh becomes final for the waves h == hh
#pragma unroll 1
    for (int hh = 0; hh < 2; ++hh) {
        const int cl = ra + hh;
        if (hh == 1) {
            // column 2FJ as one more K-loop column for tile column 2FJ+1: A = the pair tile (2FJ+1, 2FJ) (first tile of the
            // stage: the diagonal factor, unused), B = the two tiles published below
            if (!kloop(ra, ra + 1, 1)) return false;
        }
        {
            FAR_T0();
            const bool okf = tail_wait2(flag(cl, cl), flag(cl, cl), abortw, sh, tid, spin_limit, info, tail_timeout_code(TAIL_WAIT_FACTOR, gb + cl));
            FAR_T1();
            if (!okf) return false;
        }
        tail_tile_in(sh.Ld, tile(cl, cl), ld, tid);
        __syncthreads();
        double (*Lout)[MP_LD] = sh.Ld;
        if (tid < 16) {
            const int o = 4 * tid;
            const double l10 = Lout[o + 1][o];
            const double l20 = Lout[o + 2][o], l21 = Lout[o + 2][o + 1];
            const double l30 = Lout[o + 3][o], l31 = Lout[o + 3][o + 1], l32 = Lout[o + 3][o + 2];
            const double i00 = 1.0 / Lout[o][o], i11 = 1.0 / Lout[o + 1][o + 1], i22 = 1.0 / Lout[o + 2][o + 2],
                         i33 = 1.0 / Lout[o + 3][o + 3];
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
        if (h == hh) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const double av = sh.Linv[p][ia][i];
                double Yn[4];
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const double Y = __builtin_amdgcn_mfma_f64_4x4x4f64(av, acc[p][cg], 0.0, 0, 0, 0);
                    acc[p][cg] = Y;
                    Yn[cg] = -Y;
                }
#pragma unroll
                for (int g = p + 1; g < 16; ++g) {
                    if ((g & 3) == 0) asm volatile("" ::: "memory");
                    const double al = Lout[4 * g + ia][4 * p + i];
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
                        acc[g][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(al, Yn[cg], acc[g][cg], 0, 0, 0);
                }
            }
        }
        __syncthreads();          // Lout (Ld) is free again
        if (h == hh) {
            double (*S)[FAR_SLD] = r2 ? S1 : S0;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) S[16 * cg + c][4 * rg + i] = acc[rg][cg];
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const __amdgpu_buffer_rsrc_t r = tail_rsrc(tile(rb + rr, cl));
            double (*S)[FAR_SLD] = rr ? S1 : S0;
            const int voff = (int)(((tid >> 5) * ld + (tid & 31) * 2) * sizeof(double)), step = (int)(8 * ld * sizeof(double));
#pragma unroll
            for (int t = 0; t < 8; ++t)
                TailRaw<double>::store(r, voff + t * step, 0, *reinterpret_cast<const d2_t*>(&S[8 * t + (tid >> 5)][(tid & 31) * 2]));
        }
        tail_drain();
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(flag(cl, rb), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag(cl, rb + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    FAR_STAMP(3);
    if (stamps != nullptr && tid == 0) stamps[6] = waited;
#undef FAR_STAMP
#undef FAR_T0
#undef FAR_T1
    return true;
}

#endif
