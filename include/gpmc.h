/* gpmc.h -- C ABI of libgpmc.so: MI355X (gfx950) GP-likelihood core.
 *
 * Drop-in boundary for the hot path of t-kychen/GaussianProcess-MCMC (the reference is pure
 * Python; what it binds at this boundary is the `kcGP` object protocol and NumPy/LAPACK calls,
 * cited per entry point as <reference file>:<line>).  Plain pointers and sizes only; no torch
 * types.  All pointers named *_dev / W / K / X ... are DEVICE pointers owned by the caller; the
 * library allocates no persistent memory, launches everything on the caller's stream and never
 * synchronises unless the function says so.  Matrices are ROW-MAJOR (C order, as NumPy hands them
 * to the reference) with an explicit leading dimension in elements.
 *
 * Return value: 0 = ok, <0 = bad argument (-k: k-th argument), >0 = hipError_t of a failed call.
 * gpmc_last_error() returns a thread-local message for the last non-zero return.
 *
 * "Workspace matrix" W (factorisation input/output): mp x np_ elements, ld >= np_, with
 *   np_ = gpmc_pad(n)   columns (multiple of GPMC_PAD), rows 0..np_-1 hold the symmetric matrix's
 *                       LOWER triangle, identity on the padding,
 *   rows np_..mp-1      optional extra rows ("right-hand sides", transposed): after
 *                       gpmc_potrf_* they hold rows * L^-T, i.e. row r becomes (L^-1 b_r)^T.
 *   mp multiple of GPMC_PAD.
 */
#ifndef GPMC_H
#define GPMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMC_VERSION 100 /* 0.1.0 */
#define GPMC_PAD 128     /* row/column padding granule of workspace matrices */

typedef void* gpmc_stream_t; /* hipStream_t */

/* kernel families (covK.<Kernel>) */
enum { GPMC_KERNEL_RBF = 0, GPMC_KERNEL_MATERN52_ARD = 1 };
/* gpmc_kbuild flags */
enum {
    GPMC_KB_FULL = 0,       /* write every entry                                           */
    GPMC_KB_LOWER = 1,      /* write only tiles that touch the lower triangle (train mode) */
    GPMC_KB_PAD_IDENTITY = 2 /* entries with i>=n or j>=nz: 1 on the diagonal, 0 elsewhere  */
};

int gpmc_version(void);
const char* gpmc_last_error(void);
/* padded size of an n x n problem */
int gpmc_pad(int n);
/* bytes of device scratch gpmc_potrf_* needs for a workspace of np_ columns */
size_t gpmc_potrf_scratch_bytes(int np_);

/* ---- a1: covariance build ------------------------------------------------------------------
 * Replaces kcGP.covK.RBF(log_ell, log_sf).getCovMatrix(x=, z=, mode='train'|'cross')
 * (call sites kcMCMC/sliceSample.py:104-105,136-137,255,263; demoRegression.py:124-125) and the
 * dense-S diagonal of aux_var_model (sliceSample.py:183-190) fused onto the diagonal.
 *   K[i][j] = k(x_i, z_j) (+ diag_add if i == j and x is z)
 * hyp_dev (device, T): [0] = sf^2, [1] = diag_add, [2..] = ell (1 value for RBF, d for ARD).
 * X: n x d (ldx), Z: nz x d (ldz) or NULL (=X).  K: rows_out x cols_out (ldk) with
 * rows_out >= n, cols_out >= nz: entries outside n x nz are padding (see flags).
 */
int gpmc_kbuild_f64(int kernel, const double* X, int n, int d, int ldx, const double* Z, int nz, int ldz,
                    const double* hyp_dev, double* K, int rows_out, int cols_out, int ldk, int flags,
                    gpmc_stream_t stream);
int gpmc_kbuild_f32(int kernel, const float* X, int n, int d, int ldx, const float* Z, int nz, int ldz,
                    const float* hyp_dev, float* K, int rows_out, int cols_out, int ldk, int flags,
                    gpmc_stream_t stream);

/* ---- a3/a4/a7/a9/a16: blocked Cholesky of a workspace matrix, with riding right-hand sides ---
 * Replaces kcGP.tools.jitchol (sliceSample.py:196,205,257; LAPACK dpotrf underneath) and, through
 * the extra rows, np.linalg.solve(L, .) (sliceSample.py:108,197) / tools.solve_chol's forward half
 * (sliceSample.py:258).  In place: on return the lower triangle of W[0:np_,0:np_] is L, rows
 * np_..mp-1 are rows * L^-T.  The strict upper triangle inside each 64x64 diagonal block is zeroed;
 * other upper-triangle entries are not referenced and not written.
 * info_dev (device int): 0, or 1-based index of the first non-positive pivot (LAPACK convention);
 * the factorisation keeps running on garbage after a failure -- check info before using W.
 * scratch: gpmc_potrf_scratch_bytes(np_) bytes of device memory.
 */
int gpmc_potrf_f64(double* W, int mp, int np_, int ldw, int* info_dev, void* scratch, gpmc_stream_t stream);
int gpmc_potrf_f32(float* W, int mp, int np_, int ldw, int* info_dev, void* scratch, gpmc_stream_t stream);
/* As gpmc_potrf, for riding rows whose LAST ntri rows (ntri a multiple of GPMC_PAD, <= mp - np_) are
 * upper-trapezoidal on entry: riding row mp - ntri + i is zero left of column i.  A scaled identity s I there
 * returns s L^-T, i.e. the explicit inverse factor, for N^3/3 flops instead of the N^3 of a dense right-hand side:
 * those rows stay zero left of their diagonal, so every launch stops at the last row that can hold anything.
 * This is how the SDS block gets R = S - S (K+S)^-1 S (== K - V^T V, sliceSample.py:197-198) -- see gpmc_syrk_tri. */
/* gpmc_potrf / gpmc_potrf_tri with one more fact from the caller: only the first rhs_rows rows of the riding block right
 * below the matrix (rows np_ .. np_ + 127 of a block padded to GPMC_PAD) are in use, the rest of that block is zero.  With
 * rhs_rows <= 64 the one-launch factorisation skips the block's second 64-row half (it stays zero).  ntri as in
 * gpmc_potrf_tri (0: none); rhs_rows = -1: unknown (= gpmc_potrf / gpmc_potrf_tri). */
int gpmc_potrf_rhs_f64(double* W, int mp, int np_, int ldw, int ntri, int rhs_rows, int* info_dev, void* scratch,
                       gpmc_stream_t stream);
int gpmc_potrf_rhs_f32(float* W, int mp, int np_, int ldw, int ntri, int rhs_rows, int* info_dev, void* scratch,
                       gpmc_stream_t stream);
/* small companions of the SDS proposal sequence (gpmc/sds.py): a vector copy as a kernel, and the proposal's results --
 * out4 (gpmc_logdet_quad), the second factorisation's info word and a vector (f') -- written straight into pinned,
 * device-mapped host memory: host[0..3] = out4, host[4] = *info2, host[8 .. 8 + n) = vec (as doubles). */
int gpmc_veccopy_f64(double* dst, const double* src, int n, gpmc_stream_t stream);
int gpmc_veccopy_f32(float* dst, const float* src, int n, gpmc_stream_t stream);
/* between R = -P P^T (gpmc_syrk_tri2) and chol(R + 1e-11 I) of an SDS proposal, one launch: diag(R) += *shift_dev (the s of
 * s I - P P^T), diag_out = that diagonal, diag(R) += jitter (the 1e-11 of sliceSample.py:205), row_out = f - m (:206) */
int gpmc_shift_diag_sub_f64(double* R, int ldr, int n, const double* shift_dev, double jitter, double* diag_out, const double* f,
                            const double* m, double* row_out, gpmc_stream_t stream);
int gpmc_shift_diag_sub_f32(float* R, int ldr, int n, const float* shift_dev, float jitter, float* diag_out, const float* f,
                            const float* m, float* row_out, gpmc_stream_t stream);
int gpmc_publish_host_f64(const double* out4, const int* info2, const double* vec, int n, double* host_mapped, gpmc_stream_t stream);
int gpmc_publish_host_f32(const double* out4, const int* info2, const float* vec, int n, double* host_mapped, gpmc_stream_t stream);
/* A caller's launch sequence -- library launches and the caller's own kernels, all on `stream` (the look-ahead's second
 * stream joins through its event edges) -- recorded once and replayed: begin (stream capture starts; not the NULL
 * stream), enqueue the sequence, end (instantiates), launch any number of times on any stream, retire.  The sequence
 * must hold no host value that changes between replays and allocate nothing.  Retired like the plans' execs (parked on
 * HIP runtimes < 7.2, see gpmc_graveyard_stats). */
typedef struct gpmc_seq gpmc_seq;
int gpmc_seq_begin(gpmc_seq** out, gpmc_stream_t stream);
int gpmc_seq_end(gpmc_seq* seq);
int gpmc_seq_launch(gpmc_seq* seq, gpmc_stream_t stream);
int gpmc_seq_retire(gpmc_seq* seq);
/* re-arm that trapezoidal block for the next factorisation: rows (np_ x np_, ld) <- diagonal *diag_dev (device scalar),
 * zero right of it; the strict lower part -- never written by gpmc_potrf_tri -- is left alone (half the bytes of a memset,
 * and no host value in the launch: it can be part of a recorded sequence).  np_ a multiple of 64. */
int gpmc_reset_tri_f64(double* rows, int np_, int ld, const double* diag_dev, gpmc_stream_t stream);
int gpmc_reset_tri_f32(float* rows, int np_, int ld, const float* diag_dev, gpmc_stream_t stream);
int gpmc_potrf_tri_f64(double* W, int mp, int np_, int ldw, int ntri, int* info_dev, void* scratch,
                       gpmc_stream_t stream);
int gpmc_potrf_tri_f32(float* W, int mp, int np_, int ldw, int ntri, int* info_dev, void* scratch,
                       gpmc_stream_t stream);

/* ---- a11: log-marginal reduction --------------------------------------------------------------
 * Replaces the curG/propG expression of sliceSample.py:122,147 given the factor and alpha = L^-1 g:
 *   out_dev[0] = -(0.5*sum_j alpha_j^2 + sum_i log L_ii + n/2 log 2pi),  [1] = sum log L_ii,
 *   out_dev[2] = sum alpha^2, [3] = (double) *info_dev.
 * diag of W[0:n,0:n]; alpha = n elements starting at alpha_dev.  Accumulates in f64 for both types.
 */
int gpmc_logdet_quad_f64(const double* W, int n, int ldw, const double* alpha_dev, const int* info_dev,
                         double* out_dev, gpmc_stream_t stream);
int gpmc_logdet_quad_f32(const float* W, int n, int ldw, const float* alpha_dev, const int* info_dev,
                         double* out_dev, gpmc_stream_t stream);

/* ---- packing between caller matrices and workspace matrices ------------------------------------
 * pack:   W[0:np_,0:np_] <- lower triangle of A (n x n, lda) + jitter on the diagonal, identity
 *         padding (the reference hands jitchol a dense symmetric ndarray; inputs are never mutated,
 *         sliceSample.py:196,205).  rows: W[np_+r, 0:np_] <- B[r, 0:n] (nrhs rows, ldb), 0 padding,
 *         for r < mp - np_ (rows beyond nrhs are zeroed).
 * unpack: L (n x n, ldl) <- lower triangle of W, strict upper = 0 (jitchol's return convention,
 *         callers rely on it at sliceSample.py:197).  rows: B[r, 0:n] <- W[np_+r, 0:n].
 */
int gpmc_pack_lower_f64(const double* A, int n, int lda, double jitter, double* W, int np_, int ldw,
                        gpmc_stream_t stream);
int gpmc_pack_rows_f64(const double* B, int nrhs, int n, int ldb, double* W, int mp, int np_, int ldw,
                       gpmc_stream_t stream);
int gpmc_unpack_lower_f64(const double* W, int np_, int ldw, double* L, int n, int ldl, gpmc_stream_t stream);
int gpmc_unpack_rows_f64(const double* W, int np_, int ldw, double* B, int nrhs, int n, int ldb,
                         gpmc_stream_t stream);
int gpmc_pack_lower_f32(const float* A, int n, int lda, float jitter, float* W, int np_, int ldw,
                        gpmc_stream_t stream);
int gpmc_pack_rows_f32(const float* B, int nrhs, int n, int ldb, float* W, int mp, int np_, int ldw,
                       gpmc_stream_t stream);
int gpmc_unpack_lower_f32(const float* W, int np_, int ldw, float* L, int n, int ldl, gpmc_stream_t stream);
int gpmc_unpack_rows_f32(const float* W, int np_, int ldw, float* B, int nrhs, int n, int ldb,
                         gpmc_stream_t stream);

/* ---- a5: SYRK-form difference  C = D - P P^T  (lower tiles) --------------------------------------
 * Replaces R_theta = K - np.dot(V.T, V) (sliceSample.py:198) with P = V^T = K L^-T (the riding rows
 * of gpmc_potrf).  C (np_ x np_, ldc) receives D - P P^T on tiles touching the lower triangle plus
 * diag_add on the diagonal (R + 1e-11 I of sliceSample.py:205), identity on the padding (i >= n).
 * D may alias C.  P: np_ x kp (ldp), kp multiple of GPMC_PAD.
 */
int gpmc_syrk_diff_f64(const double* D, int ldd, const double* P, int kp, int ldp, double diag_add, double* C,
                       int n, int np_, int ldc, gpmc_stream_t stream);
int gpmc_syrk_diff_f32(const float* D, int ldd, const float* P, int kp, int ldp, float diag_add, float* C,
                       int n, int np_, int ldc, gpmc_stream_t stream);

/* ---- a4+a5 in inverse form:  C = diag_add I - P P^T  for an upper-trapezoidal P ---------------------------
 * P (np_ x np_, ldp; row i zero left of column i) = S^(1/2)-scaled inverse factor left behind by gpmc_potrf_tri
 * with riding rows s I:  P P^T = s^2 (K+S)^-1, so diag_add = s (+ 1e-11) gives R_theta (+ 1e-11 I) of
 * sliceSample.py:198,205 through the identity K - K (K+S)^-1 K = S - S (K+S)^-1 S.  The k loop of a lower tile starts
 * at the tile's first row: N^3/3 flops.  Identity on the padding (i >= n).
 */
int gpmc_syrk_tri_f64(const double* P, int ldp, double diag_add, double* C, int n, int np_, int ldc,
                      gpmc_stream_t stream);
int gpmc_syrk_tri_f32(const float* P, int ldp, float diag_add, float* C, int n, int np_, int ldc,
                      gpmc_stream_t stream);
/* the same product written to two matrices of the same leading dimension (C2 != C): chol(R + 1e-11 I) works in place and
 * fails at rounding-level eigenvalues about one proposal in three at N = 8192 (jitchol's schedule then adds jitter,
 * sliceSample.py:205 -> pyGPs jitchol); the second copy lets the retry restore R instead of forming it again. */
int gpmc_syrk_tri2_f64(const double* P, int ldp, double diag_add, double* C, double* C2, int n, int np_, int ldc,
                       gpmc_stream_t stream);
int gpmc_syrk_tri2_f32(const float* P, int ldp, float diag_add, float* C, float* C2, int n, int np_, int ldc,
                       gpmc_stream_t stream);

/* ---- a6/a10: symmetric / triangular matrix-vector products on workspace matrices ----------------
 * symv_lower:  y = A_sym x   using only the lower triangle of A (n x n inside W)   (m = R (g/S), :204)
 * trmv_lower:  y = L x + y0  with L the lower triangle                              (f' = C eta + m, :140)
 */
int gpmc_symv_lower_f64(const double* W, int n, int ldw, const double* x, double* y, gpmc_stream_t stream);
int gpmc_trmv_lower_f64(const double* W, int n, int ldw, const double* x, const double* y0, double* y,
                        gpmc_stream_t stream);
int gpmc_symv_lower_f32(const float* W, int n, int ldw, const float* x, float* y, gpmc_stream_t stream);
int gpmc_trmv_lower_f32(const float* W, int n, int ldw, const float* x, const float* y0, float* y,
                        gpmc_stream_t stream);
/* y = y0 + alpha * (U x) for an upper-trapezoidal U stored by rows (row i: columns >= i; nothing left of the diagonal
 * is read); y0 may be NULL.  With U = P = s L^-T (gpmc_potrf_tri), x = alpha = L^-1 g (the riding row), y0 = g and
 * alpha = -1 this is m_theta_g = R S^-1 g = g - P (L^-1 g) of sliceSample.py:204 without forming R first. */
int gpmc_trmv_upper_f64(const double* W, int n, int ldw, const double* x, const double* y0, double alpha, double* y,
                        gpmc_stream_t stream);
int gpmc_trmv_upper_f32(const float* W, int n, int ldw, const float* x, const float* y0, double alpha, float* y,
                        gpmc_stream_t stream);

/* ---- a16: triangular solves with an existing factor (tools.solve_chol, sliceSample.py:258; also :120,145,202) ----
 * trsm_rows: R (nrows x np_, ldr; nrows multiple of 64) <- R L^-T, i.e. row r becomes (L^-1 b_r)^T, with L the lower
 *            triangle of the workspace matrix W (np_ x np_).  f64.
 * revt_lower: Lt = J L^T J (Lt[i][j] = L[n-1-j][n-1-i], lower triangular, strict upper zeroed).  Since
 *            R L^-1 = ((R J) Lt^-T) J, the backward half of solve_chol is trsm_rows on Lt with the rows' columns
 *            reversed.  n multiple of 32; Lt must not alias L.
 */
int gpmc_trsm_rows_f64(const double* W, int np_, int ldw, double* R, int nrows, int ldr, gpmc_stream_t stream);
int gpmc_revt_lower_f64(const double* L, int n, int ld, double* Lt, int ldo, gpmc_stream_t stream);

/* ---- dense product for the predictive mean of inf_mcmc (np.dot(Ks.T, alpha), sliceSample.py:266) -----------------
 * C (M x N, ldc) = A (M x K, lda) B^T (B: N x K, ldb); M, N multiples of 64, K multiple of 16.  f64, MFMA. */
int gpmc_gemm_nt_f64(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int M, int N, int K,
                     gpmc_stream_t stream);

/* ---- fused plan: one log-marginal evaluation per call -------------------------------------------
 * The metric's unit (a1 + a2 + a3 + a11): for hyper-parameters hyp (host, linear
 * scale: ell[1 or d], sf, sn) build K + S into the plan's workspace, factor it with g riding as an
 * extra row, reduce.  The launch sequence (~15 launches at N = 8192, two streams; the plan owns its second stream and
 * events) is recorded on the plan's third evaluation -- the first two launch eagerly, the second one timed as the
 * yardstick -- as ONE fork/join hipGraph and replayed; a recording that replays more than 4 % slower than the eager
 * yardstick (the runtime mapped its branch onto the main stream's hardware queue) is retired and recorded once more,
 * then the fallback form (linear hipGraphs cut at the cross-stream edges, replayed with eager event edges) is used.
 * All device memory is the caller's: `workspace` of gpmc_plan_workspace_bytes(n, d, elem_size).
 * The library assumes ONE device and ONE host thread per process (the reference is a single-threaded script,
 * framework.py:68; the deployment is one process per GPU): options, the bare entry points' eager look-ahead stream
 * and the per-kernel attribute flags are process-wide and unsynchronised.
 */
typedef struct gpmc_plan gpmc_plan;
size_t gpmc_plan_workspace_bytes(int n, int d, int elem_size);
int gpmc_plan_create(gpmc_plan** out, int kernel, int n, int d, int elem_size, void* workspace,
                     size_t workspace_bytes, gpmc_stream_t stream);
/* X: n x d host or device row-major (copied into the workspace), g: n elements */
int gpmc_plan_set_data(gpmc_plan* p, const void* X_dev, int ldx, const void* g_dev);
/* asynchronous: enqueue one evaluation; the result lands in host result slot `slot` (0..63).  Re-arming a slot whose
 * evaluation has not been read back with gpmc_plan_result returns -4 (nothing is enqueued).  The parameters are not
 * copied to the device: they are written into a ring of 64 blocks in pinned host memory, and the evaluation's kernels
 * read block (sequence number mod 64) through its device mapping (the reduction moves the device's sequence number on;
 * a block that names another execution makes gpmc_plan_result return -6 instead of a value -- it cannot happen
 * unless an enqueue failed part-way, after which the next call resynchronises).
 * kparams_host = kernel-ready parameters in gpmc_kbuild's hyp layout, as doubles: [sf^2, S_ii, ell...]
 * (nk = 3 for RBF, 2 + d for ARD) -- the caller forms them from the sampler's (ell, sf, sn) exactly as the
 * reference does: exp(log .) round trips and the S_ii expression of sliceSample.py:184-187. */
int gpmc_plan_logml_async(gpmc_plan* p, const double* kparams_host, int nk, int slot);
/* wait for the evaluation enqueued into `slot` (later evaluations keep running) and read it:
 * out_host[0]=logml, [1]=logdet, [2]=quad, [3]=info.  Frees the slot. */
int gpmc_plan_result(gpmc_plan* p, int slot, double* out_host);
/* device pointer / geometry of the factored workspace of the last evaluation (for callers that go on
 * to use L): W, mp, np_, ld */
int gpmc_plan_workspace(gpmc_plan* p, void** W, int* mp, int* np_, int* ld);
/* Several plans of one process may evaluate concurrently, each on its own stream (k independent chains on one card).  The
 * look-ahead's in-launch hand-over -- workgroups of a panel launch waiting, resident, for flags of the GEMM beside it -- is
 * only safe for a factorisation that has the card to itself, so it is used while ONE workspace has live plans in the process;
 * from the second on every plan hands over in stream order (as option shared_card does unconditionally), and a plan whose
 * launch sequence was recorded in the other mode records again at its next evaluation.  Returns 1 if evaluations of `p`
 * enqueued now use the in-launch hand-over, 0 if not.  (gpmc_potrf_* called directly follow shared_card only: do not run them
 * beside a plan's evaluation at look-ahead sizes without setting it.) */
int gpmc_plan_handover_in_launch(const gpmc_plan* p);
/* synchronises the plan's streams and releases the plan (its look-ahead stream, events and recorded launch graphs).
 * The stream passed to gpmc_plan_create must still be valid; the workspace is not touched. */
int gpmc_plan_destroy(gpmc_plan* p);
/* HIP runtimes before 7.2 (the one the torch wheel bundles is 7.0.51831) corrupt the heap when a two-stream graph exec
 * is destroyed (a use-after-free inside the runtime, profiles/r02_graph_abort_isolation.txt): on those runtimes retired
 * execs are parked for the life of the process instead.  execs / nodes = how many are parked and how many graph nodes
 * they hold (either pointer may be NULL).  Option "graveyard_max" (default 512) bounds the count: beyond it new plans
 * record the fallback form, whose single-stream execs are destroyed normally. */
int gpmc_graveyard_stats(int* execs, size_t* nodes);

/* instrumentation of the two kernels of the factorisation: between begin and read every launch of the trailing SYRK
 * update and of the tile-DAG kernel (look-ahead panels, tail) is bracketed by HIP events on the stream it goes to;
 * read() returns out6 = {total ms inside the update kernel, number of launches, algorithmic flops (2 K x
 * lower-triangle elements updated); the same three for the tile-DAG kernel (history + m^3/3 + m^2 per row below)}.
 * Not for graph capture. */
int gpmc_profile_begin(void);
int gpmc_profile_read(double* out6);

/* tuning knobs (process-wide, not thread-safe; defaults are the measured auto rules -- csrc/gpmc_common.h `Options`
 * documents each): schedule "outer_nb" (0 = auto, else a multiple of 128), "lookahead" (-1 auto / 0 / 1), "la_min_np",
 * "la_min_m", "la_nb128_m", "la_depth", "la_order", "panel_pair", "panel_prio", "panel_lds_pad", "use_graph",
 * "small_tile_m", "f32_nb", "gemm_lds_pad"; the tile-DAG schedule (round 2) "tail_m" / "tail_whole_m" (columns the
 * one-launch factorisation takes as a tail / whole; "_f32" variants for float storage), "tail_workers", "dag_workers",
 * "la_dag", "dag_hist", "dag_ahead", "dag_wprio", "result_poll_us" (gpmc_plan_result polls the result's stamp in pinned memory this long before waiting on the event; default 20000, 0 = event only), "tail_m_tri" / "tail_whole_m_tri" (tail_m / tail_whole_m for factorisations with trapezoidal riding rows, gpmc_potrf_tri: default 3072 / -1 = as tail_whole_m), "kbuild_split" (a plan builds the first look-ahead panel's columns first and the rest of K beside that panel; default 1), "tail_park", "tail_xcd" (1: worker tickets in eight queues by column, served XCD by XCD; default 0), "tail_skip" (bit 0: skip the unused half of a riding block, bit 1: its idle waves; gpmc_potrf_rhs), "tail_warm", "tail_roles_by_index" (measurements only), "tail_spin_limit",
 * "tail_acc16" (1: off-diagonal accumulation on v_mfma_f64_16x16x4; 0: the 4x4x4 form, same bits), "tail_prefetch_t" (tail /
 * whole launches of at least this many tile columns prefetch the next column's operands; 0 = never), "dag_prefetch",
 * "replay_form", "graveyard_max", "shared_card" (1: other processes use this card too -- see INTEGRATION.md section 4); alternative kernels
 * "gemm_impl", "gemm_persistent", "gemm_small_bk32", "gemm_8wave", "panel_impl",
 * "diag_impl".  Unknown names return an error. */
int gpmc_set_option(const char* name, int value);
int gpmc_get_option(const char* name);

/* ---- diagnostics (used by tests/ and scripts/, not by the product path) ------------------------------------------
 * panel_stamps / tail_stamps: s_memtime phase stamps of the fused panel kernel / of the tile-DAG kernel's pair
 *   workgroups (eager launches write 16 stamps per tile column into dbg_dev until called again with NULL);
 * graph_churn: per iteration record the plan's sequence (form 1 fork/join, 0 segments), replay, sync, optionally
 *   hipDeviceSynchronize, retire it -- the pattern that aborted in round 1;
 * plan_record_info: out4 = {eager yardstick ms, first replay ms, recording attempts, form (2 fork/join, 1 segments)};
 * plan_ops: ops / graph launches of the recorded sequence. */
int gpmc_debug_panel_stamps(double* W, int ld, int mp, int k0, int c0, int* info_dev, void* scratch, int np_,
                            unsigned long long* dbg_dev, gpmc_stream_t stream);
int gpmc_debug_tail_stamps(unsigned long long* dbg_dev);
int gpmc_debug_graph_churn(gpmc_plan* p, int form, int device_sync, int iters);
int gpmc_debug_plan_record_info(gpmc_plan* p, double* out4);
int gpmc_debug_plan_ops(gpmc_plan* p, int* n_ops, int* n_graphs);

#ifdef __cplusplus
}
#endif
#endif /* GPMC_H */
