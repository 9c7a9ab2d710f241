// gpmc_common.h -- shared device/host helpers for libgpmc (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#include "../../include/gpmc.h"

namespace gpmc {

// ---------------------------------------------------------------- error plumbing
extern thread_local char g_err[512];
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define GPMC_HIP(expr)                                                                            \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return gpmc::fail((int)e__, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
    } while (0)
#define GPMC_ARG(cond, k)                                                                         \
    do {                                                                                          \
        if (!(cond)) return gpmc::fail(-(k), "%s: bad argument %d: !(%s)", __func__, (k), #cond); \
    } while (0)
#define GPMC_LAUNCH_CHECK() GPMC_HIP(hipGetLastError())

// ---------------------------------------------------------------- options
struct Options {
    // ---- schedule (defaults = the auto rules of potrf())
    int outer_nb = 0;         // outer block of the right-looking Cholesky (multiple of 128); 0 = auto: 128 serial (f32: f32_nb
                              // from np_ 4096), under look-ahead 256 / 384 / 512 by remaining columns
    int lookahead = -1;       // factor panel k+1 on a second stream while the rest of update k runs; -1 = auto: np_ >= la_min_np
    int la_min_np = 4096;     // auto look-ahead from this padded size on (4096: 2.25 -> 2.10 ms, 5120: 3.16 -> 2.88; 3072: no gain);
                              // problems the tile-DAG launch (tail.h) takes whole never look ahead
    int la_nb128_m = 0;       // under look-ahead: outer block 128 once fewer columns than this remain (0 = never)
    int la_min_m = 2048;      // under look-ahead, where the tile-DAG launch is not available (f32, trapezoidal riding rows):
                              // switch to the serial NB = 128 chain once fewer columns than this remain
                              // (2048 vs 3072 with the 128-column panel launch: 8192 6.41 vs 6.51 ms, 6144 3.83 vs 3.88)
    int la_depth = 1;         // 1: next-panel-column update on the main stream; 2: on the panel stream (measured slower)
    int la_order = 0;         // look-ahead enqueue order per step: 0 = panel work first, 1 = big GEMM first (measured slower)
    int panel_pair = 1;       // first 128 columns of an outer panel in one launch (panel_pair.h): 0 never, 1 under look-ahead
                              // (6144: 3.96 -> 3.83 ms, 8192: 6.69 -> 6.50, 12288: -1.3 %, f32 16384: -2.3 %), 2 everywhere
                              // (serial phases: +1.7 % at 2048 / 4096)
    int panel_lds_pad = 0;    // extra LDS bytes requested by the 64-column panel kernel beside the GEMM; 16384+ makes it wait
                              // for an empty CU instead of sharing one: 8192 6.58 vs 6.41 ms, 16384 33.6 vs 32.8 -> stays 0
    int panel_prio = 1;       // the fused panel kernel raises its wave priority (s_setprio 3): 0-1.5 % under look-ahead, never worse
    int tail_m = 6144;        // f64: the last <= tail_m columns (a multiple of 64, <= 8192; 0 = never) are factored by the
                              // one-launch tile-DAG kernel (tail.h) instead of the serial NB = 128 launch chain
                              // (N = 8192, ms/eval: 4096 5.20, 5120 5.00, 6144 5.01)
    int tail_whole_m = 7168;  // problems of at most this many columns are ONE tile-DAG launch (7168: 3.91 ms vs 4.17 with a
                              // look-ahead phase in front; 8192: 5.57 vs 5.0)
    int tail_m_f32 = 6144, tail_whole_m_f32 = 6144;   // the same two extents for float storage (16x16x4 accumulation: f32 N = 8192
                              // 4.25 -> 3.93 ms with 6144 instead of 4096, 16384 19.2 -> 18.85; 5120 4.03, 7168 4.10)
    int tail_warm = 1;        // tile-DAG pairs run the factor once on an identity tile before their real pass (instruction cache, tail.h)
    int tail_roles_by_index = 0;   // measurements only: pairs = blocks 0 .. T-1 (can deadlock into the spin limit when processes share a card)
    int tail_acc16 = 1;       // tile-DAG off-diagonal accumulation on v_mfma_f64_16x16x4 (0: the 4x4x4 form, same bits; tail.h)
    int tail_prefetch_t = 72; // whole / tail tile-DAG launches of at least this many tile columns prefetch the next column's
                              // operands under the MFMA phase (0 = never; needs tail_acc16)
    int dag_prefetch = 0;     // ... look-ahead panel launches too (8192: +-0, 9216: -0.9 %, 12288: +2 %)
    int tail_skip = 3;        // what a tile-DAG launch leaves out when the caller has said how many riding rows are in use
                              // (gpmc_potrf_rhs): bit 0 the unused 64-row half of the riding block, bit 1 the MFMAs of waves
                              // whose 16 rows of the used half hold no right-hand side
    int tail_xcd = 0;         // 1: tile-DAG worker tickets in eight queues by column (mod 8), each served first by the workgroups of
                              // one XCD, so that an XCD's L2 holds one column's A operands instead of five columns' -- in launches
                              // that have the chip to themselves, never when other processes share the card.  Measured (same
                              // process, ms/eval off / on): N = 5120 1.548 / 1.517, 6144 2.130 / 2.093, 7168 3.022 / 3.010,
                              // 8192 4.270 / 4.269 (its tail launch carries 8 history columns), <= 4096 +-0; in the look-ahead
                              // panels (8 columns, 32 workers per queue) beside the GEMM +0.8 %.  Off: nothing for the headline.
    int result_poll_us = 20000;   // gpmc_plan_result polls the result's stamp in pinned memory for up to this long before it waits on the event (0: event only)
    int tail_whole_m_tri = -1;   // tail_whole_m for factorisations with trapezoidal riding rows; -1 = tail_whole_m
    int tail_m_tri = 3072;    // tail_m for factorisations with trapezoidal riding rows (gpmc_potrf_tri; 0 = tail_m): the riding trapezoid keeps the look-ahead GEMMs big, so the hand-over to the tile-DAG launch pays later (SDS proposal at N = 8192: 15.74 ms at 6144, 15.10 at 4096, 15.04 at 3072, 14.99 at 2048; N = 12288: 47.9 / 46.9 / 47.7 at 6144 / 3072 / 2048)
    int kbuild_split = 1;     // a plan builds the first look-ahead panel's columns first and the rest of K beside that panel's factorisation
    int tail_park = 1;        // tile-DAG workers step aside while a pair on their CU is inside its critical window (tail.h)
    int tail_workers = 0;     // worker workgroups of a tile-DAG launch that has the chip to itself (plus one per tile column);
                              // 0 = auto by the launch's bulk-to-chain ratio (potrf.hip, alone_workers: 256 - T, 256 or 480)
    int dag_workers = 0;      // ... of one that factors a look-ahead panel beside the trailing GEMM; 0 = auto: half the panel width
                              // (N = 8192, NB 256: 32 workers 9.16 ms, 64 6.42, 96 5.63, 128 5.38, 192 5.48, 480 6.35)
    int tail_spin_limit = 1 << 20;   // polls a waiting tile-DAG workgroup makes before it gives up (test hook: 0 forces the
                              // timeout path -- every wait that is not already satisfied aborts the launch)
    int dag_ahead = 1;        // look-ahead: the trailing GEMM hands the next-but-one panel's columns over tile by tile inside its
                              // launch (priority tile columns + flags) instead of at its end
    int shared_card = 0;      // other processes run kernels on this card too (chains.init_distributed sets it when ranks
                              // outnumber cards; independent processes set it themselves): the in-launch hand-over is
                              // replaced by the event edge, so that no resident workgroup waits for an unscheduled kernel
    int dag_wprio = 0;        // wave priority (0-2) of the worker workgroups of a look-ahead panel launch (the pairs run at 3)
    int dag_hist = 1;         // look-ahead panels: 1 = the tile-DAG launch accumulates the previous panel's update itself,
                              // 0 = a GEMM launch on the panel stream does it first
    int la_dag = 1;           // f64 look-ahead: factor every outer panel with one tile-DAG launch (potrf.hip), 0: round 1's
                              // next-panel update + per-64/128-column panel launches
    int use_graph = 1;        // record the plan's launch sequence once and replay it
    int replay_form = 1;      // 1: one fork/join hipGraph (6.6 ms at N=8192); 0: linear graphs cut at the cross-stream edges
                              // (7.9 ms; no exec spans two streams) -- see Sched below
    int graveyard_max = 512;  // HIP runtimes < 7.2: fork/join execs parked instead of destroyed (sched.hip); once this many
                              // are parked, new sequences are recorded in the SEGMENTS form
    int small_tile_m = 3072;  // trailing matrices with at most this many rows use 64x64 tiles
    int f32_nb = 256;         // outer block of the serial f32 factorisation for np_ >= 4096
    int gemm_lds_pad = -1;    // extra dynamic LDS bytes requested by the big-tile f64 GEMM under look-ahead; -1 = auto: 16000
                              // (one workgroup per CU) below np_ = 8192, 0 from there on
    // ---- alternative kernels kept for the measurements that led to the defaults
    int gemm_impl = 1;        // 1: v_mfma_f64_4x4x4 slab GEMM (f64), 0: v_mfma_*_16x16x4 GEMM
    int gemm_persistent = 0;  // big-tile f64 trailing update: persistent 8-wave kernel with C-tile prefetch (gemm4p.h)
    int gemm_small_bk32 = 0;  // small-tile (64x64) f64 GEMM: 32 k per LDS stage instead of 16
    int gemm_8wave = 0;       // with a padded LDS request (one GEMM workgroup per CU) use the 8-wave 128x128 variant
    int panel_impl = 1;       // 1: fused kernel (4 waves); 2: fused kernel with in-block look-ahead (5 waves, +-0.5 %);
                              // 0: three kernels (diag factor, strip TRSM, GEMM)
    int diag_impl = 3;        // with panel_impl 0: 0 LDS column loop, 1 single wave, 2 register-blocked VALU, 3 micro-panel + MFMA
};
extern Options g_opt;

// ---------------------------------------------------------------- vector / MFMA traits
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;

// v_mfma_f64_16x16x4_f64: A[row=l&15][k=l>>4], B[k=l>>4][col=l&15], D: col=l&15, row=(l>>4)+4*reg.
template <>
struct Mfma<double> {
    typedef d4_t acc_t;
    typedef d2_t frag_t;                 // one 16-byte LDS read = 2 k-values
    static constexpr int EPL = 2;        // elements per 16-byte lane read
    __device__ static inline acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // row of accumulator register e for lane-quarter q, and its inverse (row i lives in quarter rq(i), register re(i))
    __host__ __device__ static constexpr int crow(int q, int e) { return q + 4 * e; }
    __host__ __device__ static constexpr int rq(int i) { return i & 3; }
    __host__ __device__ static constexpr int re(int i) { return i >> 2; }
};
// v_mfma_f32_16x16x4_f32: same A/B maps, D: col=l&15, row=4*(l>>4)+reg.
template <>
struct Mfma<float> {
    typedef f4_t acc_t;
    typedef f4_t frag_t;                 // 4 k-values per 16-byte read
    static constexpr int EPL = 4;
    __device__ static inline acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __host__ __device__ static constexpr int crow(int q, int e) { return 4 * q + e; }
    __host__ __device__ static constexpr int rq(int i) { return i >> 2; }
    __host__ __device__ static constexpr int re(int i) { return i & 3; }
};

template <typename T>
__device__ inline typename Mfma<T>::acc_t acc_zero() {
    typename Mfma<T>::acc_t z;
    z[0] = 0; z[1] = 0; z[2] = 0; z[3] = 0;
    return z;
}

constexpr int NB_IN = 64;  // inner (diagonal-block) size of the Cholesky

// device scratch of potrf(): 16x16 inverse tiles (three-kernel panel path) | parked diagonal factors | parked L21
// blocks | sync words of the tail kernel
size_t potrf_scratch_bytes(int np_);
// set around a factorisation whose launches must not wait for one another inside the kernels (plan.hip: every plan but
// the process's one hand-over owner): the look-ahead hand-over is then the stream-ordered one, as with shared_card
extern bool g_ordered_handover;
constexpr int GPMC_RING_OUT_OF_STEP = -3000001;   // *info of a plan evaluation whose parameter block belongs to another execution
extern unsigned long long* g_tail_dbg;   // diagnostic stamp buffer of the tail kernel (nullptr in production)

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------- internal entry points
// housekeeping a kernel-build launch can carry for the log-marginal plan (kbuild.hip): copy a vector, clear a word
template <typename T>
struct KbuildExtras {
    const T* vec_src = nullptr;
    T* vec_dst = nullptr;
    int vec_n = 0;
    int* zero_word = nullptr;
    // column range of this launch, in tiles of 128 columns: [ctile0, ctile1); ctile1 = 0: every column.  The log-marginal plan
    // builds the first panel's columns first and the rest beside that panel's factorisation (plan.hip, Sched::after_fork).
    int ctile0 = 0, ctile1 = 0;
    // 16-byte words to clear (the factorisation's sync words: the fill launch in front of a tile-DAG launch rides here too),
    // by the workgroups of the launch's first column of tiles
    uint4* zero_ptr = nullptr;
    unsigned zero_quads = 0;
    // parameter ring (log-marginal plans): hyp_dev is then the base of a ring of 64 parameter blocks of ring_stride elements
    // in pinned, device-mapped HOST memory, and this launch reads block (*seq & 63) -- no host-to-device copy per evaluation
    const unsigned* seq = nullptr;
    int ring_stride = 0;
};
template <typename T>
int kbuild(int kernel, const T* X, int n, int d, int ldx, const T* Z, int nz, int ldz, const T* hyp_dev, T* K,
           int rows_out, int cols_out, int ldk, int flags, hipStream_t s, KbuildExtras<T> ex = KbuildExtras<T>());
struct Sched;
template <typename T>
int potrf(T* W, int mp, int np_, int ldw, int* info_dev, void* scratch, hipStream_t s, int ntri = 0, Sched* sc = nullptr,
          int rhs_rows = -1);
// host_base / slot_dev (both or neither): the four results also go straight to pinned host memory, 4 doubles at
// host_base + 4 * (int)*slot_dev, so that no copy has to follow the reduction
template <typename T>
int logdet_quad(const T* W, int n, int ldw, const T* alpha, const int* info_dev, double* out_dev, hipStream_t s,
                double* host_base = nullptr, const T* slot_dev = nullptr, unsigned* seq = nullptr, int ring_stride = 0);
template <typename T>
int pack_lower(const T* A, int n, int lda, T jitter, T* W, int np_, int ldw, hipStream_t s);
template <typename T>
int pack_rows(const T* B, int nrhs, int n, int ldb, T* W, int mp, int np_, int ldw, hipStream_t s);
template <typename T>
int unpack_lower(const T* W, int np_, int ldw, T* L, int n, int ldl, hipStream_t s);
template <typename T>
int unpack_rows(const T* W, int np_, int ldw, T* B, int nrhs, int n, int ldb, hipStream_t s);
template <typename T>
int reset_tri(T* R, int np_, int ld, const T* diag_dev, hipStream_t s);
template <typename T>
int veccopy(T* dst, const T* src, int n, hipStream_t s);
template <typename T>
int publish_host(const double* out4, const int* info2, const T* vec, int n, double* host_mapped, hipStream_t s);
template <typename T>
int shift_diag_sub(T* R, int ldr, int n, const T* shift_dev, T jitter, T* diag_out, const T* f, const T* m, T* row_out,
                   hipStream_t s);
// a caller's launch sequence (library launches and the caller's own kernels on one stream) recorded and replayed:
// gpmc_seq_* in plan.hip; retirement follows the same rule as the plans' execs (graveyard on runtimes < 7.2)
struct Seq {
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    hipStream_t stream = nullptr;
    bool capturing = false;
};
int seq_retire(Seq* q);
template <typename T>
int syrk_diff(const T* D, int ldd, const T* P, int kp, int ldp, T diag_add, T* C, int n, int np_, int ldc,
              hipStream_t s);
template <typename T>
int syrk_tri(const T* P, int ldp, T diag_add, T* C, int n, int np_, int ldc, hipStream_t s, T* C2 = nullptr);
template <typename T>
int symv_lower(const T* W, int n, int ldw, const T* x, T* y, hipStream_t s);
template <typename T>
int trmv_lower(const T* W, int n, int ldw, const T* x, const T* y0, T* y, hipStream_t s);
template <typename T>
int trmv_upper(const T* W, int n, int ldw, const T* x, const T* y0, double alpha, T* y, hipStream_t s);

// ---------------------------------------------------------------- launch schedule of one factorisation
// Where a factorisation's launches go: a main stream, a panel stream for the look-ahead, and the cross-stream edges
// between them.  Three ways to run the same host code (potrf() only ever calls edge_record / edge_wait and launches
// kernels on sc.main / sc.panel):
//   EAGER     edges are hipEventRecord / hipStreamWaitEvent, kernels run as they are launched.
//   FORKJOIN  (the plan's default replay form) main is in stream capture and the edges pull the panel stream into the
//             same capture: ONE graph whose cross-stream dependencies the runtime resolves itself -- measured free on
//             the device timeline (scripts/graph_launch_cost.hip: a 240-kernel two-stream ping-pong replays in
//             1798 us as one fork/join graph, 1794 us as a single-stream graph).
//   SEGMENTS  both streams are in stream capture and every edge CUTS the capture of the stream it touches: an op list
//             [linear graph | record | wait | linear graph | ...] that replay() re-issues in order.  No exec spans two
//             streams, so none of the fork/join hazards below exist -- but every eager edge costs ~13-16 us and every
//             small graph launch ~5 us on the device timeline (same script: 3478 us for the ping-pong), +1.2 ms per
//             evaluation at N = 8192 (7.9 vs 6.6 ms).  Kept as the deterministic fallback (replay_form = 0).
// Fork/join hazard on the HIP runtime bundled with the torch wheel (7.0.51831), root-caused in round 2
// (profiles/r02_graph_abort_isolation.txt): destroying a fork/join exec after a device-wide synchronisation is a
// use-after-free INSIDE THE RUNTIME -- its signal-handler thread (amd::roc::HsaAmdSignalHandler ->
// VirtualGPU::updateCommandsState -> ReferenceCountedObject::release -> free()) still owns the marker command the
// device sync put on the exec's internal branch stream when hipGraphExecDestroy frees it.  The identical libgpmc
// binary, capture and destroy sequence is clean on runtime 7.2.26015 (with or without torch in the process), and the
// 7.0 runtime aborts the same way from a plain C++ harness, so neither torch nor this library's capture pattern is an
// ingredient.  Round 2 flushed the exec (one launch with every node disabled + a stream sync) before destroying it,
// which narrowed the race but left one host segmentation fault in a later capture.  Round 3: on runtimes < 7.2 a
// fork/join exec is never destroyed -- drop_ops() parks it in a process-wide graveyard (sched.hip; count and node
// total readable through gpmc_graveyard_stats, bounded by the option graveyard_max, beyond which sequences are
// recorded in the SEGMENTS form).
struct SeqOp {
    int kind;              // 0 graph launch, 1 event record, 2 stream wait
    int on_panel;          // which stream: 0 main, 1 panel
    hipGraphExec_t exec;
    hipEvent_t ev;
    hipGraph_t graph;      // two-stream fork/join op only: the template graph (parked with the exec on runtimes < 7.2)
};
struct Sched {
    enum Mode { EAGER = 0, SEGMENTS = 1, FORKJOIN = 2 };
    int mode = EAGER;
    hipStream_t main = nullptr, panel = nullptr;
    bool own_panel = false;
    std::vector<hipEvent_t> events;      // EAGER / FORKJOIN: a ring; SEGMENTS: one event per edge
    size_t ev_next = 0;
    std::vector<SeqOp> ops;
    bool cap_main = false, cap_panel = false;
    bool forked = false;                 // FORKJOIN: an edge has pulled the panel stream into the capture
    bool info_cleared = false;           // the caller has already cleared *info_dev on the main stream (potrf skips its memset)

    int init(hipStream_t main_stream);                       // creates the panel stream and a first batch of events
    int reserve_events(size_t n);
    // The caller's matrix builder (optional): build_cols(stream, c0, c1) enqueues whatever writes columns [c0, c1) of the
    // matrix (multiples of 128; all rows, riding rows included).  When it is set the caller has NOT built the matrix:
    // potrf() calls it -- for everything before its first launch, or, where its schedule starts with a look-ahead panel on
    // the second stream, for that panel's columns first and for the rest on the main stream BESIDE the panel's
    // factorisation (the log-marginal plan's K build: 72 us at N = 8192, of which the first panel needs an eighth).
    // (zero_ptr, zero_bytes: sync words potrf wants cleared before its first launch; the builder clears them in its own
    // launch that holds column 0 -- no fill launch in front of the tile-DAG launch)
    std::function<int(hipStream_t, int, int, void*, size_t)> build_cols;
    int edge_record(hipStream_t s, hipEvent_t* out);         // out = the event recorded on s
    int edge_wait(hipStream_t s, hipEvent_t e);
    int begin(int mode_);                                    // SEGMENTS / FORKJOIN: start capturing
    int end();                                               // ... stop; the op list is complete
    int replay();
    int drop_ops();                                          // retire the recorded graphs (see the hazard note above)
    int destroy();                                           // + events and the panel stream
  private:
    int cut(hipStream_t s);
};
bool forkjoin_allowed();                        // false once the graveyard of un-destroyable execs is full
void graveyard_stats(int* execs, size_t* nodes);
Sched* eager_sched(hipStream_t main_stream);   // process-wide EAGER schedule for the bare gpmc_potrf_* entry points
int trsm_rows_f64(const double* W, int np_, int ldw, double* R, int nrows, int ldr, hipStream_t s);
int revt_lower_f64(const double* L, int n, int ld, double* Lt, int ldo, hipStream_t s);
int debug_panel_stamps(double* W, int ld, int mp, int k0, int c0, int* info_dev, double* lpark,
                       unsigned long long* dbg_dev, hipStream_t s);
int gemm_nt_f64(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int M, int N, int K,
                hipStream_t s);
int profile_begin();
int profile_read(double* out);

}  // namespace gpmc
