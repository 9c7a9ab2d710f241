// plan.hip -- fused log-marginal plan + the extern "C" surface of libgpmc.so.
//
// One plan = one (kernel, n, d, dtype) problem bound to caller-owned device memory.  An evaluation is
//   H2D(16..80 B of kernel parameters + the result slot's index) -> [ kbuild(lower, S on the diagonal, identity
//   padding; also copies g into its riding row and clears info) -> blocked Cholesky -> log-det / quadratic-form
//   reduction, which writes the pinned result slot itself ]
// The bracketed part is a static launch sequence (f64: ~15 launches at N = 8192 on two streams -- kernel build, one fill,
// 6 tile-DAG panel launches beside 5 trailing GEMMs, the tile-DAG tail, the reduction; one launch chain of ~200 for
// f32 storage).  It is recorded once through `Sched` (gpmc_common.h) and replayed: by default as ONE fork/join hipGraph
// whose cross-stream edges the runtime resolves, with the SEGMENTS form (linear graphs cut at the edges) as the
// deterministic fallback.
#include <cmath>
#include <new>

#include "gpmc_common.h"
#include <atomic>
#include <chrono>
#include <map>

namespace gpmc {

constexpr int PLAN_SLOTS = 64;
constexpr int PLAN_MAXD = 62;

struct PlanLayout {
    int np_, mp, ld;
    size_t off_W, off_X, off_g, off_hyp, off_out, off_info, off_scratch, total;
};

static PlanLayout plan_layout(int n, int d, int es) {
    PlanLayout L;
    L.np_ = round_up(n, GPMC_PAD);
    L.mp = L.np_ + GPMC_PAD;
    L.ld = L.np_;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    L.off_W = take((size_t)L.mp * L.ld * es);
    L.off_X = take((size_t)n * d * es);
    L.off_g = take((size_t)n * es);
    L.off_hyp = take((size_t)(2 + PLAN_MAXD + 2) * es);      // (round 1-3a: the device copy of the parameter block; unused since the ring)
    L.off_out = take(4 * sizeof(double));
    L.off_info = take(sizeof(int));
    L.off_scratch = take(potrf_scratch_bytes(L.np_));
    L.total = o;
    return L;
}

}  // namespace gpmc

struct gpmc_plan {
    int kernel, n, d, es;
    gpmc::PlanLayout L;
    char* ws;
    hipStream_t stream;
    double* host_pin;     // PLAN_SLOTS x (2+MAXD) staged parameters (as T bytes) + PLAN_SLOTS x 4 results
    size_t hyp_slot_bytes;
    double* host_res_dev; // device address of the pinned results (PLAN_SLOTS x 4 doubles)
    // Parameter ring: the staged parameters are never copied to the device.  Every EXECUTION of the launch sequence (eager,
    // replay, a recording attempt) reads block (seq & 63) of the pinned ring through its device mapping and its reduction
    // moves seq on; the host writes the block for execution number `execs` before it launches it.  At most 64 evaluations
    // are in flight (each holds a result slot), so a block is rewritten only after its execution has finished.
    void* ring_dev;       // device address of host_pin
    unsigned* seq_dev;    // device word: executions completed (its own small allocation: plans may share a workspace)
    unsigned execs;       // executions launched
    unsigned slot_exec[64];   // the execution whose result a busy slot waits for (its number mod 2^20 is the slot's stamp)
    bool resync;          // an execution may have been launched in part: set seq = execs before the next one
    char cur_stage[(2 + 62 + 2) * sizeof(double)];   // the current evaluation's parameter block (as T bytes) + slot index + execution number
    gpmc::Sched sched;    // the plan's own panel stream, events and recorded launch sequence
    bool recorded;
    hipEvent_t ev0, ev1;  // timing of the eager first evaluation and of the first replay (see record_sequence)
    float eager_ms, replay_ms;
    int record_attempts, recorded_form;
    int evals;
    bool have_data;
    bool ordered;         // hand-over mode of the launches being enqueued / of the recorded sequence (plans_must_order)
    unsigned char slot_busy[64];   // PLAN_SLOTS: 1 while an evaluation using the slot's staging / result is in flight
    hipEvent_t slot_ev[64];        // recorded behind the slot's result copy
};

// The in-launch hand-over of the look-ahead (potrf.hip, dag_ahead) lets a panel launch's workgroups wait, resident, for
// flags of the trailing GEMM that runs beside it.  A factorisation that has the card to itself always finishes; next to
// ANY other launch with resident waiting workgroups -- another process's (round 2's two-processes-on-one-card abort), or
// another plan of this process evaluating concurrently on its own stream, its tile-DAG tail launch included -- the
// waiters of both can hold every CU slot while the GEMM they wait for cannot place a block.  So the in-launch form is
// used only while ONE workspace has live plans in the process (plans on the same workspace cannot run concurrently:
// LogMarginalPlan's settle creates its second candidate there); as soon as a second one exists every plan hands over in
// stream order (what option shared_card selects unconditionally), and a plan whose recorded sequence was made in the
// other mode records again at its next evaluation.
static std::map<char*, int> g_live_ws;
static bool plans_must_order() { return g_live_ws.size() > 1; }

namespace gpmc {

struct OrderedScope {
    bool saved;
    explicit OrderedScope(bool on) : saved(g_ordered_handover) { g_ordered_handover = on; }
    ~OrderedScope() { g_ordered_handover = saved; }
};

template <typename T>
static int plan_enqueue_body(gpmc_plan* p, hipStream_t s) {
    OrderedScope ordered(p->ordered);
    const PlanLayout& L = p->L;
    T* W = reinterpret_cast<T*>(p->ws + L.off_W);
    const T* X = reinterpret_cast<const T*>(p->ws + L.off_X);
    const T* g = reinterpret_cast<const T*>(p->ws + L.off_g);
    const T* hyp = reinterpret_cast<const T*>(p->ring_dev);     // ring base (pinned host memory): block (*seq & 63)
    constexpr int RING_STRIDE = 2 + PLAN_MAXD + 2;
    double* out = reinterpret_cast<double*>(p->ws + L.off_out);
    int* info = reinterpret_cast<int*>(p->ws + L.off_info);
    // the kernel build also puts g into its riding row and clears info (no copy / fill launches of their own), and the
    // reduction writes the result into the pinned slot whose index travelled with the parameters (no copy after it)
    KbuildExtras<T> ex;
    ex.vec_src = g;
    ex.vec_dst = W + (size_t)L.np_ * L.ld;
    ex.vec_n = p->n;
    ex.zero_word = info;
    ex.seq = p->seq_dev;
    ex.ring_stride = RING_STRIDE;
    // potrf() calls the build itself (Sched::build_cols): all of K before its first launch, or -- where its schedule opens
    // with a look-ahead panel on the second stream -- that panel's columns first and the rest beside its factorisation
    gpmc_plan* const pp = p;
    p->sched.build_cols = [=](hipStream_t st, int c0, int c1, void* zero_ptr, size_t zero_bytes) -> int {
        KbuildExtras<T> e2;
        if (c0 == 0) e2 = ex;          // g and the info word ride on the launch that holds tile (0, 0)
        e2.seq = ex.seq;
        e2.ring_stride = ex.ring_stride;
        if (zero_bytes) {
            e2.zero_ptr = reinterpret_cast<uint4*>(zero_ptr);
            e2.zero_quads = (unsigned)(zero_bytes / 16);
        }
        e2.ctile0 = c0 / 128;
        e2.ctile1 = (c1 + 127) / 128;
        return kbuild<T>(pp->kernel, X, pp->n, pp->d, pp->d, nullptr, 0, 0, hyp, W, pp->L.np_, pp->L.np_, pp->L.ld,
                         GPMC_KB_LOWER | GPMC_KB_PAD_IDENTITY, st, e2);
    };
    p->sched.info_cleared = true;
    int rc = potrf<T>(W, L.mp, L.np_, L.ld, info, p->ws + L.off_scratch, s, 0, &p->sched, 1);      // one riding row: g
    p->sched.info_cleared = false;
    p->sched.build_cols = nullptr;
    if (rc) return rc;
    return logdet_quad<T>(W, p->n, L.ld, W + (size_t)L.np_ * L.ld, info, out, s, p->host_res_dev, hyp + (2 + PLAN_MAXD),
                          p->seq_dev, RING_STRIDE);
}

static int plan_enqueue(gpmc_plan* p, hipStream_t s) {
    return p->es == 8 ? plan_enqueue_body<double>(p, s) : plan_enqueue_body<float>(p, s);
}

// Record the plan's launch sequence and run it once, timed (the run IS the evaluation the caller asked for).
// A fork/join exec's branch is mapped onto a hardware queue by the runtime at instantiation; measured on both HIP
// runtimes here, about one instantiation in ten lands the branch on the main stream's queue and the look-ahead
// serialises (+12 % at N = 8192 with the tile-DAG schedule; each exec is stable for its lifetime).  The recorded
// sequence must not be slower than launching the same kernels eagerly, so a replay more than 4 % over the eager
// yardstick is retired (Sched::drop_ops, the one way an exec leaves: parked, never destroyed, on runtimes < 7.2) and
// recorded ONCE more; then the linear-graph form, which has no runtime-chosen queue, is recorded instead.  The
// yardstick is the plan's SECOND eager evaluation (the first also pays hipFuncSetAttribute, code-object loads and cold
// host launch paths, which let a +12 % exec through).  One-stream schedules have nothing to alias: they are recorded
// once and never timed against anything.
// the parameter block of the execution about to be launched (see gpmc_plan: parameter ring)
static void stage_params(gpmc_plan* p) {
    const size_t bytes = (size_t)(2 + PLAN_MAXD + 2) * p->es;
    // the block carries its execution number (mod 2^20: exact in float too); the reduction compares it with the device's
    // sequence number and reports GPMC_RING_OUT_OF_STEP instead of a result if they ever differ
    if (p->es == 8) reinterpret_cast<double*>(p->cur_stage)[2 + PLAN_MAXD + 1] = (double)(p->execs & 0xfffffu);
    else reinterpret_cast<float*>(p->cur_stage)[2 + PLAN_MAXD + 1] = (float)(p->execs & 0xfffffu);
    memcpy(reinterpret_cast<char*>(p->host_pin) + (size_t)(p->execs & 63u) * bytes, p->cur_stage, bytes);
    p->execs++;
}

static int record_sequence(gpmc_plan* p) {
    // (a plan whose eager evaluations ran while use_graph was off has no yardstick: its first recording is accepted)
    const bool have_yardstick = p->ev0 != nullptr && p->eager_ms > 0.f;
    if (!p->ev0) { GPMC_HIP(hipEventCreate(&p->ev0)); GPMC_HIP(hipEventCreate(&p->ev1)); }
    const int max_forkjoin = (g_opt.replay_form && forkjoin_allowed()) ? 2 : 0;
    for (int attempt = 0; attempt <= max_forkjoin; ++attempt) {
        const int form = attempt < max_forkjoin ? Sched::FORKJOIN : Sched::SEGMENTS;
        int rc = p->sched.begin(form);
        if (rc) return rc;
        rc = plan_enqueue(p, p->sched.main);      // (fork/join: the stream the main chain is captured on)
        const bool two_streams = p->sched.forked;
        const int rc2 = p->sched.end();
        if (rc || rc2) { (void)p->sched.drop_ops(); return rc ? rc : rc2; }
        GPMC_HIP(hipEventRecord(p->ev0, p->stream));
        stage_params(p);
        rc = p->sched.replay();
        if (rc) return rc;
        GPMC_HIP(hipEventRecord(p->ev1, p->stream));
        GPMC_HIP(hipEventSynchronize(p->ev1));
        GPMC_HIP(hipEventElapsedTime(&p->replay_ms, p->ev0, p->ev1));
        p->record_attempts = attempt + 1;
        p->recorded_form = form;
        if (form == Sched::SEGMENTS || !two_streams || !have_yardstick || p->replay_ms <= 1.04f * p->eager_ms + 0.02f) break;
        rc = p->sched.drop_ops();
        if (rc) return rc;
    }
    p->recorded = true;
    return 0;
}

}  // namespace gpmc

using namespace gpmc;

extern "C" {

int gpmc_version(void) { return GPMC_VERSION; }
const char* gpmc_last_error(void) { return g_err; }
int gpmc_pad(int n) { return round_up(n, GPMC_PAD); }
size_t gpmc_potrf_scratch_bytes(int np_) { return potrf_scratch_bytes(np_); }

int gpmc_set_option(const char* name, int value) {
    if (!name) return fail(-1, "gpmc_set_option: null name");
    if (!strcmp(name, "outer_nb")) { if (value != 0 && (value < 128 || value % 128)) return fail(-2, "outer_nb must be 0 (auto) or a multiple of 128"); g_opt.outer_nb = value; }
    else if (!strcmp(name, "lookahead")) g_opt.lookahead = value < 0 ? -1 : (value ? 1 : 0);
    else if (!strcmp(name, "use_graph")) g_opt.use_graph = value ? 1 : 0;
    else if (!strcmp(name, "small_tile_m")) g_opt.small_tile_m = value;
#ifdef GPMC_EXPERIMENTS
    else if (!strcmp(name, "diag_impl")) g_opt.diag_impl = value;
    else if (!strcmp(name, "panel_impl")) g_opt.panel_impl = value;
    else if (!strcmp(name, "gemm_persistent")) g_opt.gemm_persistent = value;
#else
    else if (!strcmp(name, "diag_impl")) { if (value != 0 && value != 3) return fail(-2, "diag_impl %d needs a `make EXPERIMENTS=1` build", value); g_opt.diag_impl = value; }
    else if (!strcmp(name, "panel_impl")) { if (value != 0 && value != 1) return fail(-2, "panel_impl %d needs a `make EXPERIMENTS=1` build", value); g_opt.panel_impl = value; }
    else if (!strcmp(name, "gemm_persistent")) { if (value) return fail(-2, "gemm_persistent needs a `make EXPERIMENTS=1` build"); g_opt.gemm_persistent = 0; }
#endif
    else if (!strcmp(name, "gemm_impl")) g_opt.gemm_impl = value;
    else if (!strcmp(name, "gemm_lds_pad")) g_opt.gemm_lds_pad = value;
    else if (!strcmp(name, "la_min_m")) g_opt.la_min_m = value;
    else if (!strcmp(name, "la_min_np")) g_opt.la_min_np = value;
    else if (!strcmp(name, "la_nb128_m")) g_opt.la_nb128_m = value;
    else if (!strcmp(name, "panel_pair")) g_opt.panel_pair = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(name, "panel_lds_pad")) g_opt.panel_lds_pad = value < 0 ? 0 : value;
    else if (!strcmp(name, "panel_prio")) g_opt.panel_prio = value ? 1 : 0;
    else if (!strcmp(name, "la_order")) g_opt.la_order = value ? 1 : 0;
    else if (!strcmp(name, "la_depth")) g_opt.la_depth = (value == 2) ? 2 : 1;
    else if (!strcmp(name, "gemm_8wave")) g_opt.gemm_8wave = value;
    else if (!strcmp(name, "f32_nb")) g_opt.f32_nb = value;
    else if (!strcmp(name, "gemm_small_bk32")) g_opt.gemm_small_bk32 = value;
    else if (!strcmp(name, "replay_form")) g_opt.replay_form = value ? 1 : 0;
    else if (!strcmp(name, "graveyard_max")) g_opt.graveyard_max = value < 0 ? 0 : value;
    else if (!strcmp(name, "la_dag")) g_opt.la_dag = value ? 1 : 0;
    else if (!strcmp(name, "dag_hist")) g_opt.dag_hist = value ? 1 : 0;
    else if (!strcmp(name, "tail_whole_m")) g_opt.tail_whole_m = value < 0 ? 0 : value;
    else if (!strcmp(name, "tail_m_f32")) { if (value < 0 || value > 8192 || value % 64) return fail(-2, "tail_m_f32 must be a multiple of 64 in [0, 8192]"); g_opt.tail_m_f32 = value; }
    else if (!strcmp(name, "tail_whole_m_f32")) g_opt.tail_whole_m_f32 = value < 0 ? 0 : value;
    else if (!strcmp(name, "dag_ahead")) g_opt.dag_ahead = value ? 1 : 0;
    else if (!strcmp(name, "shared_card")) g_opt.shared_card = value ? 1 : 0;
    else if (!strcmp(name, "tail_spin_limit")) g_opt.tail_spin_limit = value < 0 ? 0 : value;
    else if (!strcmp(name, "dag_wprio")) g_opt.dag_wprio = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(name, "tail_workers")) g_opt.tail_workers = value < 0 ? 0 : value;   // 0 = auto (potrf.hip, alone_workers)
    else if (!strcmp(name, "tail_park")) g_opt.tail_park = value ? 1 : 0;
    else if (!strcmp(name, "kbuild_split")) g_opt.kbuild_split = value ? 1 : 0;
    else if (!strcmp(name, "tail_whole_m_tri")) { if (value < -1 || (value > 0 && value % 128)) return fail(-2, "tail_whole_m_tri must be -1 or a multiple of 128"); g_opt.tail_whole_m_tri = value; }
    else if (!strcmp(name, "tail_m_tri")) { if (value < 0 || value % 128) return fail(-2, "tail_m_tri must be a multiple of 128"); g_opt.tail_m_tri = value; }
    else if (!strcmp(name, "result_poll_us")) g_opt.result_poll_us = value < 0 ? 0 : value;
    else if (!strcmp(name, "tail_xcd")) g_opt.tail_xcd = value ? 1 : 0;
    else if (!strcmp(name, "tail_skip")) g_opt.tail_skip = value & 3;
    else if (!strcmp(name, "tail_acc16")) g_opt.tail_acc16 = value ? 1 : 0;
    else if (!strcmp(name, "tail_prefetch_t")) g_opt.tail_prefetch_t = value;
    else if (!strcmp(name, "dag_prefetch")) g_opt.dag_prefetch = value ? 1 : 0;
    else if (!strcmp(name, "tail_roles_by_index")) g_opt.tail_roles_by_index = value ? 1 : 0;
    else if (!strcmp(name, "tail_warm")) g_opt.tail_warm = value ? 1 : 0;
    else if (!strcmp(name, "dag_workers")) g_opt.dag_workers = value < 0 ? 0 : value;
    else if (!strcmp(name, "tail_m")) { if (value < 0 || value > 8192 || value % 64) return fail(-2, "tail_m must be a multiple of 64 in [0, 8192]"); g_opt.tail_m = value; }
    else return fail(-1, "gpmc_set_option: unknown option '%s'", name);
    return 0;
}
int gpmc_get_option(const char* name) {
    if (!name) return -1;
    if (!strcmp(name, "outer_nb")) return g_opt.outer_nb;
    if (!strcmp(name, "lookahead")) return g_opt.lookahead;
    if (!strcmp(name, "use_graph")) return g_opt.use_graph;
    if (!strcmp(name, "small_tile_m")) return g_opt.small_tile_m;
    if (!strcmp(name, "diag_impl")) return g_opt.diag_impl;
    if (!strcmp(name, "panel_impl")) return g_opt.panel_impl;
    if (!strcmp(name, "gemm_impl")) return g_opt.gemm_impl;
    if (!strcmp(name, "gemm_lds_pad")) return g_opt.gemm_lds_pad;
    if (!strcmp(name, "la_min_m")) return g_opt.la_min_m;
    if (!strcmp(name, "la_min_np")) return g_opt.la_min_np;
    if (!strcmp(name, "la_nb128_m")) return g_opt.la_nb128_m;
    if (!strcmp(name, "la_depth")) return g_opt.la_depth;
    if (!strcmp(name, "la_order")) return g_opt.la_order;
    if (!strcmp(name, "panel_prio")) return g_opt.panel_prio;
    if (!strcmp(name, "panel_lds_pad")) return g_opt.panel_lds_pad;
    if (!strcmp(name, "panel_pair")) return g_opt.panel_pair;
    if (!strcmp(name, "gemm_8wave")) return g_opt.gemm_8wave;
    if (!strcmp(name, "f32_nb")) return g_opt.f32_nb;
    if (!strcmp(name, "gemm_small_bk32")) return g_opt.gemm_small_bk32;
    if (!strcmp(name, "gemm_persistent")) return g_opt.gemm_persistent;
    if (!strcmp(name, "replay_form")) return g_opt.replay_form;
    if (!strcmp(name, "graveyard_max")) return g_opt.graveyard_max;
#ifdef GPMC_EXPERIMENTS
    if (!strcmp(name, "experiments")) return 1;
#else
    if (!strcmp(name, "experiments")) return 0;
#endif
    if (!strcmp(name, "la_dag")) return g_opt.la_dag;
    if (!strcmp(name, "dag_hist")) return g_opt.dag_hist;
    if (!strcmp(name, "tail_whole_m")) return g_opt.tail_whole_m;
    if (!strcmp(name, "tail_m_f32")) return g_opt.tail_m_f32;
    if (!strcmp(name, "tail_whole_m_f32")) return g_opt.tail_whole_m_f32;
    if (!strcmp(name, "dag_ahead")) return g_opt.dag_ahead;
    if (!strcmp(name, "shared_card")) return g_opt.shared_card;
    if (!strcmp(name, "tail_spin_limit")) return g_opt.tail_spin_limit;
    if (!strcmp(name, "dag_wprio")) return g_opt.dag_wprio;
    if (!strcmp(name, "tail_workers")) return g_opt.tail_workers;
    if (!strcmp(name, "tail_park")) return g_opt.tail_park;
    if (!strcmp(name, "kbuild_split")) return g_opt.kbuild_split;
    if (!strcmp(name, "tail_m_tri")) return g_opt.tail_m_tri;
    if (!strcmp(name, "tail_whole_m_tri")) return g_opt.tail_whole_m_tri;
    if (!strcmp(name, "result_poll_us")) return g_opt.result_poll_us;
    if (!strcmp(name, "tail_xcd")) return g_opt.tail_xcd;
    if (!strcmp(name, "tail_skip")) return g_opt.tail_skip;
    if (!strcmp(name, "tail_acc16")) return g_opt.tail_acc16;
    if (!strcmp(name, "tail_prefetch_t")) return g_opt.tail_prefetch_t;
    if (!strcmp(name, "dag_prefetch")) return g_opt.dag_prefetch;
    if (!strcmp(name, "tail_roles_by_index")) return g_opt.tail_roles_by_index;
    if (!strcmp(name, "tail_warm")) return g_opt.tail_warm;
    if (!strcmp(name, "dag_workers")) return g_opt.dag_workers;
    if (!strcmp(name, "tail_m")) return g_opt.tail_m;
    return -1;
}

int gpmc_trsm_rows_f64(const double* W, int np_, int ldw, double* R, int nrows, int ldr, gpmc_stream_t stream) {
    return trsm_rows_f64(W, np_, ldw, R, nrows, ldr, reinterpret_cast<hipStream_t>(stream));
}
int gpmc_revt_lower_f64(const double* L, int n, int ld, double* Lt, int ldo, gpmc_stream_t stream) {
    return revt_lower_f64(L, n, ld, Lt, ldo, reinterpret_cast<hipStream_t>(stream));
}
int gpmc_gemm_nt_f64(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int M, int N, int K,
                     gpmc_stream_t stream) {
    return gemm_nt_f64(A, lda, B, ldb, C, ldc, M, N, K, reinterpret_cast<hipStream_t>(stream));
}
/* diagnostic only (gpmc.h, diagnostics): phase stamps of the fused panel kernel */
int gpmc_debug_panel_stamps(double* W, int ld, int mp, int k0, int c0, int* info_dev, void* scratch, int np_,
                            unsigned long long* dbg_dev, gpmc_stream_t stream) {
    double* lpark = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + (size_t)(np_ / NB_IN + 1) * 1024 * sizeof(double));
    return debug_panel_stamps(W, ld, mp, k0, c0, info_dev, lpark, dbg_dev, reinterpret_cast<hipStream_t>(stream));
}
/* diagnostic only (gpmc.h, diagnostics): eager launches of the tail kernel write 16 s_memtime stamps per tile column into
   dbg_dev (>= 16 * 64 u64) until this is called again with NULL */
int gpmc_debug_tail_stamps(unsigned long long* dbg_dev) { gpmc::g_tail_dbg = dbg_dev; return 0; }
int gpmc_profile_begin(void) { return profile_begin(); }
int gpmc_profile_read(double* out6) {
    if (!out6) return fail(-1, "gpmc_profile_read: null out");
    return profile_read(out6);
}

#define S_(x) reinterpret_cast<hipStream_t>(x)

int gpmc_kbuild_f64(int kernel, const double* X, int n, int d, int ldx, const double* Z, int nz, int ldz,
                    const double* hyp_dev, double* K, int rows_out, int cols_out, int ldk, int flags,
                    gpmc_stream_t stream) {
    return kbuild<double>(kernel, X, n, d, ldx, Z, nz, ldz, hyp_dev, K, rows_out, cols_out, ldk, flags, S_(stream));
}
int gpmc_kbuild_f32(int kernel, const float* X, int n, int d, int ldx, const float* Z, int nz, int ldz,
                    const float* hyp_dev, float* K, int rows_out, int cols_out, int ldk, int flags,
                    gpmc_stream_t stream) {
    return kbuild<float>(kernel, X, n, d, ldx, Z, nz, ldz, hyp_dev, K, rows_out, cols_out, ldk, flags, S_(stream));
}
int gpmc_potrf_f64(double* W, int mp, int np_, int ldw, int* info_dev, void* scratch, gpmc_stream_t stream) {
    return potrf<double>(W, mp, np_, ldw, info_dev, scratch, S_(stream));
}
int gpmc_potrf_f32(float* W, int mp, int np_, int ldw, int* info_dev, void* scratch, gpmc_stream_t stream) {
    return potrf<float>(W, mp, np_, ldw, info_dev, scratch, S_(stream));
}
int gpmc_potrf_tri_f64(double* W, int mp, int np_, int ldw, int ntri, int* info_dev, void* scratch, gpmc_stream_t stream) {
    return potrf<double>(W, mp, np_, ldw, info_dev, scratch, S_(stream), ntri);
}
int gpmc_potrf_tri_f32(float* W, int mp, int np_, int ldw, int ntri, int* info_dev, void* scratch, gpmc_stream_t stream) {
    return potrf<float>(W, mp, np_, ldw, info_dev, scratch, S_(stream), ntri);
}
int gpmc_veccopy_f64(double* dst, const double* src, int n, gpmc_stream_t stream) { return veccopy<double>(dst, src, n, S_(stream)); }
int gpmc_veccopy_f32(float* dst, const float* src, int n, gpmc_stream_t stream) { return veccopy<float>(dst, src, n, S_(stream)); }
int gpmc_shift_diag_sub_f64(double* R, int ldr, int n, const double* shift_dev, double jitter, double* diag_out, const double* f,
                            const double* m, double* row_out, gpmc_stream_t stream) {
    return shift_diag_sub<double>(R, ldr, n, shift_dev, jitter, diag_out, f, m, row_out, S_(stream));
}
int gpmc_shift_diag_sub_f32(float* R, int ldr, int n, const float* shift_dev, float jitter, float* diag_out, const float* f,
                            const float* m, float* row_out, gpmc_stream_t stream) {
    return shift_diag_sub<float>(R, ldr, n, shift_dev, jitter, diag_out, f, m, row_out, S_(stream));
}
int gpmc_publish_host_f64(const double* out4, const int* info2, const double* vec, int n, double* host_mapped, gpmc_stream_t stream) {
    return publish_host<double>(out4, info2, vec, n, host_mapped, S_(stream));
}
int gpmc_publish_host_f32(const double* out4, const int* info2, const float* vec, int n, double* host_mapped, gpmc_stream_t stream) {
    return publish_host<float>(out4, info2, vec, n, host_mapped, S_(stream));
}

/* ---- a caller's launch sequence, recorded once and replayed (the SDS proposal: gpmc/sds.py) ---- */
int gpmc_seq_begin(gpmc_seq** out, gpmc_stream_t stream) {
    GPMC_ARG(out != nullptr, 1);
    GPMC_ARG(stream != nullptr, 2);                 /* the NULL stream cannot be captured */
    Seq* q = new (std::nothrow) Seq();
    if (!q) return fail(-1, "gpmc_seq_begin: out of host memory");
    q->stream = S_(stream);
    hipError_t e = hipStreamBeginCapture(q->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { delete q; return fail((int)e, "hipStreamBeginCapture: %s", hipGetErrorString(e)); }
    q->capturing = true;
    *out = reinterpret_cast<gpmc_seq*>(q);
    return 0;
}
int gpmc_seq_end(gpmc_seq* h) {
    Seq* q = reinterpret_cast<Seq*>(h);
    GPMC_ARG(q != nullptr && q->capturing, 1);
    q->capturing = false;
    hipError_t e = hipStreamEndCapture(q->stream, &q->graph);
    if (e != hipSuccess) { q->graph = nullptr; return fail((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
    e = hipGraphInstantiate(&q->exec, q->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { q->exec = nullptr; return fail((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    return 0;
}
int gpmc_seq_launch(gpmc_seq* h, gpmc_stream_t stream) {
    Seq* q = reinterpret_cast<Seq*>(h);
    GPMC_ARG(q != nullptr && q->exec != nullptr, 1);
    GPMC_HIP(hipGraphLaunch(q->exec, S_(stream)));
    return 0;
}
int gpmc_seq_retire(gpmc_seq* h) { return seq_retire(reinterpret_cast<Seq*>(h)); }

int gpmc_reset_tri_f64(double* R, int np_, int ld, const double* diag_dev, gpmc_stream_t stream) {
    return reset_tri<double>(R, np_, ld, diag_dev, S_(stream));
}
int gpmc_reset_tri_f32(float* R, int np_, int ld, const float* diag_dev, gpmc_stream_t stream) {
    return reset_tri<float>(R, np_, ld, diag_dev, S_(stream));
}
int gpmc_potrf_rhs_f64(double* W, int mp, int np_, int ldw, int ntri, int rhs_rows, int* info_dev, void* scratch, gpmc_stream_t stream) {
    return potrf<double>(W, mp, np_, ldw, info_dev, scratch, S_(stream), ntri, nullptr, rhs_rows);
}
int gpmc_potrf_rhs_f32(float* W, int mp, int np_, int ldw, int ntri, int rhs_rows, int* info_dev, void* scratch, gpmc_stream_t stream) {
    return potrf<float>(W, mp, np_, ldw, info_dev, scratch, S_(stream), ntri, nullptr, rhs_rows);
}
int gpmc_syrk_tri_f64(const double* P, int ldp, double diag_add, double* C, int n, int np_, int ldc, gpmc_stream_t stream) {
    return syrk_tri<double>(P, ldp, diag_add, C, n, np_, ldc, S_(stream));
}
int gpmc_syrk_tri_f32(const float* P, int ldp, float diag_add, float* C, int n, int np_, int ldc, gpmc_stream_t stream) {
    return syrk_tri<float>(P, ldp, diag_add, C, n, np_, ldc, S_(stream));
}
int gpmc_syrk_tri2_f64(const double* P, int ldp, double diag_add, double* C, double* C2, int n, int np_, int ldc, gpmc_stream_t stream) {
    GPMC_ARG(C2 != nullptr && C2 != C, 5);
    return syrk_tri<double>(P, ldp, diag_add, C, n, np_, ldc, S_(stream), C2);
}
int gpmc_syrk_tri2_f32(const float* P, int ldp, float diag_add, float* C, float* C2, int n, int np_, int ldc, gpmc_stream_t stream) {
    GPMC_ARG(C2 != nullptr && C2 != C, 5);
    return syrk_tri<float>(P, ldp, diag_add, C, n, np_, ldc, S_(stream), C2);
}
int gpmc_logdet_quad_f64(const double* W, int n, int ldw, const double* alpha_dev, const int* info_dev,
                         double* out_dev, gpmc_stream_t stream) {
    return logdet_quad<double>(W, n, ldw, alpha_dev, info_dev, out_dev, S_(stream));
}
int gpmc_logdet_quad_f32(const float* W, int n, int ldw, const float* alpha_dev, const int* info_dev,
                         double* out_dev, gpmc_stream_t stream) {
    return logdet_quad<float>(W, n, ldw, alpha_dev, info_dev, out_dev, S_(stream));
}
#define PACK_API(SFX, T)                                                                                          \
    int gpmc_pack_lower_##SFX(const T* A, int n, int lda, T jitter, T* W, int np_, int ldw, gpmc_stream_t st) {   \
        return pack_lower<T>(A, n, lda, jitter, W, np_, ldw, S_(st));                                             \
    }                                                                                                             \
    int gpmc_pack_rows_##SFX(const T* B, int nrhs, int n, int ldb, T* W, int mp, int np_, int ldw,                \
                             gpmc_stream_t st) {                                                                  \
        return pack_rows<T>(B, nrhs, n, ldb, W, mp, np_, ldw, S_(st));                                            \
    }                                                                                                             \
    int gpmc_unpack_lower_##SFX(const T* W, int np_, int ldw, T* L, int n, int ldl, gpmc_stream_t st) {           \
        return unpack_lower<T>(W, np_, ldw, L, n, ldl, S_(st));                                                   \
    }                                                                                                             \
    int gpmc_unpack_rows_##SFX(const T* W, int np_, int ldw, T* B, int nrhs, int n, int ldb, gpmc_stream_t st) {  \
        return unpack_rows<T>(W, np_, ldw, B, nrhs, n, ldb, S_(st));                                              \
    }                                                                                                             \
    int gpmc_syrk_diff_##SFX(const T* D, int ldd, const T* P, int kp, int ldp, T diag_add, T* C, int n, int np_,  \
                             int ldc, gpmc_stream_t st) {                                                         \
        return syrk_diff<T>(D, ldd, P, kp, ldp, diag_add, C, n, np_, ldc, S_(st));                                \
    }                                                                                                             \
    int gpmc_symv_lower_##SFX(const T* W, int n, int ldw, const T* x, T* y, gpmc_stream_t st) {                   \
        return symv_lower<T>(W, n, ldw, x, y, S_(st));                                                            \
    }                                                                                                             \
    int gpmc_trmv_lower_##SFX(const T* W, int n, int ldw, const T* x, const T* y0, T* y, gpmc_stream_t st) {      \
        return trmv_lower<T>(W, n, ldw, x, y0, y, S_(st));                                                        \
    }
int gpmc_trmv_upper_f64(const double* W, int n, int ldw, const double* x, const double* y0, double alpha, double* y,
                        gpmc_stream_t st) {
    return trmv_upper<double>(W, n, ldw, x, y0, alpha, y, S_(st));
}
int gpmc_trmv_upper_f32(const float* W, int n, int ldw, const float* x, const float* y0, double alpha, float* y,
                        gpmc_stream_t st) {
    return trmv_upper<float>(W, n, ldw, x, y0, alpha, y, S_(st));
}
PACK_API(f64, double)
PACK_API(f32, float)

// ------------------------------------------------------------------------------------- plan
size_t gpmc_plan_workspace_bytes(int n, int d, int elem_size) {
    if (n <= 0 || d <= 0 || (elem_size != 4 && elem_size != 8)) return 0;
    return plan_layout(n, d, elem_size).total;
}

int gpmc_plan_create(gpmc_plan** out, int kernel, int n, int d, int elem_size, void* workspace,
                     size_t workspace_bytes, gpmc_stream_t stream) {
    GPMC_ARG(out != nullptr, 1);
    GPMC_ARG(kernel == GPMC_KERNEL_RBF || kernel == GPMC_KERNEL_MATERN52_ARD, 2);
    GPMC_ARG(n > 0, 3);
    GPMC_ARG(d > 0 && d <= PLAN_MAXD, 4);
    GPMC_ARG(elem_size == 4 || elem_size == 8, 5);
    GPMC_ARG(workspace != nullptr && reinterpret_cast<uintptr_t>(workspace) % 256 == 0, 6);
    PlanLayout L = plan_layout(n, d, elem_size);
    GPMC_ARG(workspace_bytes >= L.total, 7);
    gpmc_plan* p = new (std::nothrow) gpmc_plan();
    if (!p) return fail(-1, "gpmc_plan_create: out of host memory");
    p->kernel = kernel; p->n = n; p->d = d; p->es = elem_size; p->L = L;
    p->ws = reinterpret_cast<char*>(workspace);
    p->stream = S_(stream);
    p->recorded = false; p->evals = 0; p->have_data = false;
    p->ordered = false;
    p->ring_dev = nullptr; p->seq_dev = nullptr; p->execs = 0; p->resync = false;
    memset(p->cur_stage, 0, sizeof(p->cur_stage));
    p->ev0 = p->ev1 = nullptr; p->eager_ms = p->replay_ms = 0.f; p->record_attempts = 0; p->recorded_form = -1;
    memset(p->slot_busy, 0, sizeof(p->slot_busy));
    memset(p->slot_ev, 0, sizeof(p->slot_ev));
    {   // the plan's own look-ahead stream and one event per cross-stream edge of its schedule (2 per 128 columns
        // bounds every schedule potrf() builds), created here, destroyed in gpmc_plan_destroy
        int rc = p->sched.init(p->stream);
        if (!rc) rc = p->sched.reserve_events((size_t)(2 * (L.np_ / GPMC_PAD) + 8));
        if (rc) { (void)p->sched.destroy(); delete p; return rc; }
    }
    p->hyp_slot_bytes = (size_t)(2 + PLAN_MAXD + 2) * sizeof(double);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p->host_pin),
                                 PLAN_SLOTS * (p->hyp_slot_bytes + 5 * sizeof(double)), hipHostMallocDefault);
    if (e != hipSuccess) { (void)p->sched.destroy(); delete p; return fail((int)e, "hipHostMalloc: %s", hipGetErrorString(e)); }
    for (int i = 0; i < PLAN_SLOTS; ++i)      // result stamps: no execution yet
        (reinterpret_cast<double*>(reinterpret_cast<char*>(p->host_pin) + PLAN_SLOTS * p->hyp_slot_bytes))[4 * PLAN_SLOTS + i] = -1.0;
    {   // the reduction kernel writes each result straight into its pinned slot
        void* dres = nullptr;
        e = hipHostGetDevicePointer(&dres, reinterpret_cast<char*>(p->host_pin) + PLAN_SLOTS * p->hyp_slot_bytes, 0);
        if (e != hipSuccess) { (void)hipHostFree(p->host_pin); (void)p->sched.destroy(); delete p; return fail((int)e, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
        p->host_res_dev = reinterpret_cast<double*>(dres);
        e = hipHostGetDevicePointer(&dres, p->host_pin, 0);
        if (e == hipSuccess) {
            p->ring_dev = dres;
            e = hipMalloc(reinterpret_cast<void**>(&p->seq_dev), 256);
            if (e == hipSuccess) e = hipMemsetAsync(p->seq_dev, 0, 256, p->stream);
        }
        if (e != hipSuccess) { if (p->seq_dev) (void)hipFree(p->seq_dev); (void)hipHostFree(p->host_pin); (void)p->sched.destroy(); delete p; return fail((int)e, "parameter ring: %s", hipGetErrorString(e)); }
    }
    // zero the riding rows once: only row np_ is ever rewritten (with g), the others stay zero
    e = hipMemsetAsync(p->ws + L.off_W + (size_t)L.np_ * L.ld * elem_size, 0,
                       (size_t)(L.mp - L.np_) * L.ld * elem_size, p->stream);
    if (e != hipSuccess) { (void)hipFree(p->seq_dev); (void)hipHostFree(p->host_pin); (void)p->sched.destroy(); delete p; return fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e)); }
    ++g_live_ws[p->ws];
    p->ordered = plans_must_order();
    *out = p;
    return 0;
}

int gpmc_plan_set_data(gpmc_plan* p, const void* X_dev, int ldx, const void* g_dev) {
    GPMC_ARG(p != nullptr, 1);
    GPMC_ARG(X_dev != nullptr && ldx >= p->d, 2);
    GPMC_ARG(g_dev != nullptr, 4);
    GPMC_HIP(hipMemcpy2DAsync(p->ws + p->L.off_X, (size_t)p->d * p->es, X_dev, (size_t)ldx * p->es,
                              (size_t)p->d * p->es, p->n, hipMemcpyDeviceToDevice, p->stream));
    GPMC_HIP(hipMemcpyAsync(p->ws + p->L.off_g, g_dev, (size_t)p->n * p->es, hipMemcpyDeviceToDevice, p->stream));
    p->have_data = true;
    return 0;
}

int gpmc_plan_logml_async(gpmc_plan* p, const double* kparams_host, int nk, int slot) {
    GPMC_ARG(p != nullptr && p->have_data, 1);
    const int need = 2 + (p->kernel == GPMC_KERNEL_RBF ? 1 : p->d);
    GPMC_ARG(kparams_host != nullptr && nk == need, 3);
    GPMC_ARG(slot >= 0 && slot < PLAN_SLOTS, 4);
    // a slot's result lands asynchronously, and the number of busy slots bounds the executions in flight, which is what keeps
    // the parameter ring's blocks from being rewritten under a running evaluation
    if (p->slot_busy[slot])
        return fail(-4, "gpmc_plan_logml_async: slot %d still has an evaluation in flight (call gpmc_plan_result first)", slot);
    (reinterpret_cast<double*>(reinterpret_cast<char*>(p->host_pin) + PLAN_SLOTS * p->hyp_slot_bytes))[4 * PLAN_SLOTS + slot] = -1.0;   // the slot's stamp: no result yet
    // staged: the parameters, then (at element 2 + PLAN_MAXD) the index of the result slot, which the reduction kernel
    // reads back to find where in pinned memory this evaluation's result belongs
    if (p->es == 8) {
        double* d = reinterpret_cast<double*>(p->cur_stage);
        memcpy(d, kparams_host, (size_t)nk * sizeof(double));
        d[2 + PLAN_MAXD] = (double)slot;
    } else {
        float* f = reinterpret_cast<float*>(p->cur_stage);
        for (int i = 0; i < nk; ++i) f[i] = (float)kparams_host[i];
        f[2 + PLAN_MAXD] = (float)slot;
    }
    int rc = 0;
    if (p->resync) {      // an earlier execution failed part-way: executions launched and executions completed must agree
        GPMC_HIP(hipStreamSynchronize(p->stream));
        if (p->sched.panel) GPMC_HIP(hipStreamSynchronize(p->sched.panel));
        GPMC_HIP(hipMemcpy(p->seq_dev, &p->execs, sizeof(unsigned), hipMemcpyHostToDevice));
        p->resync = false;
    }
    if (p->ordered != plans_must_order()) {
        // another plan appeared (or the last other one went): the sequence recorded in the old mode is retired and this plan
        // starts over -- two eager evaluations (the second one the yardstick), then a new recording
        if (p->recorded) {
            GPMC_HIP(hipStreamSynchronize(p->stream));
            if (p->sched.panel) GPMC_HIP(hipStreamSynchronize(p->sched.panel));
            rc = p->sched.drop_ops();
            if (rc) return rc;
            p->recorded = false;
        }
        p->evals = 0;
        p->ordered = plans_must_order();
    }
    const bool graph_ok = g_opt.use_graph && p->stream != nullptr;   // the NULL stream cannot be captured
    if (!graph_ok || p->evals < 2) {
        // eager launches: the first evaluation always (it sets the kernels' attributes, which a capture would record);
        // the second one timed -- the yardstick for the recorded sequence below
        const bool timeit = graph_ok && p->evals == 1;
        if (timeit) {
            if (!p->ev0) { GPMC_HIP(hipEventCreate(&p->ev0)); GPMC_HIP(hipEventCreate(&p->ev1)); }
            GPMC_HIP(hipEventRecord(p->ev0, p->stream));
        }
        p->sched.mode = Sched::EAGER;
        stage_params(p);
        rc = plan_enqueue(p, p->stream);
        if (rc) { p->resync = true; return rc; }
        if (timeit) {
            GPMC_HIP(hipEventRecord(p->ev1, p->stream));
            GPMC_HIP(hipEventSynchronize(p->ev1));
            GPMC_HIP(hipEventElapsedTime(&p->eager_ms, p->ev0, p->ev1));
        }
    } else if (p->recorded) {
        stage_params(p);
        rc = p->sched.replay();
        if (rc) { p->resync = true; return rc; }
    } else {
        rc = record_sequence(p);      // records AND replays once (a real evaluation of the current parameters)
        if (rc) { p->resync = true; return rc; }
    }
    p->evals++;
    p->slot_busy[slot] = 1;
    p->slot_exec[slot] = (p->execs - 1u) & 0xfffffu;
    // (no result copy: the reduction kernel has written the pinned slot itself)
    if (!p->slot_ev[slot]) GPMC_HIP(hipEventCreateWithFlags(&p->slot_ev[slot], hipEventDisableTiming));
    GPMC_HIP(hipEventRecord(p->slot_ev[slot], p->stream));
    return 0;
}

/* diagnostic only (gpmc.h, diagnostics): churn graph objects -- the destroy patterns that aborted in round 1.
   form 0: the plan's segmented sequence (linear graphs);  form 1: round 1's single fork/join graph (kept so that the
   runtime defect stays reproducible: on HIP runtime 7.0.51831 form 1 + device_sync aborts, on 7.2.26015 it does not).
   Per iteration: record, replay, stream sync, optionally hipDeviceSynchronize, destroy every exec. */
int gpmc_debug_graph_churn(gpmc_plan* p, int form, int device_sync, int iters) {
    GPMC_ARG(p != nullptr && p->have_data && p->stream != nullptr && p->evals > 1, 1);
    GPMC_ARG(iters > 0 && iters <= 4096, 4);
    if (form && !forkjoin_allowed()) return fail(-5, "gpmc_debug_graph_churn: the graveyard of parked fork/join execs is full");
    GPMC_HIP(hipStreamSynchronize(p->stream));
    (void)p->sched.drop_ops();
    p->recorded = false;
    p->resync = true;     // (the executions below consume sequence numbers without parameter blocks of their own)
    for (int i = 0; i < iters; ++i) {
        int rc = p->sched.begin(form ? Sched::FORKJOIN : Sched::SEGMENTS);
        if (rc) return rc;
        rc = plan_enqueue(p, p->sched.main);
        const int rc2 = p->sched.end();
        if (rc || rc2) { (void)p->sched.drop_ops(); return rc ? rc : rc2; }
        rc = p->sched.replay();
        if (rc) return rc;
        GPMC_HIP(hipStreamSynchronize(p->stream));
        if (device_sync) GPMC_HIP(hipDeviceSynchronize());
        rc = p->sched.drop_ops();
        if (rc) return rc;
    }
    return 0;
}

/* diagnostic only: number of ops / graph launches in the plan's recorded sequence (0 before the second evaluation) */
int gpmc_debug_plan_record_info(gpmc_plan* p, double* out4) {   /* eager ms, replay ms, attempts, form */
    GPMC_ARG(p != nullptr && out4 != nullptr, 1);
    out4[0] = p->eager_ms; out4[1] = p->replay_ms; out4[2] = p->record_attempts; out4[3] = p->recorded_form;
    return 0;
}
int gpmc_debug_plan_ops(gpmc_plan* p, int* n_ops, int* n_graphs) {
    GPMC_ARG(p != nullptr, 1);
    int g = 0;
    for (const SeqOp& o : p->sched.ops) g += (o.kind == 0);
    if (n_ops) *n_ops = (int)p->sched.ops.size();
    if (n_graphs) *n_graphs = g;
    return 0;
}

int gpmc_graveyard_stats(int* execs, size_t* nodes) {
    graveyard_stats(execs, nodes);
    return 0;
}

int gpmc_plan_result(gpmc_plan* p, int slot, double* out_host) {
    GPMC_ARG(p != nullptr, 1);
    GPMC_ARG(slot >= 0 && slot < PLAN_SLOTS, 2);
    GPMC_ARG(out_host != nullptr, 3);
    const double* hbase = reinterpret_cast<const double*>(reinterpret_cast<char*>(p->host_pin) + PLAN_SLOTS * p->hyp_slot_bytes);
    if (p->slot_busy[slot]) {                        // wait for THIS slot's evaluation only: later ones keep running
        // the reduction writes the execution's number behind the results as its last store: poll that word for a while (a
        // chain reads every result before it can go on, and the event path costs several microseconds after the kernel has
        // ended), then fall back to the event
        const volatile double* stamp = hbase + 4 * PLAN_SLOTS + slot;
        const double want = (double)p->slot_exec[slot];
        bool seen = false;
        if (g_opt.result_poll_us > 0) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0;; ++it) {
                if (*stamp == want) { seen = true; break; }
                if ((it & 63u) == 63u &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > g_opt.result_poll_us)
                    break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) GPMC_HIP(hipEventSynchronize(p->slot_ev[slot]));
        p->slot_busy[slot] = 0;
    }
    const double* hres = hbase + 4 * slot;
    for (int i = 0; i < 4; ++i) out_host[i] = hres[i];
    if (out_host[3] == (double)GPMC_RING_OUT_OF_STEP) {
        p->resync = true;
        return fail(-6, "gpmc_plan_result: the evaluation read the parameter block of another execution (host and device "
                        "sequence numbers differ); the next evaluation resynchronises -- repeat this one");
    }
    return 0;
}

int gpmc_plan_workspace(gpmc_plan* p, void** W, int* mp, int* np_, int* ld) {
    GPMC_ARG(p != nullptr, 1);
    if (W) *W = p->ws + p->L.off_W;
    if (mp) *mp = p->L.mp;
    if (np_) *np_ = p->L.np_;
    if (ld) *ld = p->L.ld;
    return 0;
}

int gpmc_plan_handover_in_launch(const gpmc_plan* p) { return (p != nullptr && !plans_must_order()) ? 1 : 0; }

int gpmc_plan_destroy(gpmc_plan* p) {
    if (!p) return 0;
    {
        auto it = g_live_ws.find(p->ws);
        if (it != g_live_ws.end() && --it->second == 0) g_live_ws.erase(it);
    }
    (void)hipStreamSynchronize(p->stream);
    if (p->sched.panel) (void)hipStreamSynchronize(p->sched.panel);
    (void)p->sched.destroy();     // retires the recorded execs (drop_ops), then the events and the panel stream
    for (hipEvent_t e : p->slot_ev) if (e) (void)hipEventDestroy(e);
    if (p->ev0) { (void)hipEventDestroy(p->ev0); (void)hipEventDestroy(p->ev1); }
    if (p->host_pin) (void)hipHostFree(p->host_pin);
    if (p->seq_dev) (void)hipFree(p->seq_dev);
    delete p;
    return 0;
}

}  // extern "C"
