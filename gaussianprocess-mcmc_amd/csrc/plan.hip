// plan.hip -- fused log-marginal plan + the extern "C" surface of libgpmc.so.
//
// One plan = one (kernel, n, d, dtype) problem bound to caller-owned device memory.  An evaluation is
//   H2D(16..80 B of kernel parameters) -> [ kbuild(lower, S on the diagonal, identity padding)
//   -> copy g into the riding row -> blocked Cholesky -> log-det / quadratic-form reduction ] -> D2H(32 B)
// The bracketed part is a static launch sequence (~400 launches at N=8192, two streams for the
// look-ahead); it is captured once into a hipGraph and replayed, so the host cost per evaluation is
// one graph launch instead of hundreds of kernel launches.
#include <cmath>
#include <new>

#include "gpmc_common.h"

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
    L.off_hyp = take((size_t)(2 + PLAN_MAXD) * es);
    L.off_out = take(4 * sizeof(double));
    L.off_info = take(sizeof(int));
    L.off_scratch = take((size_t)(L.np_ / NB_IN + 1) * (1024 + 4096 + 4096) * sizeof(double));
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
    hipGraph_t graph;
    hipGraphExec_t exec;
    int evals;
    bool have_data;
    // graph validation (see gpmc_plan_logml_async): wall time of the eager first evaluation, and whether the
    // instantiated graph has been checked against it / abandoned
    hipEvent_t ev0, ev1;
    float eager_ms;
    bool graph_checked, graph_off;
    int graph_retries;
};

namespace gpmc {

template <typename T>
static int plan_enqueue_body(gpmc_plan* p, hipStream_t s) {
    const PlanLayout& L = p->L;
    T* W = reinterpret_cast<T*>(p->ws + L.off_W);
    const T* X = reinterpret_cast<const T*>(p->ws + L.off_X);
    const T* g = reinterpret_cast<const T*>(p->ws + L.off_g);
    const T* hyp = reinterpret_cast<const T*>(p->ws + L.off_hyp);
    double* out = reinterpret_cast<double*>(p->ws + L.off_out);
    int* info = reinterpret_cast<int*>(p->ws + L.off_info);
    int rc = kbuild<T>(p->kernel, X, p->n, p->d, p->d, nullptr, 0, 0, hyp, W, L.np_, L.np_, L.ld,
                       GPMC_KB_LOWER | GPMC_KB_PAD_IDENTITY, s);
    if (rc) return rc;
    GPMC_HIP(hipMemcpyAsync(W + (size_t)L.np_ * L.ld, g, (size_t)p->n * sizeof(T), hipMemcpyDeviceToDevice, s));
    rc = potrf<T>(W, L.mp, L.np_, L.ld, info, p->ws + L.off_scratch, s);
    if (rc) return rc;
    return logdet_quad<T>(W, p->n, L.ld, W + (size_t)L.np_ * L.ld, info, out, s);
}

static int plan_enqueue(gpmc_plan* p, hipStream_t s) {
    return p->es == 8 ? plan_enqueue_body<double>(p, s) : plan_enqueue_body<float>(p, s);
}

// Retiring a graph exec.  ROCm 7.2: destroying an exec whose launch sequence spans two streams (the look-ahead
// schedule) after a device-wide synchronisation (hipDeviceSynchronize, torch.cuda.synchronize()) has touched its
// internal streams corrupts the runtime's heap -- scripts/graph_stress.py modes 3 and 7 abort within a few cycles,
// single-stream graphs and destroys not preceded by a device sync (modes 0/1/4) never do.  One more replay followed
// by a stream sync right before the destroy makes it safe (modes 8/9, 480 cycles clean), so every exec leaves this
// way.  The replay recomputes the last evaluation into the plan's workspace, which must therefore still be valid.
static void retire_exec(gpmc_plan* p) {
    if (!p->exec) return;
    if (hipGraphLaunch(p->exec, p->stream) == hipSuccess) (void)hipStreamSynchronize(p->stream);
    (void)hipGraphExecDestroy(p->exec);
    p->exec = nullptr;
}

}  // namespace gpmc

using namespace gpmc;

extern "C" {

int gpmc_version(void) { return GPMC_VERSION; }
const char* gpmc_last_error(void) { return g_err; }
int gpmc_pad(int n) { return round_up(n, GPMC_PAD); }
size_t gpmc_potrf_scratch_bytes(int np_) { return (size_t)(np_ / NB_IN + 1) * (1024 + 4096 + 4096) * sizeof(double); }

int gpmc_set_option(const char* name, int value) {
    if (!name) return fail(-1, "gpmc_set_option: null name");
    if (!strcmp(name, "outer_nb")) { if (value != 0 && (value < 128 || value % 128)) return fail(-2, "outer_nb must be 0 (auto) or a multiple of 128"); g_opt.outer_nb = value; }
    else if (!strcmp(name, "lookahead")) g_opt.lookahead = value < 0 ? -1 : (value ? 1 : 0);
    else if (!strcmp(name, "use_graph")) g_opt.use_graph = value ? 1 : 0;
    else if (!strcmp(name, "small_tile_m")) g_opt.small_tile_m = value;
    else if (!strcmp(name, "diag_impl")) g_opt.diag_impl = value;
    else if (!strcmp(name, "panel_impl")) g_opt.panel_impl = value;
    else if (!strcmp(name, "gemm_impl")) g_opt.gemm_impl = value;
    else if (!strcmp(name, "gemm_desync")) g_opt.gemm_desync = value;
    else if (!strcmp(name, "gemm_xcd_map")) g_opt.gemm_xcd_map = value;
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
    else if (!strcmp(name, "gemm_persistent")) g_opt.gemm_persistent = value;
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
    if (!strcmp(name, "gemm_desync")) return g_opt.gemm_desync;
    if (!strcmp(name, "gemm_xcd_map")) return g_opt.gemm_xcd_map;
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
/* diagnostic only (not in gpmc.h): phase stamps of the fused panel kernel */
int gpmc_debug_panel_stamps(double* W, int ld, int mp, int k0, int c0, int* info_dev, void* scratch, int np_,
                            unsigned long long* dbg_dev, gpmc_stream_t stream) {
    double* lpark = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + (size_t)(np_ / NB_IN + 1) * 1024 * sizeof(double));
    return debug_panel_stamps(W, ld, mp, k0, c0, info_dev, lpark, dbg_dev, reinterpret_cast<hipStream_t>(stream));
}
int gpmc_profile_begin(void) { return profile_begin(); }
int gpmc_profile_read(double* out3) {
    if (!out3) return fail(-1, "gpmc_profile_read: null out");
    return profile_read(out3);
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
int gpmc_syrk_tri_f64(const double* P, int ldp, double diag_add, double* C, int n, int np_, int ldc, gpmc_stream_t stream) {
    return syrk_tri<double>(P, ldp, diag_add, C, n, np_, ldc, S_(stream));
}
int gpmc_syrk_tri_f32(const float* P, int ldp, float diag_add, float* C, int n, int np_, int ldc, gpmc_stream_t stream) {
    return syrk_tri<float>(P, ldp, diag_add, C, n, np_, ldc, S_(stream));
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
    p->graph = nullptr; p->exec = nullptr; p->evals = 0; p->have_data = false;
    p->ev0 = p->ev1 = nullptr; p->eager_ms = 0.f; p->graph_checked = false; p->graph_off = false; p->graph_retries = 0;
    p->hyp_slot_bytes = (size_t)(2 + PLAN_MAXD) * sizeof(double);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p->host_pin),
                                 PLAN_SLOTS * (p->hyp_slot_bytes + 4 * sizeof(double)), hipHostMallocDefault);
    if (e != hipSuccess) { delete p; return fail((int)e, "hipHostMalloc: %s", hipGetErrorString(e)); }
    // zero the riding rows once: only row np_ is ever rewritten (with g), the others stay zero
    e = hipMemsetAsync(p->ws + L.off_W + (size_t)L.np_ * L.ld * elem_size, 0,
                       (size_t)(L.mp - L.np_) * L.ld * elem_size, p->stream);
    if (e != hipSuccess) { (void)hipHostFree(p->host_pin); delete p; return fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e)); }
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
    char* hslot = reinterpret_cast<char*>(p->host_pin) + (size_t)slot * p->hyp_slot_bytes;
    if (p->es == 8) {
        memcpy(hslot, kparams_host, (size_t)nk * sizeof(double));
    } else {
        float* f = reinterpret_cast<float*>(hslot);
        for (int i = 0; i < nk; ++i) f[i] = (float)kparams_host[i];
    }
    GPMC_HIP(hipMemcpyAsync(p->ws + p->L.off_hyp, hslot, (size_t)nk * p->es, hipMemcpyHostToDevice, p->stream));
    int rc = 0;
    const bool graph_ok = g_opt.use_graph && !p->graph_off && p->stream != nullptr;   // the NULL stream cannot be captured
    if (!graph_ok || p->evals == 0) {
        // first evaluation runs eagerly (creates streams, sets attributes) and is timed: the reference for the
        // graph check below
        const bool timeit = graph_ok && p->evals == 0;
        if (timeit) {
            if (!p->ev0) { GPMC_HIP(hipEventCreate(&p->ev0)); GPMC_HIP(hipEventCreate(&p->ev1)); }
            GPMC_HIP(hipEventRecord(p->ev0, p->stream));
        }
        rc = plan_enqueue(p, p->stream);
        if (rc) return rc;
        if (timeit) {
            GPMC_HIP(hipEventRecord(p->ev1, p->stream));
            GPMC_HIP(hipEventSynchronize(p->ev1));
            GPMC_HIP(hipEventElapsedTime(&p->eager_ms, p->ev0, p->ev1));
        }
    } else {
        if (!p->graph) {
            GPMC_HIP(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
            rc = plan_enqueue(p, p->stream);
            hipError_t e = hipStreamEndCapture(p->stream, &p->graph);
            if (rc) return rc;
            if (e != hipSuccess) return fail((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
        }
        if (!p->exec) GPMC_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
        if (p->graph_checked) {
            GPMC_HIP(hipGraphLaunch(p->exec, p->stream));
        } else {
            // Validate the instantiation once.  Observed on ROCm 7.2: hipGraphInstantiate does not always map the
            // two-stream look-ahead schedule the same way -- about one exec in ten replays it with the branches
            // serialised (13.5 ms instead of 6.7 ms at N=8192), one in six at 7.05 ms; each exec is stable.  So:
            // time one replay, destroy, re-instantiate, and accept the first exec whose replay is within 2% of the
            // fastest seen so far and not slower than the eager run; after 5 attempts fall back to eager launches
            // if the last one is anomalously slow.  Every replay is a real evaluation of the current parameters,
            // so whichever attempt is accepted leaves the right result behind.  (No device-wide sync happens
            // between a replay and its destroy here, the safe pattern of retire_exec's note.)
            float seen_ms = 0.f, ms = 0.f;
            for (int attempt = 0; attempt < 5; ++attempt) {
                if (!p->exec) GPMC_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
                GPMC_HIP(hipEventRecord(p->ev0, p->stream));
                GPMC_HIP(hipGraphLaunch(p->exec, p->stream));
                GPMC_HIP(hipEventRecord(p->ev1, p->stream));
                GPMC_HIP(hipEventSynchronize(p->ev1));
                GPMC_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
                p->graph_retries = attempt;
                const bool ok = attempt > 0 && ms <= 1.02f * seen_ms && ms <= 1.25f * p->eager_ms + 0.05f;
                if (ok || attempt == 4) break;
                seen_ms = (attempt == 0 || ms < seen_ms) ? ms : seen_ms;
                GPMC_HIP(hipGraphExecDestroy(p->exec));
                p->exec = nullptr;
            }
            if (ms > 1.25f * p->eager_ms + 0.05f) {
                retire_exec(p);
                p->graph_off = true;
            }
            p->graph_checked = true;
        }
    }
    p->evals++;
    double* hres = reinterpret_cast<double*>(reinterpret_cast<char*>(p->host_pin) + PLAN_SLOTS * p->hyp_slot_bytes) + 4 * slot;
    GPMC_HIP(hipMemcpyAsync(hres, p->ws + p->L.off_out, 4 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    return 0;
}

/* diagnostic only (not in gpmc.h): churn graph objects to find which runtime pattern is unsafe.
   mode 0: instantiate / launch / sync / destroy exec, same captured graph;  mode 1: capture + instantiate / launch /
   sync / destroy exec + graph;  mode 2: instantiate + launch, keep every exec alive, destroy all at the end;
   mode 3: as 0 with a device-wide sync before the destroy;  mode 4: as 0 but timed with an event pair recorded
   around the launch and waited on with hipEventSynchronize (what the plan's graph check used to do);
   mode 6: as 2 (keep every exec alive, destroy all at the end) with a device-wide sync after every launch;
   mode 7: as 1 (capture and destroy the graph every iteration too) with a device-wide sync before the destroys;
   mode 8: as 3, then one more launch + stream sync (no device sync) right before the destroy;
   mode 9: as 7 with that extra launch + stream sync before the destroys. */
int gpmc_debug_graph_churn(gpmc_plan* p, int mode, int iters) {
    GPMC_ARG(p != nullptr && p->have_data && p->stream != nullptr, 1);
    GPMC_ARG(iters > 0 && iters <= 4096, 3);
    hipGraph_t g = nullptr;
    int rc = 0;
    auto capture = [&](hipGraph_t* out) -> int {
        GPMC_HIP(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
        int r = plan_enqueue(p, p->stream);
        hipError_t e = hipStreamEndCapture(p->stream, out);
        if (r) return r;
        if (e != hipSuccess) return fail((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
        return 0;
    };
    if (mode != 1 && mode != 7 && mode != 9 && (rc = capture(&g))) return rc;
    hipGraphExec_t* keep = (mode == 2 || mode == 6) ? new hipGraphExec_t[iters]() : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (mode == 4) { GPMC_HIP(hipEventCreate(&e0)); GPMC_HIP(hipEventCreate(&e1)); }
    for (int i = 0; i < iters; ++i) {
        hipGraphExec_t x = nullptr;
        if ((mode == 1 || mode == 7 || mode == 9) && (rc = capture(&g))) return rc;
        GPMC_HIP(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
        if (mode == 4) {
            float ms = 0.f;
            GPMC_HIP(hipEventRecord(e0, p->stream));
            GPMC_HIP(hipGraphLaunch(x, p->stream));
            GPMC_HIP(hipEventRecord(e1, p->stream));
            GPMC_HIP(hipEventSynchronize(e1));
            GPMC_HIP(hipEventElapsedTime(&ms, e0, e1));
        } else {
            GPMC_HIP(hipGraphLaunch(x, p->stream));
            GPMC_HIP(hipStreamSynchronize(p->stream));
        }
        if (mode == 3 || mode == 6 || mode == 7 || mode == 8 || mode == 9) GPMC_HIP(hipDeviceSynchronize());
        if (mode == 8 || mode == 9) {
            GPMC_HIP(hipGraphLaunch(x, p->stream));
            GPMC_HIP(hipStreamSynchronize(p->stream));
        }
        if (keep) keep[i] = x; else GPMC_HIP(hipGraphExecDestroy(x));
        if (mode == 1 || mode == 7 || mode == 9) { GPMC_HIP(hipGraphDestroy(g)); g = nullptr; }
    }
    if (keep) { for (int i = 0; i < iters; ++i) GPMC_HIP(hipGraphExecDestroy(keep[i])); delete[] keep; }
    if (g) GPMC_HIP(hipGraphDestroy(g));
    if (e0) { GPMC_HIP(hipEventDestroy(e0)); GPMC_HIP(hipEventDestroy(e1)); }
    return 0;
}

int gpmc_plan_result(gpmc_plan* p, int slot, double* out_host) {
    GPMC_ARG(p != nullptr, 1);
    GPMC_ARG(slot >= 0 && slot < PLAN_SLOTS, 2);
    GPMC_ARG(out_host != nullptr, 3);
    GPMC_HIP(hipStreamSynchronize(p->stream));
    const double* hres = reinterpret_cast<const double*>(reinterpret_cast<char*>(p->host_pin) + PLAN_SLOTS * p->hyp_slot_bytes) + 4 * slot;
    for (int i = 0; i < 4; ++i) out_host[i] = hres[i];
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

int gpmc_plan_destroy(gpmc_plan* p) {
    if (!p) return 0;
    (void)hipStreamSynchronize(p->stream);
    retire_exec(p);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->ev0) { (void)hipEventDestroy(p->ev0); (void)hipEventDestroy(p->ev1); }
    if (p->host_pin) (void)hipHostFree(p->host_pin);
    delete p;
    return 0;
}

}  // extern "C"
