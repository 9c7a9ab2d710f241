// sched.hip -- the launch schedule object (gpmc_common.h `Sched`): two streams, cross-stream edges, and the
// segmented record / replay form the plan uses instead of one fork/join hipGraph.
#include <mutex>

#include "gpmc_common.h"

namespace gpmc {

int Sched::init(hipStream_t main_stream) {
    main = main_stream;
    if (!panel) {
        int lo = 0, hi = 0;
        GPMC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // (highest priority; measured in round 3: default or lowest priority for this stream moves N = 8192 / 12288 by +-0.2 %)
        GPMC_HIP(hipStreamCreateWithPriority(&panel, hipStreamNonBlocking, hi));
        own_panel = true;
    }
    return reserve_events(16);
}

int Sched::reserve_events(size_t n) {
    while (events.size() < n) {
        hipEvent_t e;
        GPMC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        events.push_back(e);
    }
    return 0;
}

// End the capture segment of stream s (if it has launched anything, it becomes one linear graph op) and open the
// next one.  A stream that was not capturing yet (the panel stream before its first edge) just starts.
int Sched::cut(hipStream_t s) {
    const bool is_panel = (s == panel);
    bool& cap = is_panel ? cap_panel : cap_main;
    if (cap) {
        hipGraph_t g = nullptr;
        GPMC_HIP(hipStreamEndCapture(s, &g));
        cap = false;
        size_t nodes = 0;
        GPMC_HIP(hipGraphGetNodes(g, nullptr, &nodes));
        if (nodes > 0) {
            hipGraphExec_t x = nullptr;
            hipError_t e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
            if (e != hipSuccess) {
                (void)hipGraphDestroy(g);
                return fail((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e));
            }
            ops.push_back(SeqOp{0, is_panel ? 1 : 0, x, nullptr, nullptr});
        }
        GPMC_HIP(hipGraphDestroy(g));
    }
    GPMC_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    cap = true;
    return 0;
}

int Sched::edge_record(hipStream_t s, hipEvent_t* out) {
    if (mode == SEGMENTS) {
        if (ev_next == events.size()) return fail(-1, "Sched: event pool exhausted (%zu)", events.size());
        hipEvent_t e = events[ev_next++];
        int rc = cut(s);
        if (rc) return rc;
        ops.push_back(SeqOp{1, s == panel ? 1 : 0, nullptr, e, nullptr});
        *out = e;
        return 0;
    }
    hipEvent_t e = events[ev_next];
    ev_next = (ev_next + 1) % events.size();
    GPMC_HIP(hipEventRecord(e, s));
    *out = e;
    return 0;
}

int Sched::edge_wait(hipStream_t s, hipEvent_t e) {
    if (mode == SEGMENTS) {
        int rc = cut(s);
        if (rc) return rc;
        ops.push_back(SeqOp{2, s == panel ? 1 : 0, nullptr, e, nullptr});
        return 0;
    }
    if (mode == FORKJOIN && s == panel) forked = true;      // the capture now spans two streams
    GPMC_HIP(hipStreamWaitEvent(s, e, 0));
    return 0;
}

int Sched::begin(int mode_) {
    if (!ops.empty()) return fail(-1, "Sched::begin: a recorded sequence already exists");
    mode = mode_;
    ev_next = 0;
    forked = false;
    if (mode == EAGER) return 0;
    GPMC_HIP(hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal));
    cap_main = true;
    return 0;
}

int Sched::end() {
    if (mode == EAGER) return 0;
    // only an exec whose graph really spans two streams gets an internal branch stream from the runtime: a FORKJOIN
    // recording of a one-stream schedule (every problem the tile-DAG kernel takes whole) is a linear graph
    const bool forkjoin = (mode == FORKJOIN) && forked;
    int rc = 0;

    // close what is still open; a failed launch sequence must not leave a stream in capture mode
    for (int k = 0; k < 2; ++k) {
        hipStream_t s = k ? panel : main;
        bool& cap = k ? cap_panel : cap_main;
        if (!cap) continue;
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(s, &g);
        cap = false;
        if (e != hipSuccess) { if (!rc) rc = fail((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e)); continue; }
        size_t nodes = 0;
        e = hipGraphGetNodes(g, nullptr, &nodes);
        bool keep_graph = false;
        if (e == hipSuccess && nodes > 0) {
            hipGraphExec_t x = nullptr;
            e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
            if (e == hipSuccess) {
                ops.push_back(SeqOp{0, k, x, nullptr, forkjoin ? g : nullptr});
                keep_graph = forkjoin;
            }
        }
        if (e != hipSuccess && !rc) rc = fail((int)e, "closing capture: %s", hipGetErrorString(e));
        if (!keep_graph) (void)hipGraphDestroy(g);
    }
    mode = EAGER;
    ev_next = 0;
    return rc;
}

int Sched::replay() {
    for (const SeqOp& o : ops) {
        hipStream_t s = o.on_panel ? panel : main;
        if (o.kind == 0) GPMC_HIP(hipGraphLaunch(o.exec, s));
        else if (o.kind == 1) GPMC_HIP(hipEventRecord(o.ev, s));
        else GPMC_HIP(hipStreamWaitEvent(s, o.ev, 0));
    }
    return 0;
}

// HIP runtimes before 7.2: destroying a fork/join exec is a use-after-free inside the runtime (gpmc_common.h, Sched),
// and "flush, then destroy" (round 2) only narrowed the window -- one host segmentation fault in a later capture was
// the same heap corruption.  On those runtimes a retired fork/join exec is therefore NEVER destroyed while the process
// lives: it is parked, with its template graph, in a process-wide graveyard (a leaked object: no destructor runs at
// unload either, so nothing of the runtime is touched after main()).  The graveyard is bounded: once it holds
// g_opt.graveyard_max execs (default 512) forkjoin_allowed() turns false and new sequences are recorded in the SEGMENTS form, whose linear
// single-stream execs have no internal branch stream and are destroyed normally.
static bool forkjoin_destroy_unsafe() {
    static int unsafe = -1;
    if (unsafe < 0) {
        int v = 0;
        unsafe = (hipRuntimeGetVersion(&v) != hipSuccess || v < 70200000) ? 1 : 0;
    }
    return unsafe == 1;
}

namespace {

struct Graveyard {
    std::mutex mu;
    std::vector<hipGraphExec_t> execs;
    std::vector<hipGraph_t> graphs;
    size_t nodes = 0;
};
Graveyard& graveyard() {
    static Graveyard* g = new Graveyard();      // leaked on purpose
    return *g;
}
}  // namespace

void graveyard_stats(int* execs, size_t* nodes) {
    Graveyard& g = graveyard();
    std::lock_guard<std::mutex> lk(g.mu);
    if (execs) *execs = (int)g.execs.size();
    if (nodes) *nodes = g.nodes;
}

bool forkjoin_allowed() {
    if (!forkjoin_destroy_unsafe()) return true;
    Graveyard& g = graveyard();
    std::lock_guard<std::mutex> lk(g.mu);
    return (int)g.execs.size() < g_opt.graveyard_max;
}

int Sched::drop_ops() {
    int rc = 0;
    for (SeqOp& o : ops) {
        if (o.kind != 0 || !o.exec) continue;
        if (o.graph && forkjoin_destroy_unsafe()) {
            Graveyard& g = graveyard();
            std::lock_guard<std::mutex> lk(g.mu);
            size_t n = 0;
            if (hipGraphGetNodes(o.graph, nullptr, &n) == hipSuccess) g.nodes += n;
            g.execs.push_back(o.exec);
            g.graphs.push_back(o.graph);
            continue;
        }
        hipError_t e = hipGraphExecDestroy(o.exec);
        if (e != hipSuccess && !rc) rc = fail((int)e, "hipGraphExecDestroy: %s", hipGetErrorString(e));
        if (o.graph) (void)hipGraphDestroy(o.graph);
    }
    ops.clear();
    return rc;
}

// a caller-recorded sequence (plan.hip, gpmc_seq_*): it may span the look-ahead's two streams, so it leaves like a
// fork/join exec
int seq_retire(Seq* q) {
    if (!q) return 0;
    int rc = 0;
    if (q->exec) {
        if (forkjoin_destroy_unsafe()) {
            Graveyard& g = graveyard();
            std::lock_guard<std::mutex> lk(g.mu);
            size_t n = 0;
            if (q->graph && hipGraphGetNodes(q->graph, nullptr, &n) == hipSuccess) g.nodes += n;
            g.execs.push_back(q->exec);
            if (q->graph) g.graphs.push_back(q->graph);
        } else {
            hipError_t e = hipGraphExecDestroy(q->exec);
            if (e != hipSuccess) rc = fail((int)e, "hipGraphExecDestroy: %s", hipGetErrorString(e));
            if (q->graph) (void)hipGraphDestroy(q->graph);
        }
    }
    delete q;
    return rc;
}

int Sched::destroy() {
    int rc = drop_ops();
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    events.clear();
    if (own_panel && panel) (void)hipStreamDestroy(panel);
    panel = nullptr;
    return rc;
}

Sched* eager_sched(hipStream_t main_stream) {
    static Sched g;     // EAGER only: never captured, its event ring is only ever re-recorded eagerly
    if (!g.panel && g.init(main_stream)) return nullptr;
    g.main = main_stream;
    g.mode = Sched::EAGER;
    return &g;
}

}  // namespace gpmc
