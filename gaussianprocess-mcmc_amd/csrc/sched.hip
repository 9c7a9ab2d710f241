// sched.hip -- the launch schedule object (gpmc_common.h `Sched`): two streams, cross-stream edges, and the
// segmented record / replay form the plan uses instead of one fork/join hipGraph.
#include "gpmc_common.h"

namespace gpmc {

int Sched::init(hipStream_t main_stream) {
    main = main_stream;
    if (!panel) {
        int lo = 0, hi = 0;
        GPMC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
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
    GPMC_HIP(hipStreamWaitEvent(s, e, 0));
    return 0;
}

int Sched::begin(int mode_) {
    if (!ops.empty()) return fail(-1, "Sched::begin: a recorded sequence already exists");
    mode = mode_;
    ev_next = 0;
    if (mode == EAGER) return 0;
    GPMC_HIP(hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal));
    cap_main = true;
    return 0;
}

int Sched::end() {
    if (mode == EAGER) return 0;
    const bool forkjoin = (mode == FORKJOIN);
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

// HIP runtimes before 7.2 need a fork/join exec flushed before it is destroyed (gpmc_common.h, Sched)
static bool forkjoin_needs_flush() {
    static int need = -1;
    if (need < 0) {
        int v = 0;
        need = (hipRuntimeGetVersion(&v) != hipSuccess || v < 70200000) ? 1 : 0;
    }
    return need == 1;
}

int Sched::drop_ops() {
    int rc = 0;
    for (SeqOp& o : ops) {
        if (o.kind != 0 || !o.exec) continue;
        if (o.graph && forkjoin_needs_flush()) {
            // one launch with every node disabled: nothing runs, nothing is read or written
            size_t n = 0;
            bool ok = (hipGraphGetNodes(o.graph, nullptr, &n) == hipSuccess);
            std::vector<hipGraphNode_t> nodes(n);
            ok = ok && (n == 0 || hipGraphGetNodes(o.graph, nodes.data(), &n) == hipSuccess);
            for (size_t i = 0; ok && i < n; ++i) ok = (hipGraphNodeSetEnabled(o.exec, nodes[i], 0) == hipSuccess);
            if (ok && hipGraphLaunch(o.exec, main) == hipSuccess) (void)hipStreamSynchronize(main);
            else if (!rc) rc = fail(-1, "Sched::drop_ops: could not flush a fork/join exec before destroying it");
        }
        hipError_t e = hipGraphExecDestroy(o.exec);
        if (e != hipSuccess && !rc) rc = fail((int)e, "hipGraphExecDestroy: %s", hipGetErrorString(e));
        if (o.graph) (void)hipGraphDestroy(o.graph);
    }
    ops.clear();
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
