// graph_launch_cost.hip -- what does ONE hipGraphLaunch of a small linear graph cost on the device timeline?
// 240 kernels of ~spin us each on one stream, issued as: eager launches / one 240-node graph / 240 one-node graphs /
// 80 three-node graphs; plus the same with an event record + cross-stream wait pair after every third kernel.
//   hipcc --offload-arch=gfx950 -O2 scripts/graph_launch_cost.hip -o scripts/graph_launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void work(double* p, int spin) {
    double v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int spin = argc > 1 ? atoi(argv[1]) : 4000;
    { int v = 0; CK(hipRuntimeGetVersion(&v)); printf("hip runtime %d, spin %d\n", v, spin); }
    hipStream_t s, ps;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, hi));
    double *a, *b;
    CK(hipMalloc(&a, 256 * 256 * 8)); CK(hipMemset(a, 0, 256 * 256 * 8));
    CK(hipMalloc(&b, 256 * 256 * 8)); CK(hipMemset(b, 0, 256 * 256 * 8));
    const int NK = 240;
    auto launch = [&](hipStream_t st, double* p) { hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, p, spin); };
    auto make_graph = [&](int nodes, hipStream_t st, double* p) {
        hipGraph_t g; hipGraphExec_t x;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < nodes; ++i) launch(st, p);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        return x;
    };
    hipGraphExec_t g240 = make_graph(NK, s, a), g1 = make_graph(1, s, a), g3 = make_graph(3, s, a), p3 = make_graph(3, ps, b);
    std::vector<hipEvent_t> ev(2 * NK);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto timeit = [&](const char* name, auto fn) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            fn();
            const double t1 = now();
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(ps));
            const double t2 = now();
            if (rep == 2) printf("%-58s host issue %7.1f us   total %8.1f us   per kernel %6.2f us\n", name, (t1 - t0) * 1e6, (t2 - t0) * 1e6, (t2 - t0) * 1e6 / NK);
        }
    };
    timeit("240 eager launches", [&] { for (int i = 0; i < NK; ++i) launch(s, a); });
    timeit("one 240-node graph", [&] { CK(hipGraphLaunch(g240, s)); });
    timeit("240 one-node graphs", [&] { for (int i = 0; i < NK; ++i) CK(hipGraphLaunch(g1, s)); });
    timeit("80 three-node graphs", [&] { for (int i = 0; i < NK / 3; ++i) CK(hipGraphLaunch(g3, s)); });
    // cross-stream ping-pong as in the look-ahead: main 3 kernels | rec | panel waits, 3 kernels, rec | main waits ...
    timeit("40 x [main g3 | edge | panel g3 | edge]  (serial ping-pong)", [&] {
        int k = 0;
        for (int i = 0; i < NK / 6; ++i) {
            CK(hipGraphLaunch(g3, s)); CK(hipEventRecord(ev[k], s)); CK(hipStreamWaitEvent(ps, ev[k], 0)); ++k;
            CK(hipGraphLaunch(p3, ps)); CK(hipEventRecord(ev[k], ps)); CK(hipStreamWaitEvent(s, ev[k], 0)); ++k;
        }
    });
    timeit("same ping-pong with eager kernels instead of graphs", [&] {
        int k = 0;
        for (int i = 0; i < NK / 6; ++i) {
            for (int j = 0; j < 3; ++j) launch(s, a);
            CK(hipEventRecord(ev[k], s)); CK(hipStreamWaitEvent(ps, ev[k], 0)); ++k;
            for (int j = 0; j < 3; ++j) launch(ps, b);
            CK(hipEventRecord(ev[k], ps)); CK(hipStreamWaitEvent(s, ev[k], 0)); ++k;
        }
    });
    // the same ping-pong captured as ONE fork/join graph
    {
        hipGraph_t g; hipGraphExec_t x;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int k = 0;
        for (int i = 0; i < NK / 6; ++i) {
            for (int j = 0; j < 3; ++j) launch(s, a);
            CK(hipEventRecord(ev[k], s)); CK(hipStreamWaitEvent(ps, ev[k], 0)); ++k;
            for (int j = 0; j < 3; ++j) launch(ps, b);
            CK(hipEventRecord(ev[k], ps)); CK(hipStreamWaitEvent(s, ev[k], 0)); ++k;
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
        timeit("same ping-pong as one fork/join graph", [&] { CK(hipGraphLaunch(x, s)); });
    }
    return 0;
}
