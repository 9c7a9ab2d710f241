// graph_repro.hip -- minimal, libgpmc-free reproducer for the hipGraphExecDestroy abort recorded in round 1
// (gpurun_out/stress_3.txt: "corrupted size vs. prev_size").  Question it answers: is the abort a property of ANY
// two-stream captured graph destroyed after a device-wide sync on ROCm 7.2, or of the way libgpmc captured its
// look-ahead schedule (process-global panel stream, a 16-event ring reused inside one capture and shared with eager
// launches)?
//
//   hipcc --offload-arch=gfx950 -O2 scripts/graph_repro.hip -o scripts/graph_repro
//   scripts/graph_repro FLAGS ITERS        (one flag set per process: the failure is a process abort)
//
// FLAGS (bit mask):
//    1  hipDeviceSynchronize() between the replay and the destroy        (the trigger seen in round 1)
//    2  20 fork/join pairs per capture instead of 1
//    4  fork/join events come from a ring of 16 that wraps inside one capture (else: one fresh event per use)
//    8  second stream = hipStreamNonBlocking + highest priority (else plain hipStreamCreate)
//   16  run the same sequence EAGERLY with the same streams/events before every capture
//   32  kernels spin ~20 us each (else trivial)
//   64  workaround of round 1: one more replay + stream sync right before the destroy
//  128  capture once, instantiate/destroy per iteration (else capture + graph destroy per iteration too)
//  256  single-stream capture (no fork) -- control
//  512  a 4-byte memset node and a D2D memcpy node open the captured sequence (as the plan's does)
// 1024  40 idle extra streams exist in the process (torch's stream pool)
// 2048  the side-branch kernels take 14 arguments and 80 KB of dynamic LDS (the panel kernel's launch shape),
//       10 of them per fork instead of 2
// third argument: fork/join pairs per capture when bit 2 is set (default 20)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void work(double* p, int spin) {
    double v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

__global__ void fat(double* p, int a1, int a2, int a3, int* info, double* q, int a6, unsigned long long* dbg, int a8,
                    const double* r, int a10, int a11, int a12, int spin) {
    extern __shared__ double sm[];
    sm[threadIdx.x] = p[blockIdx.x * blockDim.x + threadIdx.x] + a1 + a2 + a3 + a6 + a8 + a10 + a11 + a12;
    __syncthreads();
    double v = sm[(threadIdx.x + 1) & 255];
    for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
    if (v == 12345.678 && info) *info = 1;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v * 1e-30;
}

struct Ctx {
    bool fatk = false;
    int* info = nullptr;
    hipStream_t s, ps;
    std::vector<hipEvent_t> ring;
    std::vector<hipEvent_t> fresh;   // one per use, created on demand
    size_t fresh_used = 0;
    int ring_i = 0;
    bool use_ring;
    hipEvent_t next() {
        if (use_ring) { hipEvent_t e = ring[ring_i]; ring_i = (ring_i + 1) % 16; return e; }
        if (fresh_used == fresh.size()) { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); fresh.push_back(e); }
        return fresh[fresh_used++];
    }
    void rewind() { fresh_used = 0; ring_i = 0; }
};

static void enqueue(Ctx& c, double* a, double* b, int pairs, int spin, bool fork) {
    for (int k = 0; k < pairs; ++k) {
        hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, c.s, a, spin);
        if (fork) {
            hipEvent_t e1 = c.next();
            CK(hipEventRecord(e1, c.s));
            CK(hipStreamWaitEvent(c.ps, e1, 0));
            if (c.fatk) {
                for (int r = 0; r < 10; ++r)
                    hipLaunchKernelGGL(fat, dim3(96 - r), dim3(256), 80 * 1024, c.ps, b, 1, 2, 3, c.info, a, 6,
                                       (unsigned long long*)nullptr, 8, (const double*)a, 10, 11, 12, spin);
            } else {
                hipLaunchKernelGGL(work, dim3(8), dim3(256), 0, c.ps, b, spin);
                hipLaunchKernelGGL(work, dim3(8), dim3(256), 0, c.ps, b, spin);
            }
            hipEvent_t e2 = c.next();
            CK(hipEventRecord(e2, c.ps));
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, c.s, a, 4 * spin);
            CK(hipStreamWaitEvent(c.s, e2, 0));
        } else {
            hipLaunchKernelGGL(work, dim3(8), dim3(256), 0, c.s, b, spin);
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, c.s, a, 4 * spin);
        }
    }
    CK(hipGetLastError());
}

int main(int argc, char** argv) {
    const int flags = argc > 1 ? atoi(argv[1]) : 1;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    const bool dev_sync = flags & 1, many = flags & 2, eager = flags & 16, heavy = flags & 32;
    const bool workaround = flags & 64, capture_once = flags & 128, single = flags & 256;
    { int v = 0; CK(hipRuntimeGetVersion(&v)); printf("hip runtime %d\n", v); }
    Ctx c;
    c.use_ring = flags & 4;
    CK(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking));
    if (flags & 8) {
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&c.ps, hipStreamNonBlocking, hi));
    } else {
        CK(hipStreamCreate(&c.ps));
    }
    c.ring.resize(16);
    for (auto& e : c.ring) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    double *a, *b;
    CK(hipMalloc(&a, 256 * 256 * sizeof(double)));
    CK(hipMalloc(&b, 256 * 256 * sizeof(double)));
    CK(hipMemset(a, 0, 256 * 256 * sizeof(double)));
    CK(hipMemset(b, 0, 256 * 256 * sizeof(double)));
    const int pairs = many ? (argc > 3 ? atoi(argv[3]) : 20) : 1, spin = heavy ? 4000 : 4;
    c.fatk = flags & 2048;
    CK(hipMalloc(&c.info, 256));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fat), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<hipStream_t> idle;
    if (flags & 1024) {
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        for (int i = 0; i < 40; ++i) {
            hipStream_t t;
            CK(hipStreamCreateWithPriority(&t, hipStreamNonBlocking, (i & 1) ? hi : lo));
            idle.push_back(t);
        }
    }
    const bool mnodes = flags & 512;

    hipGraph_t g = nullptr;
    auto capture = [&]() {
        if (eager) {
            c.rewind();
            enqueue(c, a, b, pairs, spin, !single);
            CK(hipStreamSynchronize(c.s));
        }
        c.rewind();
        CK(hipStreamBeginCapture(c.s, hipStreamCaptureModeThreadLocal));
        if (mnodes) {
            CK(hipMemsetAsync(c.info, 0, sizeof(int), c.s));
            CK(hipMemcpyAsync(a + 4096, b, 4096 * sizeof(double), hipMemcpyDeviceToDevice, c.s));
        }
        enqueue(c, a, b, pairs, spin, !single);
        CK(hipStreamEndCapture(c.s, &g));
    };
    if (capture_once) capture();
    for (int it = 0; it < iters; ++it) {
        if (!capture_once) capture();
        hipGraphExec_t x = nullptr;
        CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(x, c.s));
        CK(hipStreamSynchronize(c.s));
        if (dev_sync) CK(hipDeviceSynchronize());
        if (workaround) {
            CK(hipGraphLaunch(x, c.s));
            CK(hipStreamSynchronize(c.s));
        }
        CK(hipGraphExecDestroy(x));
        if (!capture_once) { CK(hipGraphDestroy(g)); g = nullptr; }
        if ((it + 1) % 16 == 0) { printf("flags %d: %d iterations ok\n", flags, it + 1); fflush(stdout); }
    }
    if (g) CK(hipGraphDestroy(g));
    printf("flags %d: done, %d iterations clean\n", flags, iters);
    return 0;
}
