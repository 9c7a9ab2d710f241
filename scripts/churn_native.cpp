// churn_native.cpp -- gpmc_debug_graph_churn driven from plain C++ (no python, no torch): does the round-1 abort
// (graph exec destroyed after a device-wide sync) need torch's streams/allocator in the process, or is it libgpmc's
// own capture pattern?
//   hipcc -O2 scripts/churn_native.cpp -Iinclude -Lgaussianprocess-mcmc_amd/gpmc -lgpmc -o scripts/churn_native
//   LD_LIBRARY_PATH=gaussianprocess-mcmc_amd/gpmc scripts/churn_native FORM*2+DEVSYNC ITERS N [opt=value ...]
//   (FORM 0 = segmented linear graphs, 1 = round 1's fork/join graph; run on torch's bundled runtime with
//    LD_PRELOAD=<torch>/lib/libamdhip64.so)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gpmc.h"

extern "C" int gpmc_debug_graph_churn(gpmc_plan* p, int form, int device_sync, int iters);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define GK(x) do { int r_ = (x); if (r_) { printf("%s -> %d: %s\n", #x, r_, gpmc_last_error()); exit(3); } } while (0)

int main(int argc, char** argv) {
    const int mode = atoi(argv[1]), iters = atoi(argv[2]), n = atoi(argv[3]), d = 2;
    for (int i = 4; i < argc; ++i) {
        char* eq = strchr(argv[i], '=');
        if (!eq) continue;
        *eq = 0;
        GK(gpmc_set_option(argv[i], atoi(eq + 1)));
    }
    GK(gpmc_set_option("use_graph", 0));
    { int v = 0; CK(hipRuntimeGetVersion(&v)); printf("hip runtime %d\n", v); }
    std::vector<double> X((size_t)n * d), g(n);
    srand(124);
    for (auto& v : X) v = 10.0 * rand() / RAND_MAX;
    for (auto& v : g) v = 5.0 * (2.0 * rand() / RAND_MAX - 1.0);
    double *Xd, *gd;
    CK(hipMalloc(&Xd, X.size() * 8));
    CK(hipMalloc(&gd, g.size() * 8));
    CK(hipMemcpy(Xd, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(gd, g.data(), g.size() * 8, hipMemcpyHostToDevice));
    const size_t wb = gpmc_plan_workspace_bytes(n, d, 8);
    void* ws;
    CK(hipMalloc(&ws, wb));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    gpmc_plan* p = nullptr;
    GK(gpmc_plan_create(&p, GPMC_KERNEL_RBF, n, d, 8, ws, wb, (gpmc_stream_t)s));
    GK(gpmc_plan_set_data(p, Xd, d, gd));
    const double kp[3] = {100.0, 1.44, 1.0};
    double out[4];
    GK(gpmc_plan_logml_async(p, kp, 3, 0));
    GK(gpmc_plan_result(p, 0, out));
    printf("eager logml %.6f info %d\n", out[0], (int)out[3]);
    fflush(stdout);
    for (int done = 0; done < iters;) {
        const int k = iters - done < 8 ? iters - done : 8;
        GK(gpmc_debug_graph_churn(p, mode >> 1, mode & 1, k));
        done += k;
        printf("mode %d: %d iterations ok\n", mode, done);
        fflush(stdout);
    }
    GK(gpmc_plan_destroy(p));
    printf("mode %d: closed cleanly\n", mode);
    return 0;
}
