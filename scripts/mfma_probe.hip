// Empirical lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950: one-hot A at lane la, one-hot B at lane lb,
// report which output lanes become non-zero; for cbsz/abid (A-block broadcast) variants too.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID, int BLGP>
__global__ void probe(unsigned long long* mask) {
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
    double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, BLGP);
    unsigned long long m = __ballot(d != 0.0);
    if (lane == 0) mask[blockIdx.x] = m;
}
static unsigned long long h[4096];
template <int CBSZ, int ABID, int BLGP>
static void run(unsigned long long* dm) {
    hipLaunchKernelGGL((probe<CBSZ, ABID, BLGP>), dim3(4096), dim3(64), 0, 0, dm);
    (void)hipMemcpy(h, dm, sizeof(h), hipMemcpyDeviceToHost);
    printf("== cbsz=%d abid=%d blgp=%d\n", CBSZ, ABID, BLGP);
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            unsigned long long m = h[la * 64 + lb];
            if (!m) continue;
            printf(" B%02d>", lb);
            bool first = true;
            for (int l = 0; l < 64; ++l) if (m >> l & 1) { printf(first ? "%d" : ",%d", l); first = false; }
        }
        printf("\n");
    }
}
int main() {
    unsigned long long* dm; (void)hipMalloc(&dm, sizeof(h));
    run<0, 0, 0>(dm);
    run<2, 0, 0>(dm);
    run<2, 1, 0>(dm);
    run<2, 3, 0>(dm);
    run<1, 0, 0>(dm);
    return 0;
}
