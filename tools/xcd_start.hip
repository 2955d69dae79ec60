// Which XCD does block 0 of a launch land on?  (MI355X_MICROARCH.md: blocks are dealt round-robin over the 8 XCDs, but the XCD of block 0 is not fixed.)
// Launches a sequence of kernels with known grid sizes on one stream — once as plain launches, once as a captured graph replayed twice —
// and records the XCC id of blocks 0..15 of every launch: does the dispatcher keep ONE running pointer across launches (start of launch i+1 =
// start of launch i + grid i mod 8)?   build: hipcc -O2 --offload-arch=gfx950 tools/xcd_start.hip -o tools/xcd_start
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void probe_busy(int* out, int slot, int spin) {   // the same with work in every block: consecutive launches now overlap at their tails
    if (threadIdx.x == 0 && blockIdx.x < 16) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[slot * 16 + blockIdx.x] = (int)(x & 15);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin * (1 + (blockIdx.x & 3))) __builtin_amdgcn_s_sleep(8);   // 100 MHz ticks; uneven block lengths
}
__global__ void probe(int* out, int slot) {
    if (threadIdx.x == 0 && blockIdx.x < 16) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[slot * 16 + blockIdx.x] = (int)(x & 15);
    }
}

int main() {
    const int grids[] = {8, 8, 16, 5, 8, 3, 192, 256, 240, 720, 7, 1, 8, 144, 250, 8};
    const int n = sizeof(grids) / sizeof(grids[0]);
    int* d;
    CK(hipMalloc(&d, 3 * n * 16 * sizeof(int)));
    CK(hipMemset(d, 0xFF, 3 * n * 16 * sizeof(int)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(probe, dim3(grids[i]), dim3(256), 0, s, d, i);
    CK(hipStreamSynchronize(s));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(probe, dim3(grids[i]), dim3(256), 0, s, d, n + i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<int> h(3 * n * 16);
    CK(hipMemcpy(h.data(), d, 2 * n * 16 * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> first(h.begin() + n * 16, h.begin() + 2 * n * 16);
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data() + 2 * n * 16, d + n * 16, n * 16 * sizeof(int), hipMemcpyDeviceToHost));
    {   // busy kernels back to back (5-20 us per block), stream-ordered: does block 0 still start on XCC 0?
        CK(hipMemset(d, 0xFF, n * 16 * sizeof(int)));
        for (int rep = 0; rep < 3; ++rep)
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(probe_busy, dim3(grids[i]), dim3(256), 0, s, d, i, 500);
        CK(hipStreamSynchronize(s));
        std::vector<int> hb(n * 16);
        CK(hipMemcpy(hb.data(), d, n * 16 * sizeof(int), hipMemcpyDeviceToHost));
        printf("busy kernels back to back (third pass):\n");
        for (int i = 0; i < n; ++i) {
            printf("  grid %4d: xcc of blocks 0..7:", grids[i]);
            for (int b = 0; b < 8 && b < grids[i]; ++b) printf(" %d", hb[i * 16 + b]);
            printf("\n");
        }
    }
    const char* names[3] = {"stream launches", "graph replay 1", "graph replay 2"};
    for (int m = 0; m < 3; ++m) {
        printf("%s\n", names[m]);
        int pred = -1;
        for (int i = 0; i < n; ++i) {
            const int* r = (m == 1 ? first.data() : h.data() + (m == 2 ? 2 : 0) * n * 16) + i * 16;
            printf("  grid %4d: xcc of blocks 0..%d:", grids[i], grids[i] < 16 ? grids[i] - 1 : 15);
            for (int b = 0; b < 16 && b < grids[i]; ++b) printf(" %d", r[b]);
            if (pred >= 0) printf("   (running-pointer model predicts block 0 on %d)", pred);
            printf("\n");
            pred = (r[0] + grids[i]) & 7;
        }
    }
    return 0;
}
