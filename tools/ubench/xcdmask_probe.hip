// xcdmask_probe.hip -- round 6: can a CU-masked stream own WHOLE XCDs (all 32 CUs of XCDs 0-3, none of 4-7)?
// Each case runs in its own process (argv[1] = case) so that a hang costs one `timeout`, not the whole probe.
//   cases: all | lo | hi | lo256 | graph | both | xs=<hexmask of XCDs> [blocks]
// For a case: launch a census kernel (512 threads, 100 KiB LDS -> one block per CU) on the masked stream and report
// where block b ran (XCD, and whether b -> XCD is b % nx over the enabled set), how many distinct CU slots were used
// and the launch time.  `graph`: the same launch captured on an ordinary stream and replayed with hipGraphLaunch on the
// masked stream (does the replay honour the mask?).  `both`: the two halves at once on two masked streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__global__ void census(unsigned* out, int spin) {
    extern __shared__ unsigned char dyn[];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc_id(); out[blockIdx.x * 2 + 1] = hw_id(); }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    dyn[threadIdx.x] = 0;
}
static std::vector<uint32_t> xcd_mask(unsigned xs) {  // bit i of the CU mask = CU (i / 8) of XCD (i % 8)
    std::vector<uint32_t> r(8, 0);
    for (int i = 0; i < 256; ++i) if ((xs >> (i & 7)) & 1) r[i >> 5] |= 1u << (i & 31);
    return r;
}
static void report(const char* name, const std::vector<unsigned>& h, int nblocks, unsigned xs, float us) {
    int cnt[16] = {0};
    std::set<unsigned> slots;
    int order[8], nx = 0;
    for (int x = 0; x < 8; ++x) if ((xs >> x) & 1) order[nx++] = x;
    int modhit = 0, rot = -1;
    for (int i = 0; i < nblocks; ++i) {
        cnt[h[2 * i] & 15]++;
        slots.insert((h[2 * i] << 16) | ((h[2 * i + 1] >> 8) & 0xff) | (((h[2 * i + 1] >> 13) & 7) << 8));
    }
    // rotation: which enabled XCD did block 0 take?
    for (int k = 0; k < nx; ++k) if (order[k] == (int)(h[0] & 15)) rot = k;
    for (int i = 0; i < nblocks; ++i) if (rot >= 0 && (int)(h[2 * i] & 15) == order[(i + rot) % nx]) ++modhit;
    printf("%-34s %3d blocks: per-XCD", name, nblocks);
    for (int x = 0; x < 8; ++x) printf(" %3d", cnt[x]);
    printf(" ; distinct slots %zu ; block b on enabled[(b+%d)%%%d]: %d of %d ; %.1f us ; first 10:", slots.size(), rot, nx, modhit, nblocks, us);
    for (int i = 0; i < 10 && i < nblocks; ++i) printf(" %u", h[2 * i] & 15);
    printf("\n");
    fflush(stdout);
}
static std::vector<uint32_t> g_raw;   // cases `bits=lo-hi`: a raw bit range instead of an XCD set
static void run(const char* name, unsigned xs, int nblocks, bool graph) {
    hipStream_t s;
    std::vector<uint32_t> mask = g_raw.empty() ? xcd_mask(xs) : g_raw;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-34s hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    {   // what the runtime says the stream's mask is
        std::vector<uint32_t> back(8, 0);
        hipError_t ge = hipExtStreamGetCUMask(s, 8, back.data());
        printf("  hipExtStreamGetCUMask: %s", ge == hipSuccess ? "" : hipGetErrorString(ge));
        for (int i = 0; i < 8; ++i) printf(" %08x", back[i]);
        printf("\n");
    }
    unsigned* d;
    CK(hipMalloc(&d, nblocks * 8));
    CK(hipMemset(d, 0xff, nblocks * 8));
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipGraphExec_t ge = nullptr;
    if (graph) {
        hipStream_t cap; CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        hipGraph_t g;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(census, dim3(nblocks), dim3(512), 100 * 1024, cap, d, 2000);
        CK(hipStreamEndCapture(cap, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));  // warm
        CK(hipStreamSynchronize(s));
    }
    CK(hipEventRecord(a, s));
    if (graph) CK(hipGraphLaunch(ge, s));
    else hipLaunchKernelGGL(census, dim3(nblocks), dim3(512), 100 * 1024, s, d, 2000);  // 20 us spin
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned> h(nblocks * 2);
    CK(hipMemcpy(h.data(), d, nblocks * 8, hipMemcpyDeviceToHost));
    report(name, h, nblocks, xs, ms * 1e3f);
}
static void both() {
    hipStream_t s[2];
    unsigned xs[2] = {0x0f, 0xf0};
    unsigned* d[2];
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int k = 0; k < 2; ++k) {
        std::vector<uint32_t> m = xcd_mask(xs[k]);
        CK(hipExtStreamCreateWithCUMask(&s[k], 8, m.data()));
        CK(hipMalloc(&d[k], 128 * 8));
    }
    hipEvent_t a[2], b[2];
    for (int k = 0; k < 2; ++k) { CK(hipEventCreate(&a[k])); CK(hipEventCreate(&b[k])); }
    for (int rep = 0; rep < 2; ++rep) {
        for (int k = 0; k < 2; ++k) CK(hipEventRecord(a[k], s[k]));
        for (int r = 0; r < 20; ++r)
            for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(census, dim3(128), dim3(512), 100 * 1024, s[k], d[k], 2000);
        for (int k = 0; k < 2; ++k) CK(hipEventRecord(b[k], s[k]));
        for (int k = 0; k < 2; ++k) CK(hipStreamSynchronize(s[k]));
    }
    for (int k = 0; k < 2; ++k) {
        float ms; CK(hipEventElapsedTime(&ms, a[k], b[k]));
        std::vector<unsigned> h(256);
        CK(hipMemcpy(h.data(), d[k], 128 * 8, hipMemcpyDeviceToHost));
        report(k ? "both: hi half, 20 launches" : "both: lo half, 20 launches", h, 128, xs[k], ms * 1e3f);
    }
    printf("(20 x 20 us spins per half: ~400 us + launch floors if the halves really run side by side, ~800 us if they serialise)\n");
}
int main(int argc, char** argv) {
    const char* c = argc > 1 ? argv[1] : "all";
    if (!strcmp(c, "all")) run("all XCDs", 0xff, 256, false);
    else if (!strcmp(c, "lo")) run("XCDs 0-3 whole", 0x0f, 128, false);
    else if (!strcmp(c, "hi")) run("XCDs 4-7 whole", 0xf0, 128, false);
    else if (!strcmp(c, "lo256")) run("XCDs 0-3 whole, 256 blocks", 0x0f, 256, false);
    else if (!strcmp(c, "graph")) run("XCDs 0-3 whole, graph replay", 0x0f, 128, true);
    else if (!strcmp(c, "both")) both();
    else if (!strncmp(c, "bits=", 5)) {
        int lo = 0, hi = 0;
        sscanf(c + 5, "%d-%d", &lo, &hi);
        g_raw.assign(8, 0);
        for (int i = lo; i < hi; ++i) g_raw[i >> 5] |= 1u << (i & 31);
        run(c, 0xff, argc > 2 ? atoi(argv[2]) : 128, false);
    }
    else if (!strncmp(c, "xs=", 3)) run(c, (unsigned)strtoul(c + 3, nullptr, 16), argc > 2 ? atoi(argv[2]) : 128, false);
    return 0;
}
