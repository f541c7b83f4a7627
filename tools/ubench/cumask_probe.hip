// cumask_probe.hip -- how do hipExtStreamCreateWithCUMask bits map to XCDs / CUs on MI355X?
// For several masks: launch a census kernel (one 512-thread block per enabled CU, 100 KB LDS) on the masked stream and
// report how many blocks ran on each XCD and how many distinct (XCD, CU) slots were used.  Tuning probe only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__global__ void census(unsigned* out, int spin) {
    extern __shared__ unsigned char dyn[];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc_id(); out[blockIdx.x * 2 + 1] = hw_id(); }
    // stay resident a little so that all blocks overlap in time (one per CU)
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    dyn[threadIdx.x] = 0;
}
static void run(const char* name, const std::vector<uint32_t>& mask, int nblocks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    unsigned* d;
    CK(hipMalloc(&d, nblocks * 8));
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s));
    hipLaunchKernelGGL(census, dim3(nblocks), dim3(512), 100 * 1024, s, d, 2000);  // 20 us spin
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned> h(nblocks * 2);
    CK(hipMemcpy(h.data(), d, nblocks * 8, hipMemcpyDeviceToHost));
    int cnt[16] = {0};
    std::set<unsigned> slots;
    for (int i = 0; i < nblocks; ++i) {
        cnt[h[2 * i] & 15]++;
        slots.insert((h[2 * i] << 16) | ((h[2 * i + 1] >> 8) & 0xff) | (((h[2 * i + 1] >> 13) & 7) << 8));  // xcc, cu_id, se_id
    }
    printf("%-28s %3d blocks: per-XCD", name, nblocks);
    for (int x = 0; x < 8; ++x) printf(" %3d", cnt[x]);
    printf(" ; distinct (xcc,se,cu) %zu ; %.1f us ; first 12 blocks on XCD", slots.size(), ms * 1e3);
    for (int i = 0; i < 12 && i < nblocks; ++i) printf(" %u", h[2 * i]);
    printf("\n");
    CK(hipFree(d));
    CK(hipStreamDestroy(s));
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d\n", p.multiProcessorCount);
    std::vector<uint32_t> m(8, 0);
    auto setbits = [&](int lo, int hi) { std::vector<uint32_t> r(8, 0); for (int i = lo; i < hi; ++i) r[i >> 5] |= 1u << (i & 31); return r; };
    run("all 256 bits", setbits(0, 256), 256);
    run("bits 0..127", setbits(0, 128), 128);
    run("bits 128..255", setbits(128, 256), 128);
    run("bits 0..31", setbits(0, 32), 32);
    run("bits 32..63", setbits(32, 64), 32);
    { std::vector<uint32_t> r(8, 0); for (int i = 0; i < 256; i += 2) r[i >> 5] |= 1u << (i & 31); run("even bits", r, 128); }
    { std::vector<uint32_t> r(8, 0); for (int i = 0; i < 256; ++i) if ((i & 7) < 4) r[i >> 5] |= 1u << (i & 31); run("bits with (i&7)<4", r, 128); }
    { std::vector<uint32_t> r(8, 0); for (int i = 0; i < 256; ++i) if ((i & 7) >= 4) r[i >> 5] |= 1u << (i & 31); run("bits with (i&7)>=4", r, 128); }
    // more blocks than enabled CUs: does the masked stream really serialise them on the enabled CUs?
    run("bits 0..127, 256 blocks", setbits(0, 128), 256);
    return 0;
}
