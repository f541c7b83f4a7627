// tr16_probe -- what ds_read_b64_tr_b16 returns on gfx950 (attention_tr.hip's V^T fragments depend on it).
// LDS holds u16 value = element index; every lane passes its own byte address; the four u16 it gets back are printed.
// Expected (attention_tr.hip header): lane i of a 16-lane group receives, as element j, element (i & 3) of the 8-byte piece
// addressed by lane 4 j + (i >> 2) of the same group.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/tr16_probe tools/ubench/tr16_probe.hip && tools/ubench/tr16_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr_bytes, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)(base + addr_bytes[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int h_addr[64];
    unsigned short h_out[256];
    int* d_addr;
    unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr));
    hipMalloc(&d_out, sizeof(h_out));
    int bad_total = 0;
    for (int pattern = 0; pattern < 2; ++pattern) {
        // pattern 0: lane l -> 8 l bytes (contiguous);  pattern 1: lane p of a group -> row (p >> 2) of 128 B, column piece p & 3, groups 1 KiB apart
        for (int l = 0; l < 64; ++l) h_addr[l] = pattern == 0 ? 8 * l : (l >> 4) * 1024 + ((l & 15) >> 2) * 128 + (l & 3) * 8;
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int src_lane = (l & ~15) + 4 * j + ((l & 15) >> 2);
                const int expect = h_addr[src_lane] / 2 + (l & 3);
                if (h_out[4 * l + j] != expect) ++bad;
            }
        printf("pattern %d: %d of 256 elements differ from the assumed mapping\n", pattern, bad);
        for (int l = 0; l < 64; l += (pattern == 0 ? 1 : 5))
            printf("  lane %2d addr %5d -> %5d %5d %5d %5d\n", l, h_addr[l], h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
        bad_total += bad;
    }
    printf(bad_total ? "TR16 MAPPING MISMATCH\n" : "tr16 mapping as assumed\n");
    return bad_total ? 1 : 0;
}
