// v_permlane32_swap through the builtin: which lanes end up where (used by csrc/mlp_fused.hip to turn accumulator order into operand order)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d;
    unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a[l] = l, b[l] = 100 + l;  r = permlane32_swap(a, b)\n");
    printf("r[0]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[0], h[31], h[32], h[63]);
    printf("r[1]: lane 0 -> %u, lane 31 -> %u, lane 32 -> %u, lane 63 -> %u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
