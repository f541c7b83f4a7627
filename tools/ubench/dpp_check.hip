// dpp_check.hip -- the DPP / readlane wave reductions of csrc/common.hpp against the __shfl_xor butterflies they replace,
// bit for bit, on random data.  Build: hipcc --offload-arch=gfx950 -O3 -I dyadic-interaction-modeling_amd/csrc -o dpp_check dpp_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../dyadic-interaction-modeling_amd/csrc/common.hpp"
using namespace dimx;

__global__ void k(const float* in, const int* idx, float* out, int* outi) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    const float v = in[t];
    float a = v;
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    float a1 = v;
    for (int o = 1; o < 64; o <<= 1) a1 += __shfl_xor(a1, o);
    float m = v;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float g8 = v;
    for (int o = 1; o < 8; o <<= 1) g8 += __shfl_xor(g8, o);
    float g16 = v;
    for (int o = 1; o < 16; o <<= 1) g16 += __shfl_xor(g16, o);
    float bv = v;
    int bi = idx[t];
    for (int o = 1; o < 64; o <<= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    float nv = v;
    int ni = idx[t];
    wave_argmax(nv, ni);
    int xbad = 0;
    const int iv = idx[t] * 131 + t;
    xbad += xor_lane_i32<1>(iv) != __shfl_xor(iv, 1);
    xbad += xor_lane_i32<2>(iv) != __shfl_xor(iv, 2);
    xbad += xor_lane_i32<4>(iv) != __shfl_xor(iv, 4);
    xbad += xor_lane_i32<8>(iv) != __shfl_xor(iv, 8);
    xbad += xor_lane_i32<16>(iv) != __shfl_xor(iv, 16);
    xbad += xor_lane_i32<32>(iv) != __shfl_xor(iv, 32);
    out[t * 12 + 0] = a1; out[t * 12 + 1] = wave_sum_fast(v);
    out[t * 12 + 2] = m; out[t * 12 + 3] = wave_max(v);
    out[t * 12 + 4] = g8; out[t * 12 + 5] = group8_sum(v);
    out[t * 12 + 6] = g16; out[t * 12 + 7] = row16_sum(v);
    out[t * 12 + 8] = bv; out[t * 12 + 9] = nv;
    out[t * 12 + 10] = a; out[t * 12 + 11] = wave_sum(v);
    outi[t * 2] = bi; outi[t * 2 + 1] = ni;
    if (xbad) atomicAdd(outi + 2 * 64 * 256, xbad);
}

int main() {
    const int N = 64 * 256;
    float* h = (float*)malloc(N * 4);
    int* hi = (int*)malloc(N * 4);
    srand(1);
    for (int i = 0; i < N; ++i) { h[i] = (float)rand() / RAND_MAX * 2.f - 1.f + (i % 7 == 0 ? 100.f : 0.f); if (i % 97 < 5) h[i] = 0.5f; hi[i] = rand() % 512; }
    float *d, *o; int *di, *oi;
    hipMalloc(&d, N * 4); hipMalloc(&o, N * 48); hipMalloc(&di, N * 4); hipMalloc(&oi, N * 8 + 4); hipMemset(oi, 0, N * 8 + 4);
    hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice); hipMemcpy(di, hi, N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 64), dim3(64), 0, 0, d, di, o, oi);
    float* ho = (float*)malloc(N * 48); int* hoi = (int*)malloc(N * 8 + 4);
    hipMemcpy(ho, o, N * 48, hipMemcpyDeviceToHost); hipMemcpy(hoi, oi, N * 8 + 4, hipMemcpyDeviceToHost);
    int bad[6] = {0, 0, 0, 0, 0, 0};
    for (int t = 0; t < N; ++t) {
        for (int p = 0; p < 5; ++p) if (memcmp(&ho[t * 12 + 2 * p], &ho[t * 12 + 2 * p + 1], 4)) ++bad[p];
        if (hoi[2 * t] != hoi[2 * t + 1]) ++bad[5];
    }
    int bad_exact = 0, dif = 0;
    for (int t = 0; t < N; ++t) {
        if (memcmp(&ho[t * 12 + 10], &ho[t * 12 + 11], 4)) ++bad_exact;
        if (memcmp(&ho[t * 12 + 0], &ho[t * 12 + 10], 4)) ++dif;
    }
    printf("mismatches vs the xor butterflies: wave_sum_fast (ascending) %d, wave_sum (descending) %d, wave_max %d, group8 %d, row16 %d, "
           "argmax value %d, argmax index %d\n", bad[0], bad_exact, bad[1], bad[2], bad[3], bad[4], bad[5]);
    printf("xor_lane<1..32> vs __shfl_xor: %d mismatching lanes\n", hoi[2 * N]);
    printf("descending (32..1) vs ascending (1..32) xor butterfly: %d of %d lanes differ in the last bits\n", dif, N);
    return 0;
}
