// vq.hip -- nearest-codebook search of the VQ-VAE quantiser.
//
// Reference: VectorQuantizer.forward, code/models/lib/quantizer.py:35-47:
//     d = sum(z^2) + sum(E^2) - 2 z.E^T ;  idx = argmin_j d   (first index on ties)
// evaluated in float32 with the same association ((zz + ee_j) - 2*dot_j).
//
// Wavefront-reduced L2 search: one wave handles 4 latent vectors at a time; lane l owns the 8 codes
// l, l+64, ..., l+448, reads the k-major codebook E^T[128][512] fully coalesced (256 B per wave load,
// L2-resident: 256 KiB) and accumulates dot_j as a k-ascending fmaf chain (deterministic order; the C
// oracle oracle/vq_argmin.c reproduces the distances bit for bit).  z values are broadcast with
// v_readlane.  Each lane keeps (best, index, second best) over its codes, then the wave reduces them
// with lexicographic (distance, index) ordering so ties resolve to the smallest index like torch.argmin.
#include "common.hpp"

namespace dimx {
namespace {

constexpr int TPW = 4;  // latent vectors per wave iteration

struct Best {
    float d;
    int i;
    float d2;
};

__device__ __forceinline__ void best_push(Best& b, float d, int i) {
    if (d < b.d || (d == b.d && i < b.i)) {
        b.d2 = b.d;
        b.d = d;
        b.i = i;
    } else {
        b.d2 = fminf(b.d2, d);
    }
}

__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ z, int N,
                                                        const float* __restrict__ Et,
                                                        const float* __restrict__ ee, int32_t* __restrict__ idx,
                                                        float* __restrict__ best_d, float* __restrict__ margin) {
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nw = (gridDim.x * 256) >> 6;
    float eev[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) eev[c] = ee[lane + 64 * c];

    for (int t0 = gw * TPW; t0 < N; t0 += nw * TPW) {
        float zr[TPW][2];
        float zz[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tok = (t0 + t) < N ? (t0 + t) : N - 1;
            zr[t][0] = z[(size_t)tok * 128 + lane];
            zr[t][1] = z[(size_t)tok * 128 + 64 + lane];
            zz[t] = wave_sum(fmaf(zr[t][1], zr[t][1], zr[t][0] * zr[t][0]));  // explicit order: oracle/vq_argmin.c
        }
        float acc[TPW][8];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[t][c] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll 4
            for (int k = 0; k < 64; ++k) {
                float e[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) e[c] = Et[(size_t)(kk * 64 + k) * 512 + lane + 64 * c];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const float zk = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zr[t][kk]), k));
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[t][c] = fmaf(zk, e[c], acc[t][c]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            Best bst = {3.0e38f, 0x7fffffff, 3.0e38f};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float d = fmaf(-2.0f, acc[t][c], zz[t] + eev[c]);  // = (zz + ee) - 2 dot, exactly
                best_push(bst, d, lane + 64 * c);
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                Best ot;
                ot.d = __shfl_xor(bst.d, o);
                ot.i = __shfl_xor(bst.i, o);
                ot.d2 = __shfl_xor(bst.d2, o);
                const bool take = ot.d < bst.d || (ot.d == bst.d && ot.i < bst.i);
                const float loser = take ? bst.d : ot.d;
                const float nd2 = fminf(fminf(bst.d2, ot.d2), loser);
                if (take) {
                    bst.d = ot.d;
                    bst.i = ot.i;
                }
                bst.d2 = nd2;
            }
            if (lane == 0 && (t0 + t) < N) {
                idx[t0 + t] = bst.i;
                if (best_d) best_d[t0 + t] = bst.d;
                if (margin) margin[t0 + t] = bst.d2 - bst.d;
            }
        }
    }
}

}  // namespace

int launch_vq_argmin(const float* z, int N, const float* Et, const float* ee, int32_t* idx, float* best_d,
                     float* margin, hipStream_t s) {
    DIMX_REQUIRE(z && Et && ee && idx && N > 0, DIMX_ERR_ARG, "vq_argmin: null operand");
    int blocks = ceil_div(N, 4 * TPW);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(vq_argmin_kernel, dim3(blocks), dim3(256), 0, s, z, N, Et, ee, idx, best_d, margin);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
