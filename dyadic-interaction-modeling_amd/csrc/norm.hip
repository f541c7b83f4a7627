// norm.hip -- LayerNorm (rows of 384 / 512 / 768 / 1152 f32) and InstanceNorm1d over time.
//
// LayerNorm: reference nn.LayerNorm(eps=1e-5) in code/models/lib/base_models.py:13 (VQ blocks, with
// bias) and x-transformers' bias-free LayerNorm (pre-norms / final_norm of every encoder/decoder
// layer) plus SLMFT.norm_s (code/seq2seq_pretrain.py:411).  One wave per row, the row lives in
// registers, mean and variance are two exact passes (mean of squared deviations, like torch).
//
// InstanceNorm: nn.InstanceNorm1d(affine=False) of code/models/stage1_BIWI.py:266,333 applied to a
// [B,L,C] activation: per (clip, channel) statistics over the first len_b frames, biased variance.
#include "common.hpp"

namespace dimx {

template <typename OutT, int C>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M) {
    constexpr int PER = C / 64;  // elements per lane (6 or 18), read as float2
    constexpr int NV = PER / 2;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float2* xr = (const float2*)(x + (size_t)row * C);
    const float2* g2 = (const float2*)gamma;
    const float2* b2 = (const float2*)beta;
    float2 v[NV], gv[NV], bv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[lane + 64 * i];
        s += v[i].x + v[i].y;
    }
    // gamma / beta are fetched now so that their (possibly HBM-miss) latency overlaps the two reductions
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gv[i] = g2[lane + 64 * i];
        bv[i] = beta ? b2[lane + 64 * i] : make_float2(0.f, 0.f);
    }
    const float mean = wave_sum_sel<sizeof(OutT) == 2>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean;
        q += a * a + b * b;
    }
    const float rstd = rsqrtf(wave_sum_sel<sizeof(OutT) == 2>(q) * (1.0f / C) + 1e-5f);
    OutT* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c2 = lane + 64 * i;
        const float o0 = (v[i].x - mean) * rstd * gv[i].x + bv[i].x;
        const float o1 = (v[i].y - mean) * rstd * gv[i].y + bv[i].y;
        if (sizeof(OutT) == 2) {
            *(uint32_t*)(yr + 2 * c2) = pack_bf16x2(o0, o1);
        } else {
            *(float2*)(yr + 2 * c2) = make_float2(o0, o1);
        }
    }
}

// Decode-step residual update fused with the following pre-norm:
//   x[row] += slabs[0][row] + slabs[1][row] + ...   (split-K partial sums of the previous projection, added
//   in slab order -> deterministic, no atomics), x is written back, y = LayerNorm(x) (gamma only, x-tf style).
template <typename OutT, int C>
__global__ __launch_bounds__(256) void add_slabs_layernorm_kernel(float* __restrict__ x,
                                                                  const float* __restrict__ slabs, int nslab,
                                                                  long slab_stride, OutT* __restrict__ y,
                                                                  const float* __restrict__ gamma, int M) {
    constexpr int NV = C / 128;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float2* xr = (float2*)(x + (size_t)row * C);
    const float2* g2 = (const float2*)gamma;
    float2 v[NV], gv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[lane + 64 * i];
        gv[i] = g2[lane + 64 * i];
    }
    // the slabs come from another kernel's split-K blocks (memory-side latency): all loads of up to 4 slabs are in flight
    // together, the adds stay in slab order (deterministic)
    for (int s0 = 0; s0 < nslab; s0 += 4) {
        float2 t[4][NV];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = s0 + k < nslab;
            const float2* sr = (const float2*)(slabs + (size_t)(on ? s0 + k : 0) * slab_stride + (size_t)row * C);
#pragma unroll
            for (int i = 0; i < NV; ++i) t[k][i] = sr[lane + 64 * i];  // unconditional (a select per element would
                                                                        // serialise the loads); masked below
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (s0 + k < nslab) {  // wave-uniform
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    v[i].x += t[k][i].x;
                    v[i].y += t[k][i].y;
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (nslab > 0) xr[lane + 64 * i] = v[i];
        s += v[i].x + v[i].y;
    }
    const float mean = wave_sum_sel<sizeof(OutT) == 2>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean;
        q += a * a + b * b;
    }
    const float rstd = rsqrtf(wave_sum_sel<sizeof(OutT) == 2>(q) * (1.0f / C) + 1e-5f);
    OutT* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c2 = lane + 64 * i;
        const float o0 = (v[i].x - mean) * rstd * gv[i].x, o1 = (v[i].y - mean) * rstd * gv[i].y;
        if (sizeof(OutT) == 2) {
            *(uint32_t*)(yr + 2 * c2) = pack_bf16x2(o0, o1);
        } else {
            *(float2*)(yr + 2 * c2) = make_float2(o0, o1);
        }
    }
}

int launch_add_slabs_layernorm(int out_dtype, float* x, const float* slabs, int nslab, long slab_stride, void* y,
                               const float* gamma, int M, int C, hipStream_t s) {
    DIMX_REQUIRE(x && y && gamma && M > 0 && (nslab == 0 || slabs), DIMX_ERR_ARG, "add_slabs_layernorm: null operand");
    DIMX_REQUIRE(C == 1152 || C == 384 || C == 512 || C == 768, DIMX_ERR_ARG, "add_slabs_layernorm: C=%d", C);
    // decode batches are a few hundred rows: one row (one wave) per block spreads them over all CUs instead of M/4
    const int wpb = M <= 1024 ? 1 : 4;
    dim3 grid(ceil_div(M, wpb)), block(64 * wpb);
#define ASL(OT, CC) hipLaunchKernelGGL((add_slabs_layernorm_kernel<OT, CC>), grid, block, 0, s, x, slabs, nslab, slab_stride, (OT*)y, gamma, M)
#define ASL_C(OT)                      \
    do {                               \
        if (C == 384) ASL(OT, 384);    \
        else if (C == 512) ASL(OT, 512); \
        else if (C == 768) ASL(OT, 768); \
        else ASL(OT, 1152);            \
    } while (0)
    if (out_dtype == DIMX_BF16) ASL_C(bf16); else ASL_C(float);
#undef ASL_C
#undef ASL
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_layernorm(int out_dtype, const float* x, void* y, const float* gamma, const float* beta, int M, int C,
                     hipStream_t s) {
    DIMX_REQUIRE(x && y && gamma && M > 0, DIMX_ERR_ARG, "layernorm: null operand");
    DIMX_REQUIRE(C == 384 || C == 1152 || C == 512 || C == 768, DIMX_ERR_ARG,
                 "layernorm: C=%d not in {384,512,768,1152}", C);
    dim3 grid(ceil_div(M, 4)), block(256);
#define LN_LAUNCH(OT, CC) hipLaunchKernelGGL((layernorm_kernel<OT, CC>), grid, block, 0, s, x, (OT*)y, gamma, beta, M)
#define LN_C(OT)                             \
    do {                                     \
        if (C == 384) LN_LAUNCH(OT, 384);    \
        else if (C == 512) LN_LAUNCH(OT, 512); \
        else if (C == 768) LN_LAUNCH(OT, 768); \
        else LN_LAUNCH(OT, 1152);            \
    } while (0)
    if (out_dtype == DIMX_BF16) LN_C(bf16); else LN_C(float);
#undef LN_C
#undef LN_LAUNCH
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

// block = 64 channels x 8 time lanes; grid = (C/64, B)
template <typename OutT>
__global__ __launch_bounds__(512) void instnorm_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                                                       const int32_t* __restrict__ lens, int T, int C) {
    __shared__ float red[8][64];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;
    int len = lens ? lens[b] : T;
    len = len < 1 ? 1 : (len > T ? T : len);
    const float* xb = x + (size_t)b * T * C + c;
    float s = 0.f;
    for (int t = ty; t < len; t += 8) s += xb[(size_t)t * C];
    red[ty][threadIdx.x & 63] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i][threadIdx.x & 63];
    const float mean = tot / (float)len;
    __syncthreads();
    float q = 0.f;
    for (int t = ty; t < len; t += 8) {
        const float d = xb[(size_t)t * C] - mean;
        q += d * d;
    }
    red[ty][threadIdx.x & 63] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) qt += red[i][threadIdx.x & 63];
    const float rstd = rsqrtf(qt / (float)len + 1e-5f);
    OutT* yb = y + (size_t)b * T * C + c;
    // frames beyond len get the same affine map: finite, never consumed by valid rows
    for (int t = ty; t < T; t += 8) store_from_f32<OutT>(yb + (size_t)t * C, (xb[(size_t)t * C] - mean) * rstd);
}

int launch_instnorm(int out_dtype, const float* x, void* y, const int32_t* lens, int B, int T, int C, hipStream_t s) {
    DIMX_REQUIRE(x && y && B > 0 && T > 0, DIMX_ERR_ARG, "instnorm: null operand");
    DIMX_REQUIRE(C % 64 == 0, DIMX_ERR_ARG, "instnorm: C=%d must be a multiple of 64", C);
    dim3 grid(C / 64, B), block(512);
    if (out_dtype == DIMX_BF16)
        hipLaunchKernelGGL((instnorm_kernel<bf16>), grid, block, 0, s, x, (bf16*)y, lens, T, C);
    else
        hipLaunchKernelGGL((instnorm_kernel<float>), grid, block, 0, s, x, (float*)y, lens, T, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
