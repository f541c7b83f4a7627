// train_kernels.hip -- device kernels of the training step: everything of the backward pass that is not a
// GEMM, plus the attention forward of the training path (which keeps the row log-sum-exp the backward needs).
//
// Reference: the reference trains with PyTorch autograd (code/x_engine_pt.py:9-60, code/finetune_s2s_pretrain.py:105-143);
// what is differentiated is the teacher-forced x-transformers stack of SLMFT.forward(mode='train')
// (code/seq2seq_pretrain.py:431-450, 496-514).  The operators below are the hand-derived adjoints of that stack:
//   * attention (Attention + Attend of x-transformers 1.30.16: scale 64^-0.5, masked scores filled with -FLT_MAX, softmax in
//     f32): forward with saved LSE, dQ and dK/dV with the probabilities recomputed from the LSE (no T x T tensor is kept);
//   * LayerNorm (eps 1e-5, optional bias) backward: dx per row, d gamma / d beta as column sums;
//   * exact (erf) GELU forward / backward; cross entropy (ignore_index -100, mean over valid targets) forward / backward;
//   * embedding-table, bias, positional-table and patch-embedding gradients as deterministic reductions (no float atomics:
//     two runs give bit-identical gradients);
//   * transposes with zero padding, which turn dX = dY . W and dW = dY^T . X into the library's one GEMM form
//     C = A[M,K] . W[N,K]^T (csrc/gemm.hip, gemm256.hip: f32-exact MFMA in the parity mode, bf16 MFMA in the perf mode);
//   * global gradient norm + fused AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias correction, eps outside
//     the square root), with the clip factor of torch.nn.utils.clip_grad_norm_ folded in.
// All kernels are plain f32 VALU work with LDS tiles: the training step's flops are in the GEMMs.
#include "common.hpp"
#include "train.hpp"

namespace dimx {
namespace {

constexpr float kNegMax = -3.4028234663852886e38f;  // -torch.finfo(float32).max: x-transformers' mask fill value

// ------------------------------------------------------------------------------------------------ transposes / casts
// out[c][r] = in[r][c] for r < R, c < C; columns R..ld_out-1 of every output row are zero (the GEMM's K padding)
template <typename OutT>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ in, int ld_in, OutT* __restrict__ out, int ld_out,
                                                            int R, int C) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < C && r < ld_out) store_from_f32<OutT>(out + (size_t)c * ld_out + r, tile[tx][ty + 8 * i]);
    }
}

// Both operand copies of a group of weight matrices in one launch (a step makes 91 x 2 of them: as separate launches of
// ~7 us they were 1.3 ms of a 19 ms step).  One block per 32 x 32 tile of the padded [Np][Kp] extent of its matrix.
template <typename OutT> __device__ __forceinline__ void prep_tile(const PrepDesc& d, int lt, float (&tile)[32][33]) {
    const int tn = lt / d.tiles_k, tk = lt % d.tiles_k;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    OutT* w = (OutT*)d.w;
    OutT* wt = (OutT*)d.wt;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = tn * 32 + ty + 8 * r, k = tk * 32 + tx;
        float v = (n < d.N && k < d.K) ? d.src[(size_t)n * d.lds + k] : 0.f;
        if (d.gelu) v = 0.5f * v * (1.0f + erff(v * 0.7071067811865476f));   // the copies hold gelu(src): the f32 activation is never stored
        tile[ty + 8 * r][tx] = v;
        if (n < d.N && k < d.Kp) store_from_f32<OutT>(w + (size_t)n * d.Kp + k, v);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = tk * 32 + ty + 8 * r, n = tn * 32 + tx;
        if (k < d.K && n < d.Np) store_from_f32<OutT>(wt + (size_t)k * d.Np + n, tile[tx][ty + 8 * r]);
    }
}
template <typename OutT>
__global__ __launch_bounds__(256) void multi_prep_kernel(PrepTable t) {
    __shared__ float tile[32][33];
    int i = 0;
    while (i + 1 < t.n && (int)blockIdx.x >= t.d[i + 1].tile0) ++i;
    prep_tile<OutT>(t.d[i], blockIdx.x - t.d[i].tile0, tile);
}
// one matrix (an activation): cast + transposed cast in one pass over the source
template <typename OutT>
__global__ __launch_bounds__(256) void prep_pair_kernel(PrepDesc d) {
    __shared__ float tile[32][33];
    prep_tile<OutT>(d, blockIdx.x, tile);
}

// Round 4: the operand copies of an ACTIVATION, with the producer's elementwise work inside.  One pass over the f32 source yields the
// cast [rows][Kp], the transposed cast [cols][Mp] and (optionally) the column sums of what was cast -- what used to be up to four
// launches (LayerNorm | GELU' | bias column sums, then prep_pair) with an f32 round trip between them:
//   mode 0  v = src                              (gradients arriving from an attention / LayerNorm adjoint)
//   mode 1  v = erf-GELU(src)                    (the feed-forward activation as ff2's operand)
//   mode 2  v = LayerNorm(src) * gamma           (the pre-norm of a sublayer as its projection's operand; the f32 y is never stored)
//   mode 3  v = src * GELU'(src2)                (d pre-activation as ff1's adjoint operand; colpart gives d bias of ff1)
//   mode 4 / 5: modes 1 / 3 with the tanh form of GELU (the VQ-VAE's MLP, code/models/lib/base_models.py:107-123)
// (mode 2 with beta: LayerNorm with bias, the VQ-VAE's pre-norms)
// Block = 32 rows x `chunk` columns, 64 columns at a time: float4 loads, 8 / 16-byte stores on both copies (prep_tile moved 4 and 2
// bytes per thread).  Rows in [rows, Mp) of the transposed copy are written as zeros (the dW contraction runs over them).
template <typename OutT> struct Out4;
template <> struct Out4<bf16> {
    static __device__ __forceinline__ void store(bf16* p, float a, float b, float c, float d) {
        *(uint2*)p = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
    }
};
template <> struct Out4<float> {
    static __device__ __forceinline__ void store(float* p, float a, float b, float c, float d) { *(float4*)p = make_float4(a, b, c, d); }
};
template <typename OutT>
__global__ __launch_bounds__(256) void prep_fused_kernel(PrepFused d) {
    __shared__ float tile[32][65];
    __shared__ float mean_s[32], rstd_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 32;
    const int c_begin = blockIdx.y * d.chunk, c_end = min(d.Kp, c_begin + d.chunk);
    if (d.mode == 2) {   // row statistics of this block's rows: two passes over the row (the form layernorm_kernel uses)
        for (int rr = 0; rr < 8; ++rr) {
            const int r = wave * 8 + rr, row = r0 + r;
            if (row >= d.rows) break;
            const float4* x4 = (const float4*)(d.src + (size_t)row * d.lds);
            const int n4 = d.cols >> 2;
            float sm = 0.f;
            for (int i = lane; i < n4; i += 64) {
                const float4 v = x4[i];
                sm += (v.x + v.y) + (v.z + v.w);
            }
            const float mean = wave_sum(sm) / (float)d.cols;
            float q = 0.f;
            for (int i = lane; i < n4; i += 64) {
                const float4 v = x4[i];
                const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
                q += (a * a + b * b) + (c * c + e * e);
            }
            const float rstd = rsqrtf(wave_sum(q) / (float)d.cols + 1e-5f);
            if (lane == 0) {
                mean_s[r] = mean;
                rstd_s[r] = rstd;
            }
        }
        __syncthreads();
    }
    OutT* o = (OutT*)d.o;
    OutT* t = (OutT*)d.t;
    const int ty = tid >> 4, tx = tid & 15;
    // the next 64 columns are requested before the current ones are worked on (a block walks its columns in order: without the
    // prefetch every step paid a full memory round trip -- 18 of them for a 1152-wide LayerNorm row)
    float4 xa[2], za[2], xb[2], zb[2];
    auto fetch = [&](int c0, float4 (&x)[2], float4 (&z)[2]) {
        const int col = c0 + 4 * tx;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = r0 + ty + 16 * h;
            x[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[h] = x[h];
            if (c0 < c_end && row < d.rows && col < d.cols) {   // cols % 4 == 0 (checked by the launcher): whole groups only
                x[h] = *(const float4*)(d.src + (size_t)row * d.lds + col);
                if (d.mode == 3 || d.mode == 5) z[h] = *(const float4*)(d.src2 + (size_t)row * d.lds2 + col);
            }
        }
    };
    fetch(c_begin, xa, za);
    for (int c0 = c_begin; c0 < c_end; c0 += 64) {
        const int col = c0 + 4 * tx;
        fetch(c0 + 64, xb, zb);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = ty + 16 * h, row = r0 + r;
            float v[4] = {xa[h].x, xa[h].y, xa[h].z, xa[h].w};
            if (row < d.rows && col < d.cols) {
                if (d.mode == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.7071067811865476f));
                } else if (d.mode == 2) {
                    const float4 g = *(const float4*)(d.gamma + col);
                    const float m = mean_s[r], rs = rstd_s[r];
                    v[0] = (v[0] - m) * rs * g.x; v[1] = (v[1] - m) * rs * g.y; v[2] = (v[2] - m) * rs * g.z; v[3] = (v[3] - m) * rs * g.w;
                    if (d.beta) {
                        const float4 bt = *(const float4*)(d.beta + col);
                        v[0] += bt.x; v[1] += bt.y; v[2] += bt.z; v[3] += bt.w;
                    }
                } else if (d.mode == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = 0.7978845608028654f * (v[e] + 0.044715f * v[e] * v[e] * v[e]);
                        v[e] = 0.5f * v[e] * (1.0f + tanhf(u));
                    }
                } else if (d.mode == 5) {
                    const float z[4] = {za[h].x, za[h].y, za[h].z, za[h].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // d/dz [0.5 z (1 + tanh u)], u = c (z + 0.044715 z^3)
                        const float u = 0.7978845608028654f * (z[e] + 0.044715f * z[e] * z[e] * z[e]);
                        const float th = tanhf(u);
                        const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * z[e] * z[e]);
                        v[e] *= 0.5f * (1.0f + th) + 0.5f * z[e] * (1.0f - th * th) * du;
                    }
                } else if (d.mode == 3) {
                    const float z[4] = {za[h].x, za[h].y, za[h].z, za[h].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // d/dz [0.5 z (1 + erf(z / sqrt 2))] = 0.5 (1 + erf(z / sqrt 2)) + z exp(-z^2 / 2) / sqrt(2 pi)
                        const float cdf = 0.5f * (1.0f + erff(z[e] * 0.7071067811865476f));
                        v[e] *= cdf + z[e] * 0.3989422804014327f * expf(-0.5f * z[e] * z[e]);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][4 * tx + e] = v[e];
            if (row < d.rows && col < d.Kp) Out4<OutT>::store(o + (size_t)row * d.Kp + col, v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        if (d.colpart && tid < 64 && c0 + tid < d.cols && r0 < d.rows) {   // fixed order: rows 0..31 of the block
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) a += tile[r][tid];
            d.colpart[(size_t)blockIdx.x * d.cols + c0 + tid] = a;
        }
        {
            const int c = tid >> 2, part = tid & 3, colT = c0 + c;
            if (colT < d.cols) {
                OutT* dst = t + (size_t)colT * d.Mp + r0 + 8 * part;   // r0 + 32 <= Mp (grid.x = Mp / 32)
                Out4<OutT>::store(dst, tile[8 * part][c], tile[8 * part + 1][c], tile[8 * part + 2][c], tile[8 * part + 3][c]);
                Out4<OutT>::store(dst + 4, tile[8 * part + 4][c], tile[8 * part + 5][c], tile[8 * part + 6][c], tile[8 * part + 7][c]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xa[h] = xb[h];
            za[h] = zb[h];
        }
    }
}

// ------------------------------------------------------------------------------------------------ attention
// One wave per (clip, head, query row).  Keys are dealt to the lanes (j = lane, lane + 64, ...); q, the output row and
// the gradient rows live in registers, K / V rows stream from L2 (a head's K and V are 2 x Lk x 256 B).
struct AttnShape {
    int B, H, Lq, Lk;       // D = 64
    int ldq, ldk, ldv, ldo; // row strides (elements) of the [B, L, *] operands; head h at column h * 64
    float scale;
    int causal;
    const uint8_t* kmask;   // [B, Lk] 1 = keep (padding mask), optional
    const uint8_t* kmask2;  // [B, Lk] second keep-mask (AutoregressiveWrapper's mask_prob draw), optional
};

__device__ __forceinline__ bool key_kept(const AttnShape& a, int b, int i, int j) {
    if (a.causal && j > i) return false;
    if (a.kmask && !a.kmask[(size_t)b * a.Lk + j]) return false;
    if (a.kmask2 && !a.kmask2[(size_t)b * a.Lk + j]) return false;
    return true;
}

__device__ __forceinline__ float dot64(const float (&q)[64], const float* __restrict__ row) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
        const float4 k4 = *(const float4*)(row + d);
        s = fmaf(q[d], k4.x, s);
        s = fmaf(q[d + 1], k4.y, s);
        s = fmaf(q[d + 2], k4.z, s);
        s = fmaf(q[d + 3], k4.w, s);
    }
    return s;
}

// sum over the 64 lanes of a 64-vector held as acc[64] per lane -> lane d gets element d (through a 16 KiB LDS tile)
__device__ __forceinline__ float lanes_reduce64(const float (&acc)[64], float* sm, int lane) {
#pragma unroll
    for (int d = 0; d < 64; ++d) sm[lane * 65 + d] = acc[d];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll 8
    for (int l = 0; l < 64; ++l) s += sm[l * 65 + lane];
    __builtin_amdgcn_wave_barrier();
    return s;
}

__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnShape a, const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float* __restrict__ o, float* __restrict__ lse) {
    __shared__ float sm[64 * 65];
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    float qr[64];
    {
        const float* qp = q + ((size_t)b * a.Lq + i) * a.ldq + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const float4 t = *(const float4*)(qp + d);
            qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w;
        }
    }
    const float* kb = k + (size_t)b * a.Lk * a.ldk + h * 64;
    const float* vb = v + (size_t)b * a.Lk * a.ldv + h * 64;
    // pass 1: row maximum (masked scores count as -FLT_MAX, like masked_fill before the softmax)
    float mx = kNegMax;
    for (int j = lane; j < a.Lk; j += 64) {
        const float s = key_kept(a, b, i, j) ? dot64(qr, kb + (size_t)j * a.ldk) * a.scale : kNegMax;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    // pass 2: probabilities and the output row
    float acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    float se = 0.f;
    for (int j = lane; j < a.Lk; j += 64) {
        const float s = key_kept(a, b, i, j) ? dot64(qr, kb + (size_t)j * a.ldk) * a.scale : kNegMax;
        const float p = expf(s - mx);
        se += p;
        const float* vr = vb + (size_t)j * a.ldv;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const float4 t = *(const float4*)(vr + d);
            acc[d] = fmaf(p, t.x, acc[d]);
            acc[d + 1] = fmaf(p, t.y, acc[d + 1]);
            acc[d + 2] = fmaf(p, t.z, acc[d + 2]);
            acc[d + 3] = fmaf(p, t.w, acc[d + 3]);
        }
    }
    se = wave_sum(se);
    const float od = lanes_reduce64(acc, sm, lane) / se;
    o[((size_t)b * a.Lq + i) * a.ldo + h * 64 + lane] = od;
    if (lane == 0) lse[((size_t)b * a.H + h) * a.Lq + i] = mx + logf(se);
}

// dQ row + delta_i = dO_i . O_i (kept for the dK / dV kernel)
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(AttnShape a, const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ o,
                                                         const float* __restrict__ d_o, const float* __restrict__ lse,
                                                         float* __restrict__ dq, int lddq, float* __restrict__ delta) {
    __shared__ float sm[64 * 65];
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    float qr[64], gr[64];
    float dl = 0.f;
    {
        const size_t ro = ((size_t)b * a.Lq + i);
        const float* qp = q + ro * a.ldq + h * 64;
        const float* gp = d_o + ro * a.ldo + h * 64;
        const float* op = o + ro * a.ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const float4 t = *(const float4*)(qp + d);
            const float4 g = *(const float4*)(gp + d);
            const float4 w = *(const float4*)(op + d);
            qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w;
            gr[d] = g.x; gr[d + 1] = g.y; gr[d + 2] = g.z; gr[d + 3] = g.w;
            dl += g.x * w.x + g.y * w.y + g.z * w.z + g.w * w.w;
        }
    }
    const float L = lse[((size_t)b * a.H + h) * a.Lq + i];
    const float* kb = k + (size_t)b * a.Lk * a.ldk + h * 64;
    const float* vb = v + (size_t)b * a.Lk * a.ldv + h * 64;
    float acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    for (int j = lane; j < a.Lk; j += 64) {
        const float* kr = kb + (size_t)j * a.ldk;
        // a masked key carries probability 0 and receives no gradient (autograd through masked_fill; the matrix-core kernels
        // do the same) -- expf(kNegMax - L) would be 1 for a row whose keys are ALL masked (L = -FLT_MAX + log n rounds to -FLT_MAX)
        const bool kept = key_kept(a, b, i, j);
        const float s = kept ? dot64(qr, kr) * a.scale : kNegMax;
        const float p = kept ? expf(s - L) : 0.f;
        const float dp = dot64(gr, vb + (size_t)j * a.ldv);
        const float ds = p * (dp - dl) * a.scale;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const float4 t = *(const float4*)(kr + d);
            acc[d] = fmaf(ds, t.x, acc[d]);
            acc[d + 1] = fmaf(ds, t.y, acc[d + 1]);
            acc[d + 2] = fmaf(ds, t.z, acc[d + 2]);
            acc[d + 3] = fmaf(ds, t.w, acc[d + 3]);
        }
    }
    dq[((size_t)b * a.Lq + i) * lddq + h * 64 + lane] = lanes_reduce64(acc, sm, lane);
    if (lane == 0) delta[((size_t)b * a.H + h) * a.Lq + i] = dl;
}

// dK and dV row of key j: the queries are dealt to the lanes
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(AttnShape a, const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv) {
    __shared__ float sm[64 * 65];
    const int lane = threadIdx.x;
    const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    float kr[64], vr[64];
    {
        const float* kp = k + ((size_t)b * a.Lk + j) * a.ldk + h * 64;
        const float* vp = v + ((size_t)b * a.Lk + j) * a.ldv + h * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const float4 t = *(const float4*)(kp + d);
            const float4 w = *(const float4*)(vp + d);
            kr[d] = t.x; kr[d + 1] = t.y; kr[d + 2] = t.z; kr[d + 3] = t.w;
            vr[d] = w.x; vr[d + 1] = w.y; vr[d + 2] = w.z; vr[d + 3] = w.w;
        }
    }
    float ak[64], av[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) ak[d] = av[d] = 0.f;
    const float* lp = lse + ((size_t)b * a.H + h) * a.Lq;
    const float* dp_ = delta + ((size_t)b * a.H + h) * a.Lq;
    for (int i = lane; i < a.Lq; i += 64) {
        const float* qp = q + ((size_t)b * a.Lq + i) * a.ldq + h * 64;
        const float* gp = d_o + ((size_t)b * a.Lq + i) * a.ldo + h * 64;
        const bool kept = key_kept(a, b, i, j);
        const float s = kept ? dot64(kr, qp) * a.scale : kNegMax;
        const float p = kept ? expf(s - lp[i]) : 0.f;   // masked: probability 0, no gradient (see attn_bwd_dq_kernel)
        const float dpv = dot64(vr, gp);
        const float ds = p * (dpv - dp_[i]) * a.scale;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const float4 t = *(const float4*)(qp + d);
            const float4 g = *(const float4*)(gp + d);
            ak[d] = fmaf(ds, t.x, ak[d]);
            ak[d + 1] = fmaf(ds, t.y, ak[d + 1]);
            ak[d + 2] = fmaf(ds, t.z, ak[d + 2]);
            ak[d + 3] = fmaf(ds, t.w, ak[d + 3]);
            av[d] = fmaf(p, g.x, av[d]);
            av[d + 1] = fmaf(p, g.y, av[d + 1]);
            av[d + 2] = fmaf(p, g.z, av[d + 2]);
            av[d + 3] = fmaf(p, g.w, av[d + 3]);
        }
    }
    const float rk = lanes_reduce64(ak, sm, lane);
    const float rv = lanes_reduce64(av, sm, lane);
    dk[((size_t)b * a.Lk + j) * lddk + h * 64 + lane] = rk;
    dv[((size_t)b * a.Lk + j) * lddv + h * 64 + lane] = rv;
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// dx[row] (+)= gamma o dy . rstd - xhat . mean(gamma o dy o xhat) . rstd - mean(gamma o dy) . rstd; one wave per row.
// xhat is recomputed from x (two exact passes, eps 1e-5).  ACCUM: add into dx (the residual stream's gradient).
template <bool ACCUM>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, float* __restrict__ dx, int M, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * C;
    const float* gy = dy + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / C + 1e-5f);
    float a1 = 0.f, a2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float g = gamma[c] * gy[c];
        a1 += g;
        a2 += g * (xr[c] - mean) * rstd;
    }
    a1 = wave_sum(a1) / C;
    a2 = wave_sum(a2) / C;
    float* dr = dx + (size_t)row * C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd;
        const float t = (gamma[c] * gy[c] - a1 - xh * a2) * rstd;
        dr[c] = ACCUM ? dr[c] + t : t;
    }
}

// column sums over rows of dy (d beta, bias gradients) and of dy o xhat (d gamma): block = 64 columns x 4 row lanes, rows are
// split over gridDim.y slabs; a second pass (colsum_finish) adds the slabs in a fixed order
__global__ __launch_bounds__(256) void ln_colsums_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part_g,
                                                         float* __restrict__ part_b, int M, int C, int rows_per_slab) {
    __shared__ float sm[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r_begin = blockIdx.y * rows_per_slab;
    const int r_end = min(M, r_begin + rows_per_slab);
    float sg = 0.f, sb = 0.f;
    if (c < C) {
        for (int r = r_begin + rl; r < r_end; r += 4) {
            // `x` holds xhat (tr_xhat): recomputing the row statistics per column lane would be M x C x C work
            const float g = dy[(size_t)r * C + c];
            sb += g;
            if (x) sg += g * x[(size_t)r * C + c];
        }
    }
    sm[0][rl][threadIdx.x & 63] = sg;
    sm[1][rl][threadIdx.x & 63] = sb;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int l = threadIdx.x & 63;
        if (part_g) part_g[(size_t)blockIdx.y * C + c] = (sm[0][0][l] + sm[0][1][l]) + (sm[0][2][l] + sm[0][3][l]);
        if (part_b) part_b[(size_t)blockIdx.y * C + c] = (sm[1][0][l] + sm[1][1][l]) + (sm[1][2][l] + sm[1][3][l]);
    }
}
__global__ void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int nslab, float scale, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int i = 0; i < nslab; ++i) s += part[(size_t)i * C + c];
    out[c] = accumulate ? out[c] + s * scale : s * scale;
}

// Round 4: ONE launch per LayerNorm adjoint.  dx (+)= LN'(x) dy as layernorm_bwd_kernel, and the same pass leaves the column
// sums of dy o xhat (d gamma) and dy (d beta): a wave owns one row at a time and keeps its columns' partial sums in registers over
// the rows of its block; the four waves of a block are added in wave order into partial row [block][C]; the partial rows of ALL the
// step's column reductions are added, in row order, by one multi_finish launch at the end of the backward pass (a fixed
// summation order, no float atomics).  (Before: xhat_kernel + ln_colsums_kernel + colsum_finish_kernel + layernorm_bwd_kernel = 4-5 launches, 2 extra passes.)
// NC = C / 64 columns per lane (6: the 384-wide encoders, 18: the 1152-wide decoder)
template <bool ACCUM, int kLnCols>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int M, int C, int rows_per_block, float* __restrict__ part_g,
                                                           float* __restrict__ part_b) {
    __shared__ float sm[2][4][64 * kLnCols];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    constexpr int nc = kLnCols;
    float pg[kLnCols], pb[kLnCols];
#pragma unroll
    for (int i = 0; i < kLnCols; ++i) pg[i] = pb[i] = 0.f;
    for (int row = r0 + wave; row < r1; row += 4) {
        const float* xr = x + (size_t)row * C;
        const float* gy = dy + (size_t)row * C;
        float xv[kLnCols], gv[kLnCols];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kLnCols; ++i)
            if (i < nc) {
                xv[i] = xr[lane + 64 * i];
                gv[i] = gy[lane + 64 * i];
                s += xv[i];
            }
        const float mean = wave_sum(s) / C;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < kLnCols; ++i)
            if (i < nc) {
                const float d = xv[i] - mean;
                v += d * d;
            }
        const float rstd = rsqrtf(wave_sum(v) / C + 1e-5f);
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int i = 0; i < kLnCols; ++i)
            if (i < nc) {
                const float xh = (xv[i] - mean) * rstd;
                const float g = gamma[lane + 64 * i] * gv[i];
                a1 += g;
                a2 += g * xh;
                pg[i] += gv[i] * xh;
                pb[i] += gv[i];
                xv[i] = xh;
            }
        a1 = wave_sum(a1) / C;
        a2 = wave_sum(a2) / C;
        float* dr = dx + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < kLnCols; ++i)
            if (i < nc) {
                const float t = (gamma[lane + 64 * i] * gv[i] - a1 - xv[i] * a2) * rstd;
                dr[lane + 64 * i] = ACCUM ? dr[lane + 64 * i] + t : t;
            }
    }
#pragma unroll
    for (int i = 0; i < kLnCols; ++i)
        if (i < nc) {
            sm[0][wave][lane + 64 * i] = pg[i];
            sm[1][wave][lane + 64 * i] = pb[i];
        }
    __syncthreads();
    float* pgrow = part_g + (size_t)blockIdx.x * C;
    float* pbrow = part_b ? part_b + (size_t)blockIdx.x * C : nullptr;
    for (int c = threadIdx.x; c < C; c += 256) {
        pgrow[c] = (sm[0][0][c] + sm[0][1][c]) + (sm[0][2][c] + sm[0][3][c]);
        if (pbrow) pbrow[c] = (sm[1][0][c] + sm[1][1][c]) + (sm[1][2][c] + sm[1][3][c]);
    }
}

// every pending column reduction of the step in one launch: block -> (entry, 64 columns); out[c] = sum over the entry's partial rows
__global__ __launch_bounds__(256) void multi_finish_kernel(FinTable t) {
    __shared__ float sm[4][64];
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.d[e + 1].blk0) ++e;
    const FinDesc d = t.d[e];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = ((int)blockIdx.x - d.blk0) * 64 + l;
    float s0 = 0.f;
    if (c < d.C)
        for (int i = rl; i < d.nslab; i += 4) s0 += d.part[(size_t)i * d.C + c];
    sm[rl][l] = s0;
    __syncthreads();
    if (rl == 0 && c < d.C) d.out[c] = (sm[0][l] + sm[1][l]) + (sm[2][l] + sm[3][l]);
}

// xhat[row] = (x - mean) * rstd (needed once per LayerNorm for d gamma)
__global__ __launch_bounds__(256) void xhat_kernel(const float* __restrict__ x, float* __restrict__ xh, int M, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / C + 1e-5f);
    for (int c = lane; c < C; c += 64) xh[(size_t)row * C + c] = (xr[c] - mean) * rstd;
}

// ------------------------------------------------------------------------------------------------ elementwise
__global__ void gelu_erf_fwd_kernel(const float* __restrict__ pre, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = pre[i];
        out[i] = 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
    }
}
// dpre = dh * (Phi(x) + x phi(x))
__global__ void gelu_erf_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dh, float* __restrict__ dpre, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = pre[i];
        const float cdf = 0.5f * (1.0f + erff(x * 0.7071067811865476f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
        dpre[i] = dh[i] * (cdf + x * pdf);
    }
}
__global__ void add_kernel(float* __restrict__ y, const float* __restrict__ a, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a[i];
}
// y[m, c] = a[m, c] + row[c] (+ table[m % T, c] * scale): patch-embedding add / positional rows
__global__ void add_rows_kernel(const float* __restrict__ a, int lda, const float* __restrict__ row, const float* __restrict__ table,
                                float scale, int T, float* __restrict__ y, int ldy, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        float v = a[(size_t)m * lda + c];
        if (row) v += row[c];
        if (table) v += table[(size_t)(m % T) * C + c] * scale;
        y[(size_t)m * ldy + c] = v;
    }
}
// rows whose mask byte is 0 are zeroed (x-transformers zero-fills the attention output of padded queries)
__global__ void zero_rows_kernel(float* __restrict__ y, const uint8_t* __restrict__ keep, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C);
        if (!keep[m]) y[i] = 0.f;
    }
}
// dst[m, 0:C) (+)= src[m, 0:C) with different row strides (context slice / concat)
__global__ void copy_cols_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int M, int C, int accumulate) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        const float v = src[(size_t)m * lds_ + c];
        float* d = dst + (size_t)m * ldd + c;
        *d = accumulate ? *d + v : v;
    }
}
// table[t, c] gradient of "x[b, t] += table[t] * scale": sum over clips, deterministic
__global__ void pos_grad_kernel(const float* __restrict__ dx, float* __restrict__ dtable, int B, int T, int C, float scale, int accumulate) {
    const long total = (long)T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dx[(size_t)b * T * C + i];
        dtable[i] = accumulate ? dtable[i] + s * scale : s * scale;
    }
}

// ------------------------------------------------------------------------------------------------ cross entropy
// one wave per row of 512 logits: loss row (0 where the target is ignored) and d logits = (softmax - onehot) * gscale
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ target,
                                                         float* __restrict__ row_loss, float* __restrict__ dlogits, int R,
                                                         const float* __restrict__ inv_count) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* lr = logits + (size_t)row * 512;
    float v[8];
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v[c] = lr[lane + 64 * c];
        mx = fmaxf(mx, v[c]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) se += expf(v[c] - mx);
    se = wave_sum(se);
    const int t = target[row];
    const bool valid = t >= 0 && t < 512;
    const float g = valid ? *inv_count : 0.f;
    if (lane == 0) row_loss[row] = valid ? (mx + logf(se)) - lr[t] : 0.f;
    const float inv = 1.0f / se;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = lane + 64 * c;
        dlogits[(size_t)row * 512 + col] = g * (expf(v[c] - mx) * inv - (col == t ? 1.0f : 0.f));
    }
}
// out[0] = sum(row_loss) / count, out[1] = 1 / count (count = valid targets, clamped to 1); one block
__global__ __launch_bounds__(256) void ce_count_kernel(const int32_t* __restrict__ target, int R, float* __restrict__ out) {
    __shared__ int sm[256];
    int n = 0;
    for (int i = threadIdx.x; i < R; i += 256) n += (target[i] >= 0 && target[i] < 512) ? 1 : 0;
    sm[threadIdx.x] = n;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[1] = 1.0f / (float)(sm[0] > 0 ? sm[0] : 1);
}
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ x, int n, const float* __restrict__ scale, float* __restrict__ out) {
    __shared__ float sm[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0] * scale[0];
}

// Transposed one-hot operand of the token-embedding gradient: out[v][m] = (tokens[m] == v), zero in the padding columns
// m >= M.  d table = onehot^T . dx is then one more GEMM of the library (the same form as every dW = dy^T . x): its cost does
// not depend on how the tokens are distributed (a scan-and-add kernel per vocabulary row took 1.3 - 2.1 ms when a few codes
// dominate) and the summation order is fixed.
template <typename OutT>
__global__ __launch_bounds__(256) void onehot_t_kernel(const int32_t* __restrict__ tokens, OutT* __restrict__ out, int ld_out, int M) {
    const int v = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= ld_out) return;
    store_from_f32<OutT>(out + (size_t)v * ld_out + m, (m < M && tokens[m] == v) ? 1.f : 0.f);
}

// ------------------------------------------------------------------------------------------------ optimiser
// partial sums of squares over the flat gradient arena (fixed block partition -> deterministic), then the norm
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
    __shared__ double sm[256];
    // 16-byte loads, four of them in flight per thread (the scalar-load form ran at 1.2 TB/s: 312 us for 93 M gradients);
    // 1e8 squares: f64 partial sums keep the norm (and with it the clip factor) exact to f32 rounding
    const long n4 = n >> 2;
    const float4* __restrict__ g4 = (const float4*)g;
    const long stride = (long)gridDim.x * 256;
    double s0 = 0.0, s1 = 0.0;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
        s0 += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
        s1 += (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z + (double)b.w * b.w;
        s0 += (double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z + (double)c.w * c.w;
        s1 += (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w;
    }
    for (; i < n4; i += stride) {
        const float4 a = g4[i];
        s0 += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float t = g[(n4 << 2) + threadIdx.x];
        s1 += (double)t * t;
    }
    sm[threadIdx.x] = s0 + s1;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
// out[0] = ||g||, out[1] = clip coefficient min(1, max_norm / (||g|| + 1e-6)) (1 when max_norm <= 0)
__global__ __launch_bounds__(256) void norm_finish_kernel(const double* __restrict__ part, int nblocks, float max_norm, float* __restrict__ out) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += part[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float nrm = (float)sqrt(sm[0]);
        out[0] = nrm;
        out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (nrm + 1e-6f)) : 1.0f;
    }
}
// torch.optim.AdamW: p *= 1 - lr wd; m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ clip) {
    const float cf = clip ? clip[1] : 1.0f;
    const float decay = 1.0f - lr * wd, step = lr / bc1;
    auto one = [&](float gi, float& pi, float& mi, float& vi) {
        gi *= cf;
        pi *= decay;
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    };
    const long n4 = n >> 2;   // the arenas are 16-byte aligned (dimx_train_forward_backward checks it)
    float4* __restrict__ p4 = (float4*)p;
    float4* __restrict__ m4 = (float4*)m;
    float4* __restrict__ v4 = (float4*)v;
    const float4* __restrict__ g4 = (const float4*)g;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 gi = g4[i];
        float4 pi = p4[i], mi = m4[i], vi = v4[i];
        one(gi.x, pi.x, mi.x, vi.x);
        one(gi.y, pi.y, mi.y, vi.y);
        one(gi.z, pi.z, mi.z, vi.z);
        one(gi.w, pi.w, mi.w, vi.w);
        m4[i] = mi;
        v4[i] = vi;
        p4[i] = pi;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        float pi = p[i], mi = m[i], vi = v[i];
        one(g[i], pi, mi, vi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
}

// ------------------------------------------------------------------------------------------------ VQ-VAE decoder (trainable in
// the legacy and SLM loops: code/models/stage1_BIWI.py:376-393) and the glue of the legacy generator's loss
// Conv1d(k = 5, replicate padding) as a GEMM: X5[m][c * 5 + tap] = x[clip(m), clamp(t + tap - 2)][c] -- the column order of the
// weight's own [out][in][5] layout, so the weight and its gradient are used as they lie in the arena
__global__ void im2col5_kernel(const float* __restrict__ x, float* __restrict__ x5, int B, int n, int C) {
    const long total = (long)B * n * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long m = i / C;
        const int t = (int)(m % n);
        const long base = m - t;
#pragma unroll
        for (int tap = 0; tap < 5; ++tap) {
            int ts = t + tap - 2;
            ts = ts < 0 ? 0 : (ts >= n ? n - 1 : ts);
            x5[(size_t)m * 5 * C + (size_t)c * 5 + tap] = x[(size_t)(base + ts) * C + c];
        }
    }
}
// its adjoint: dx[t'] = sum over (t, tap) with clamp(t + tap - 2) == t' of dX5[t][c * 5 + tap] (gather form: no atomics)
__global__ void col2im5_kernel(const float* __restrict__ dx5, float* __restrict__ dx, int B, int n, int C) {
    const long total = (long)B * n * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long m = i / C;
        const int tp = (int)(m % n);
        const long base = m - tp;
        float a = 0.f;
        const int lo = tp == 0 ? 0 : tp - 2, hi = tp == n - 1 ? n - 1 : tp + 2;
        for (int t = (lo < 0 ? 0 : lo); t <= (hi >= n ? n - 1 : hi); ++t)
#pragma unroll
            for (int tap = 0; tap < 5; ++tap) {
                int ts = t + tap - 2;
                ts = ts < 0 ? 0 : (ts >= n ? n - 1 : ts);
                if (ts == tp) a += dx5[(size_t)(base + t) * 5 * C + (size_t)c * 5 + tap];
            }
        dx[i] = a;
    }
}
// y = InstanceNorm_t(LeakyReLU_0.2(x)) per (clip, channel) over the n frames (biased variance, eps 1e-5, no affine).
// One thread per channel (rows are channel-contiguous), two passes over time like F.instance_norm.
__global__ __launch_bounds__(64) void lrelu_inorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int C) {
    const int b = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x;
    if (c >= C) return;
    const float* xp = x + (size_t)b * n * C + c;
    float sm = 0.f;
    for (int t = 0; t < n; ++t) {
        const float v = xp[(size_t)t * C];
        sm += v > 0.f ? v : 0.2f * v;
    }
    const float mean = sm / (float)n;
    float q = 0.f;
    for (int t = 0; t < n; ++t) {
        const float v = xp[(size_t)t * C];
        const float a = (v > 0.f ? v : 0.2f * v) - mean;
        q += a * a;
    }
    const float rstd = rsqrtf(q / (float)n + 1e-5f);
    float* yp = y + (size_t)b * n * C + c;
    for (int t = 0; t < n; ++t) {
        const float v = xp[(size_t)t * C];
        yp[(size_t)t * C] = ((v > 0.f ? v : 0.2f * v) - mean) * rstd;
    }
}
// dx = LeakyReLU'(x) * rstd * (dy - mean_t(dy) - yhat * mean_t(dy * yhat))
__global__ __launch_bounds__(64) void lrelu_inorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int n,
                                                             int C) {
    const int b = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x;
    if (c >= C) return;
    const float* xp = x + (size_t)b * n * C + c;
    const float* gp = dy + (size_t)b * n * C + c;
    float sm = 0.f;
    for (int t = 0; t < n; ++t) {
        const float v = xp[(size_t)t * C];
        sm += v > 0.f ? v : 0.2f * v;
    }
    const float mean = sm / (float)n;
    float q = 0.f;
    for (int t = 0; t < n; ++t) {
        const float v = xp[(size_t)t * C];
        const float a = (v > 0.f ? v : 0.2f * v) - mean;
        q += a * a;
    }
    const float rstd = rsqrtf(q / (float)n + 1e-5f);
    float g1 = 0.f, g2 = 0.f;
    for (int t = 0; t < n; ++t) {
        const float v = xp[(size_t)t * C], g = gp[(size_t)t * C];
        const float yh = ((v > 0.f ? v : 0.2f * v) - mean) * rstd;
        g1 += g;
        g2 += g * yh;
    }
    g1 /= (float)n;
    g2 /= (float)n;
    float* dp = dx + (size_t)b * n * C + c;
    for (int t = 0; t < n; ++t) {
        const float v = xp[(size_t)t * C], g = gp[(size_t)t * C];
        const float yh = ((v > 0.f ? v : 0.2f * v) - mean) * rstd;
        dp[(size_t)t * C] = (v > 0.f ? 1.0f : 0.2f) * rstd * (g - g1 - yh * g2);
    }
}
// y[m, c] = a[m, c] + rows[m / n, c]: the VQ decoder's positional buffer is indexed by the CLIP (pe[:B], the reference's quirk)
__global__ void add_clip_rows_kernel(const float* __restrict__ a, const float* __restrict__ rows, float* __restrict__ y, int M, int n, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        y[i] = a[i] + rows[(size_t)(m / n) * C + c];
    }
}
// idx[b, t] = argmax_c logits[b, t + t_off, c] (first maximum), t < n_out; logits rows are [B, n_in, 512].  One wave per row.
__global__ __launch_bounds__(256) void argmax512_kernel(const float* __restrict__ logits, int32_t* __restrict__ idx, int B, int n_in, int n_out,
                                                        int t_off) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * n_out) return;
    const int b = row / n_out, t = row - b * n_out;
    const float* lr = logits + ((size_t)b * n_in + t + t_off) * 512;
    float best = -3.4e38f;
    int bi = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {   // ascending columns per lane: '>' keeps the first maximum
        const float v = lr[lane + 64 * c];
        if (v > best) {
            best = v;
            bi = lane + 64 * c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) idx[row] = bi;
}
// continuous loss of the decoded motion (code/seq2seq.py:270-276): rows (b, t) with mask[b, t + 1]; per selected row
// n_exp = ||d[6:]||, n_jaw = ||d[:6]||, d = pred - target + 1e-6 (F.pairwise_distance's eps sits inside the norm);
// loss = mean n_exp + mean n_jaw.  Pass 1: per-row norms and the unscaled gradient d / norm; pass 2 (one block): sums, count.
__global__ __launch_bounds__(256) void cont_rows_kernel(const float* __restrict__ pred, const float* __restrict__ v_tgt, const uint8_t* __restrict__ mask,
                                                        int B, int T, int n, float* __restrict__ rown, float* __restrict__ dpred) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * n) return;
    const int b = row / n, t = row - b * n;
    const bool sel = mask[(size_t)b * T + t + 1] != 0;
    float d = 0.f;
    if (lane < 56 && sel) d = pred[(size_t)row * 56 + lane] - v_tgt[((size_t)b * T + t + 1) * 56 + lane] + 1e-6f;
    const float s_jaw = wave_sum(lane < 6 ? d * d : 0.f), s_exp = wave_sum(lane >= 6 ? d * d : 0.f);
    const float n_jaw = sqrtf(s_jaw), n_exp = sqrtf(s_exp);
    if (lane == 0) {
        rown[2 * row] = sel ? n_exp : 0.f;
        rown[2 * row + 1] = sel ? n_jaw : 0.f;
    }
    if (lane < 56) {
        const float nn = lane < 6 ? n_jaw : n_exp;
        dpred[(size_t)row * 56 + lane] = (sel && nn > 0.f) ? d / nn : 0.f;
    }
}
// out[0] = mean n_exp + mean n_jaw, out[1] = 1 / count
__global__ __launch_bounds__(256) void cont_finish_kernel(const float* __restrict__ rown, const uint8_t* __restrict__ mask, int B, int T, int n,
                                                          float* __restrict__ out) {
    __shared__ float sa[256];
    __shared__ int sc[256];
    float a = 0.f;
    int cnt = 0;
    for (int i = threadIdx.x; i < B * n; i += 256) {
        const int b = i / n, t = i - b * n;
        a += rown[2 * i] + rown[2 * i + 1];
        cnt += mask[(size_t)b * T + t + 1] ? 1 : 0;
    }
    sa[threadIdx.x] = a;
    sc[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sa[threadIdx.x] += sa[threadIdx.x + s];
            sc[threadIdx.x] += sc[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float inv = 1.0f / (float)(sc[0] > 0 ? sc[0] : 1);
        out[0] = sa[0] * inv;
        out[1] = inv;
    }
}
__global__ void scale_by_kernel(float* __restrict__ y, const float* __restrict__ factor, long n) {
    const float f = *factor;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] *= f;
}
// head-padded operand layouts of the VQ attention (8 heads of 48 run on the 64-wide attention kernels as zero-padded heads):
// dst row g * 64 + j <- src row g * 48 + j (j < 48), zero otherwise; `cols` contiguous columns per row.  unpad: the inverse copy.
__global__ void pad_head_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int groups, int cols, int unpad) {
    const long total = (long)groups * 64 * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const int r = (int)(i / cols), g = r >> 6, j = r & 63;
        if (unpad) {
            if (j < 48) dst[((size_t)g * 48 + j) * cols + c] = src[i];
        } else {
            dst[i] = j < 48 ? src[((size_t)g * 48 + j) * cols + c] : 0.f;
        }
    }
}
// the same along the columns: dst[r][g * 64 + j] <- src[r][g * 48 + j]
__global__ void pad_head_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int groups, int unpad) {
    const long total = (long)rows * groups * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % (groups * 64)), r = (int)(i / (groups * 64));
        const int g = cc >> 6, j = cc & 63;
        if (unpad) {
            if (j < 48) dst[(size_t)r * groups * 48 + g * 48 + j] = src[i];
        } else {
            dst[i] = j < 48 ? src[(size_t)r * groups * 48 + g * 48 + j] : 0.f;
        }
    }
}
// id conditioning (code/seq2seq.py:240-252): e[b] = relu(table[ids[b]]); adjoint: dtable[ids[b]] += de[b] * (table > 0), clips in
// order (two clips may carry the same id: a fixed order instead of atomics)
__global__ void emb_relu_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, float* __restrict__ e, int B, int E) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * E; i += gridDim.x * blockDim.x) {
        const int b = i / E, j = i - b * E;
        const float v = table[(size_t)ids[b] * E + j];
        e[i] = v > 0.f ? v : 0.f;
    }
}
__global__ void emb_relu_bwd_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, const float* __restrict__ de,
                                    float* __restrict__ dtable, int B, int E) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x)
        for (int b = 0; b < B; ++b) {
            const size_t o = (size_t)ids[b] * E + j;
            if (table[o] > 0.f) dtable[o] += de[(size_t)b * E + j];
        }
}
// context with the id row in front: ctx[b, 0] = lid[b], ctx[b, 1 + t] = enc[b, t]; mask likewise (1 in front); split: the adjoint
__global__ void prepend_row_kernel(const float* __restrict__ first, const float* __restrict__ rest, float* __restrict__ out, int B, int T, int C,
                                   int split) {
    const long total = (long)B * (T + 1) * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;
        const int b = (int)(r / (T + 1)), t = (int)(r - (long)b * (T + 1));
        if (!split) {
            out[i] = t == 0 ? first[(size_t)b * C + c] : rest[((size_t)b * T + t - 1) * C + c];
        } else {   // out = the joint gradient (read), first / rest written
            if (t == 0) ((float*)first)[(size_t)b * C + c] = out[i];
            else ((float*)rest)[((size_t)b * T + t - 1) * C + c] = out[i];
        }
    }
}
// z_ext[b] = {-100, z[b, 0..T)}, m_ext[b] = {1, mask[b, 0..T)}
__global__ void prepend_tokens_kernel(const int32_t* __restrict__ z, const uint8_t* __restrict__ mask, int32_t* __restrict__ z_ext,
                                      uint8_t* __restrict__ m_ext, int B, int T) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * (T + 1); i += gridDim.x * blockDim.x) {
        const int b = i / (T + 1), t = i - b * (T + 1);
        z_ext[i] = t == 0 ? -100 : z[(size_t)b * T + t - 1];
        m_ext[i] = t == 0 ? 1 : mask[(size_t)b * T + t - 1];
    }
}
// zq[b, t] = codebook[idx[b, t]] (rows of 128 floats)
__global__ void gather128_kernel(const float* __restrict__ book, const int32_t* __restrict__ idx, float* __restrict__ out, int R) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * 128; i += (long)gridDim.x * blockDim.x)
        out[i] = book[(size_t)idx[i >> 7] * 128 + (i & 127)];
}

// ---- SLM pre-training glue (code/seq2seq_pretrain.py:200-323)
// dst[b][t][0:C) (+)= src[b][t][0:C) (+ addrow) with independent batch / row strides: the halves of x_joint [B, 2T, C], the
// concatenated context, and their adjoints
__global__ void copy_bt_kernel(const float* __restrict__ src, long src_bs, int src_ld, float* __restrict__ dst, long dst_bs, int dst_ld, int B,
                               int T, int C, const float* __restrict__ addrow, int accumulate) {
    const long total = (long)B * T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;
        const int t = (int)(r % T), b = (int)(r / T);
        float v = src[(size_t)b * src_bs + (size_t)t * src_ld + c];
        if (addrow) v += addrow[c];
        float* d = dst + (size_t)b * dst_bs + (size_t)t * dst_ld + c;
        *d = accumulate ? *d + v : v;
    }
}
// the same for mask bytes (invert: dst = !src, a "frame masked out" mask as the keep-mask of zero_rows)
__global__ void copy_bt_u8_kernel(const uint8_t* __restrict__ src, long src_bs, uint8_t* __restrict__ dst, long dst_bs, int B, int T, int invert) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * T; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T), b = (int)(i / T);
        const uint8_t v = src[(size_t)b * src_bs + t] != 0;
        dst[(size_t)b * dst_bs + t] = invert ? !v : v;
    }
}
// z[~sel] = -100 (:308-309): only the masked-out frames' codes are targets
__global__ void mask_tokens_kernel(const int32_t* __restrict__ z, const uint8_t* __restrict__ sel, int32_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = sel[i] ? z[i] : -100;
}

// InfoNCE between the clip means of x_s and x_l (forward_contrastive, :270-289).  Three launches:
//  1. mean[side][b] = mean of the first len_b rows (len_b = number of valid frames)
//  2. one block: F.normalize (eps 1e-12), total = s . l^T / 0.05, log-softmax over dim 0, nce = -mean diag, c_acc, and the adjoint
//     down to d mean (fixed summation order throughout)
//  3. d x[b, t] = d mean[b] / len_b for t < len_b
__global__ __launch_bounds__(128) void nce_mean_kernel(const float* __restrict__ xs, const float* __restrict__ xl, const uint8_t* __restrict__ mask,
                                                       int B, int T, int C, float* __restrict__ mean, int* __restrict__ lens) {
    const int b = blockIdx.x, side = blockIdx.y;
    __shared__ int s_len;
    if (threadIdx.x < 64) {
        int n = 0;
        for (int t = threadIdx.x; t < T; t += 64) n += mask[(size_t)b * T + t] ? 1 : 0;
        n = (int)wave_sum((float)n);
        if (threadIdx.x == 0) {
            s_len = n;
            if (side == 0) lens[b] = n;
        }
    }
    __syncthreads();
    const int len = s_len;
    const float* x = (side ? xl : xs) + (size_t)b * T * C;
    for (int c = threadIdx.x; c < C; c += 128) {
        float a = 0.f;
        for (int t = 0; t < len; ++t) a += x[(size_t)t * C + c];
        mean[((size_t)side * B + b) * C + c] = len > 0 ? a / (float)len : 0.f;
    }
}
// scr: sn [2][B][C] | nrm [2][B] | total [B][B] | dtot [B][B] | dsn [2][B][C]
__global__ __launch_bounds__(256) void nce_core_kernel(const float* __restrict__ mean, int B, int C, float* __restrict__ scr, float* __restrict__ dmean,
                                                       float* __restrict__ out) {
    float* sn = scr;
    float* nrm = sn + (size_t)2 * B * C;
    float* total = nrm + 2 * B;
    float* dtot = total + (size_t)B * B;
    float* dsn = dtot + (size_t)B * B;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ float red[256];
    for (int r = wave; r < 2 * B; r += 4) {          // one wave per (side, clip): the norm
        float a = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = mean[(size_t)r * C + c];
            a += v * v;
        }
        a = wave_sum(a);
        const float n = fmaxf(sqrtf(a), 1e-12f);
        if (lane == 0) nrm[r] = n;
        for (int c = lane; c < C; c += 64) sn[(size_t)r * C + c] = mean[(size_t)r * C + c] / n;
    }
    __syncthreads();
    for (int p = wave; p < B * B; p += 4) {           // total[i][j] = s_i . l_j / 0.05
        const int i = p / B, j = p - i * B;
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += sn[(size_t)i * C + c] * sn[((size_t)B + j) * C + c];
        a = wave_sum(a);
        if (lane == 0) total[p] = a / 0.05f;
    }
    __syncthreads();
    float loss = 0.f, hit = 0.f;
    for (int j = tid; j < B; j += 256) {              // column j: softmax over the rows i
        float mx = -3.0e38f;
        int am = 0;
        for (int i = 0; i < B; ++i) {
            const float v = total[(size_t)i * B + j];
            if (v > mx) {
                mx = v;
                am = i;
            }
        }
        float se = 0.f;
        for (int i = 0; i < B; ++i) se += expf(total[(size_t)i * B + j] - mx);
        const float lse = mx + logf(se);
        loss -= total[(size_t)j * B + j] - lse;
        hit += am == j ? 1.f : 0.f;
        for (int i = 0; i < B; ++i) dtot[(size_t)i * B + j] = (expf(total[(size_t)i * B + j] - lse) - (i == j ? 1.f : 0.f)) / (float)B;
    }
    red[tid] = loss;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if (tid < s_) red[tid] += red[tid + s_];
        __syncthreads();
    }
    if (tid == 0) out[0] = red[0] / (float)B;
    __syncthreads();
    red[tid] = hit;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if (tid < s_) red[tid] += red[tid + s_];
        __syncthreads();
    }
    if (tid == 0) out[1] = red[0] / (float)B;
    __syncthreads();
    for (long e = tid; e < (long)2 * B * C; e += 256) {   // d s_i = sum_j dtot[i][j] l_j / 0.05 ; d l_j = sum_i dtot[i][j] s_i / 0.05
        const int side = (int)(e / ((long)B * C)), r = (int)((e / C) % B), c = (int)(e % C);
        float a = 0.f;
        if (side == 0) for (int j = 0; j < B; ++j) a += dtot[(size_t)r * B + j] * sn[((size_t)B + j) * C + c];
        else for (int i = 0; i < B; ++i) a += dtot[(size_t)i * B + r] * sn[(size_t)i * C + c];
        dsn[e] = a / 0.05f;
    }
    __syncthreads();
    for (int r = wave; r < 2 * B; r += 4) {          // through F.normalize: d m = (d s - s (s . d s)) / |m|
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += sn[(size_t)r * C + c] * dsn[(size_t)r * C + c];
        a = wave_sum(a);
        const float n = nrm[r];
        for (int c = lane; c < C; c += 64) {
            const float g = n > 1e-12f ? (dsn[(size_t)r * C + c] - sn[(size_t)r * C + c] * a) / n : dsn[(size_t)r * C + c] / 1e-12f;
            dmean[(size_t)r * C + c] = g;
        }
    }
}
__global__ void nce_bcast_kernel(const float* __restrict__ dmean, const int* __restrict__ lens, int B, int T, int C, float* __restrict__ dxs,
                                 float* __restrict__ dxl) {
    const long per = (long)B * T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * per; i += (long)gridDim.x * blockDim.x) {
        const int side = i >= per ? 1 : 0;
        const long e = i - side * per;
        const int c = (int)(e % C);
        const long r = e / C;
        const int t = (int)(r % T), b = (int)(r / T);
        const int len = lens[b];
        (side ? dxl : dxs)[e] = t < len ? dmean[((size_t)side * B + b) * C + c] / (float)len : 0.f;
    }
}

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

// ------------------------------------------------------------------------------------------------ launchers
int tr_transpose_pad(int out_dtype, const float* in, int ld_in, void* out, int ld_out, int R, int C, hipStream_t s) {
    DIMX_REQUIRE(in && out && R > 0 && C > 0 && ld_out >= R, DIMX_ERR_ARG, "transpose_pad: bad arguments");
    dim3 grid(ceil_div(C, 32), ceil_div(ld_out, 32));
    if (out_dtype == DIMX_BF16) hipLaunchKernelGGL(transpose_pad_kernel<bf16>, grid, dim3(256), 0, s, in, ld_in, (bf16*)out, ld_out, R, C);
    else hipLaunchKernelGGL(transpose_pad_kernel<float>, grid, dim3(256), 0, s, in, ld_in, (float*)out, ld_out, R, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int tr_prep_weights(int out_dtype, const PrepTable& t, hipStream_t s) {
    if (t.n == 0) return DIMX_OK;
    DIMX_REQUIRE(t.n <= kPrepMax && t.total_tiles > 0, DIMX_ERR_ARG, "prep_weights: bad table");
    if (out_dtype == DIMX_BF16) hipLaunchKernelGGL(multi_prep_kernel<bf16>, dim3(t.total_tiles), dim3(256), 0, s, t);
    else hipLaunchKernelGGL(multi_prep_kernel<float>, dim3(t.total_tiles), dim3(256), 0, s, t);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int tr_prep_pair(int out_dtype, const float* src, int lds, int rows, int cols, void* o, int Kp, void* t, int Mp, hipStream_t s, int gelu) {
    DIMX_REQUIRE(src && o && t && rows > 0 && cols > 0 && Kp >= cols && Mp >= rows, DIMX_ERR_ARG, "prep_pair: bad arguments");
    PrepDesc d;
    d.gelu = gelu;
    d.src = src;
    d.w = o;
    d.wt = t;
    d.N = rows;
    d.K = cols;
    d.Kp = Kp;
    d.Np = Mp;
    d.lds = lds;
    d.tile0 = 0;
    d.tiles_k = ceil_div(Kp, 32);
    const int tiles = ceil_div(Mp, 32) * d.tiles_k;
    if (out_dtype == DIMX_BF16) hipLaunchKernelGGL(prep_pair_kernel<bf16>, dim3(tiles), dim3(256), 0, s, d);
    else hipLaunchKernelGGL(prep_pair_kernel<float>, dim3(tiles), dim3(256), 0, s, d);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

// see prep_fused_kernel.  colpart: [*n_part][cols] partial column sums (n_part = ceil(rows / 32) rows are written), or null
int tr_prep_fused(int out_dtype, int mode, const float* src, int lds, const float* src2, int lds2, const float* gamma, int rows, int cols,
                  void* o, int Kp, void* t, int Mp, float* colpart, int* n_part, hipStream_t s, const float* beta) {
    DIMX_REQUIRE(src && o && t && rows > 0 && cols > 0 && Kp >= cols && Mp >= rows && mode >= 0 && mode <= 5, DIMX_ERR_ARG, "prep_fused: bad arguments");
    DIMX_REQUIRE(cols % 4 == 0 && lds % 4 == 0 && Kp % 4 == 0 && Mp % 32 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)o % 16) == 0 &&
                     ((uintptr_t)t % 16) == 0,
                 DIMX_ERR_ARG, "prep_fused: 16-byte groups (cols=%d lds=%d Kp=%d Mp=%d)", cols, lds, Kp, Mp);
    DIMX_REQUIRE(mode != 2 || (gamma && ((uintptr_t)gamma % 16) == 0), DIMX_ERR_ARG, "prep_fused: LayerNorm mode needs gamma");
    DIMX_REQUIRE((mode != 3 && mode != 5) || (src2 && lds2 % 4 == 0 && ((uintptr_t)src2 % 16) == 0), DIMX_ERR_ARG,
                 "prep_fused: GELU' mode needs the pre-activation");
    DIMX_REQUIRE(!beta || (mode == 2 && ((uintptr_t)beta % 16) == 0), DIMX_ERR_ARG, "prep_fused: beta belongs to the LayerNorm mode");
    PrepFused d;
    d.src = src; d.lds = lds; d.src2 = src2; d.lds2 = lds2; d.gamma = gamma; d.beta = beta;
    d.o = o; d.Kp = Kp; d.t = t; d.Mp = Mp; d.rows = rows; d.cols = cols; d.mode = mode; d.colpart = colpart;
    // LayerNorm: the block that owns 32 rows computes their statistics once, so it walks the whole row; otherwise 256-column chunks
    // columns per block: enough blocks to fill the chip (>= ~512) when there are few rows, at most 256 columns (LayerNorm mode: the
    // row statistics are recomputed by every column chunk of a row block, so its chunks stay >= 128 wide)
    {
        static const int force = getenv("DIMX_PREP_CHUNK") ? atoi(getenv("DIMX_PREP_CHUNK")) : 0;
        int c = (int)((long)Kp * (Mp / 32) / 512) / 64 * 64;
        c = c < 64 ? 64 : (c > 256 ? 256 : c);
        if (mode == 2 && c < 128) c = 128;
        if (mode == 2 && force > 0) c = force;
        d.chunk = c;
    }
    if (n_part) *n_part = ceil_div(rows, 32);
    const dim3 grid(Mp / 32, ceil_div(Kp, d.chunk));
    if (out_dtype == DIMX_BF16) hipLaunchKernelGGL(prep_fused_kernel<bf16>, grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL(prep_fused_kernel<float>, grid, dim3(256), 0, s, d);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

static AttnShape make_shape(const TrAttn& t) {
    AttnShape a;
    a.B = t.B; a.H = t.H; a.Lq = t.Lq; a.Lk = t.Lk;
    a.ldq = t.ldq; a.ldk = t.ldk; a.ldv = t.ldv; a.ldo = t.ldo;
    a.scale = t.scale;
    a.causal = t.causal;
    a.kmask = t.kmask;
    a.kmask2 = t.kmask2;
    return a;
}
int tr_attn_fwd(const TrAttn& t, const float* q, const float* k, const float* v, float* o, float* lse, hipStream_t s) {
    if (t.mfma) return tr_attn_fwd_mfma(t, q, k, v, o, lse, s);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(t.Lq, t.H, t.B), dim3(64), 0, s, make_shape(t), q, k, v, o, lse);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_attn_bwd(const TrAttn& t, const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, hipStream_t s) {
    if (t.mfma) return tr_attn_bwd_mfma(t, q, k, v, o, d_o, lse, delta, dq, lddq, dk, lddk, dv, lddv, s);
    const AttnShape a = make_shape(t);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(t.Lq, t.H, t.B), dim3(64), 0, s, a, q, k, v, o, d_o, lse, dq, lddq, delta);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(t.Lk, t.H, t.B), dim3(64), 0, s, a, q, k, v, d_o, lse, delta, dk, lddk, dv, lddv);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, int accumulate, int M, int C, hipStream_t s) {
    if (accumulate) hipLaunchKernelGGL(layernorm_bwd_kernel<true>, dim3(ceil_div(M, 4)), dim3(256), 0, s, x, gamma, dy, dx, M, C);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<false>, dim3(ceil_div(M, 4)), dim3(256), 0, s, x, gamma, dy, dx, M, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
// dx (+)= LN'(x) dy; partial rows of colsum(dy o xhat) -> part_g [blocks][C], of colsum(dy) -> part_b (optional).  Returns the
// number of partial rows in *nrows (the caller queues them for tr_multi_finish).
int tr_layernorm_bwd_fused(const float* x, const float* gamma, const float* dy, float* dx, int accumulate, int M, int C, float* part_g,
                           float* part_b, int* nrows, hipStream_t s) {
    DIMX_REQUIRE((C == 384 || C == 512 || C == 768 || C == 1152) && part_g && nrows, DIMX_ERR_ARG, "layernorm_bwd_fused: C=%d", C);
    int rows = ceil_div(M, kLnBlocks);
    rows = (rows + 3) / 4 * 4;
    const int blocks = ceil_div(M, rows);
    *nrows = blocks;
    if (!x) return DIMX_OK;   // sizing pass
#define LNB(AC, NC) hipLaunchKernelGGL((ln_bwd_fused_kernel<AC, NC>), dim3(blocks), dim3(256), 0, s, x, gamma, dy, dx, M, C, rows, part_g, part_b)
#define LNB_C(AC)                    \
    do {                             \
        if (C == 384) LNB(AC, 6);    \
        else if (C == 512) LNB(AC, 8); \
        else if (C == 768) LNB(AC, 12); \
        else LNB(AC, 18);            \
    } while (0)
    if (accumulate) LNB_C(true); else LNB_C(false);
#undef LNB_C
#undef LNB
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
// partial rows of colsum(dy) -> part [slabs][C] (bias gradients); *nrows = slabs
int tr_colsum_partial(const float* dy, int M, int C, float* part, int* nrows, hipStream_t s) {
    const int slabs = M < kTrSlabs * 8 ? 1 : kTrSlabs;
    const int rows = ceil_div(M, slabs);
    *nrows = slabs;
    if (!dy) return DIMX_OK;  // sizing pass
    hipLaunchKernelGGL(ln_colsums_kernel, dim3(ceil_div(C, 64), slabs), dim3(256), 0, s, (const float*)nullptr, dy, (float*)nullptr, part, M, C, rows);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_multi_finish(const FinTable& t, hipStream_t s) {
    if (t.n == 0) return DIMX_OK;
    hipLaunchKernelGGL(multi_finish_kernel, dim3(t.total_blocks), dim3(256), 0, s, t);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_xhat(const float* x, float* xh, int M, int C, hipStream_t s) {
    hipLaunchKernelGGL(xhat_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, s, x, xh, M, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
// dg[c] (+)= sum_m dy[m,c] * xh[m,c] (xh may be null), db[c] (+)= sum_m dy[m,c]; part: >= 2 * kTrSlabs * C floats of scratch
int tr_colsums(const float* xh, const float* dy, float* dg, float* db, int M, int C, float* part, int accumulate, hipStream_t s) {
    const int slabs = M < kTrSlabs * 8 ? 1 : kTrSlabs;
    const int rows = ceil_div(M, slabs);
    float* pg = dg ? part : nullptr;
    float* pb = db ? part + (size_t)kTrSlabs * C : nullptr;
    hipLaunchKernelGGL(ln_colsums_kernel, dim3(ceil_div(C, 64), slabs), dim3(256), 0, s, xh, dy, pg, pb, M, C, rows);
    if (dg) hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, pg, dg, C, slabs, 1.0f, accumulate);
    if (db) hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, pb, db, C, slabs, 1.0f, accumulate);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_gelu_fwd(const float* pre, float* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_erf_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, s, pre, out, n);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_gelu_bwd(const float* pre, const float* dh, float* dpre, long n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_erf_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, s, pre, dh, dpre, n);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_add(float* y, const float* a, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, s, y, a, n);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_add_rows(const float* a, int lda, const float* row, const float* table, float scale, int T, float* y, int ldy, int M, int C,
                hipStream_t s) {
    hipLaunchKernelGGL(add_rows_kernel, dim3(ew_grid((long)M * C)), dim3(256), 0, s, a, lda, row, table, scale, T, y, ldy, M, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_zero_rows(float* y, const uint8_t* keep, int M, int C, hipStream_t s) {
    hipLaunchKernelGGL(zero_rows_kernel, dim3(ew_grid((long)M * C)), dim3(256), 0, s, y, keep, M, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_copy_cols(const float* src, int lds_, float* dst, int ldd, int M, int C, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(copy_cols_kernel, dim3(ew_grid((long)M * C)), dim3(256), 0, s, src, lds_, dst, ldd, M, C, accumulate);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_pos_grad(const float* dx, float* dtable, int B, int T, int C, float scale, hipStream_t s, int accumulate) {
    hipLaunchKernelGGL(pos_grad_kernel, dim3(ew_grid((long)T * C)), dim3(256), 0, s, dx, dtable, B, T, C, scale, accumulate);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
// loss_out[0] = mean CE over valid targets, loss_out[1] = 1 / count; dlogits = d loss / d logits
int tr_cross_entropy(const float* logits, const int32_t* target, float* row_loss, float* dlogits, int R, float* loss_out, hipStream_t s) {
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, s, target, R, loss_out);
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, logits, target, row_loss, dlogits, R, loss_out + 1);
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, row_loss, R, loss_out + 1, loss_out);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_onehot_t(int out_dtype, const int32_t* tokens, void* out, int ld_out, int M, int rows, hipStream_t s) {
    const dim3 grid(ceil_div(ld_out, 256), rows);
    if (out_dtype == DIMX_BF16) hipLaunchKernelGGL(onehot_t_kernel<bf16>, grid, dim3(256), 0, s, tokens, (bf16*)out, ld_out, M);
    else hipLaunchKernelGGL(onehot_t_kernel<float>, grid, dim3(256), 0, s, tokens, (float*)out, ld_out, M);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
// norm_out[0] = ||g||_2, norm_out[1] = clip coefficient; part: >= 1024 floats of 8-byte aligned scratch (512 f64 partial sums)
int tr_grad_norm(const float* g, long n, float max_norm, float* part, float* norm_out, hipStream_t s) {
    const int nb = 512;   // 2 blocks per CU
    DIMX_REQUIRE(((uintptr_t)part % 8) == 0 && ((uintptr_t)g % 16) == 0, DIMX_ERR_ARG, "grad_norm: scratch must be 8-byte aligned");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, (double*)part);
    hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nb, max_norm, norm_out);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
             const float* clip, hipStream_t s) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, sqrtf(bc2), clip);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

// ---- VQ-VAE decoder / legacy generator glue
#define EW(kernel, n_, ...)                                                                      \
    do {                                                                                         \
        hipLaunchKernelGGL(kernel, dim3(ew_grid((long)(n_))), dim3(256), 0, s, __VA_ARGS__);    \
        DIMX_HIP(hipGetLastError());                                                             \
        return DIMX_OK;                                                                          \
    } while (0)
int tr_im2col5(const float* x, float* x5, int B, int n, int C, hipStream_t s) { EW(im2col5_kernel, (long)B * n * C, x, x5, B, n, C); }
int tr_col2im5(const float* dx5, float* dx, int B, int n, int C, hipStream_t s) { EW(col2im5_kernel, (long)B * n * C, dx5, dx, B, n, C); }
int tr_lrelu_inorm_fwd(const float* x, float* y, int B, int n, int C, hipStream_t s) {
    hipLaunchKernelGGL(lrelu_inorm_fwd_kernel, dim3(B, ceil_div(C, 64)), dim3(64), 0, s, x, y, n, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_lrelu_inorm_bwd(const float* x, const float* dy, float* dx, int B, int n, int C, hipStream_t s) {
    hipLaunchKernelGGL(lrelu_inorm_bwd_kernel, dim3(B, ceil_div(C, 64)), dim3(64), 0, s, x, dy, dx, n, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_add_clip_rows(const float* a, const float* rows, float* y, int M, int n, int C, hipStream_t s) {
    EW(add_clip_rows_kernel, (long)M * C, a, rows, y, M, n, C);
}
int tr_argmax512(const float* logits, int32_t* idx, int B, int n_in, int n_out, int t_off, hipStream_t s) {
    hipLaunchKernelGGL(argmax512_kernel, dim3(ceil_div(B * n_out, 4)), dim3(256), 0, s, logits, idx, B, n_in, n_out, t_off);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
// loss_out[0] = continuous loss, loss_out[1] = 1 / selected rows; dpred = d loss / d pred.  rown: 2 * B * n floats of scratch
int tr_cont_loss(const float* pred, const float* v_tgt, const uint8_t* mask, int B, int T, int n, float* rown, float* dpred, float* loss_out,
                 hipStream_t s) {
    hipLaunchKernelGGL(cont_rows_kernel, dim3(ceil_div(B * n, 4)), dim3(256), 0, s, pred, v_tgt, mask, B, T, n, rown, dpred);
    hipLaunchKernelGGL(cont_finish_kernel, dim3(1), dim3(256), 0, s, rown, mask, B, T, n, loss_out);
    hipLaunchKernelGGL(scale_by_kernel, dim3(ew_grid((long)B * n * 56)), dim3(256), 0, s, dpred, loss_out + 1, (long)B * n * 56);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
int tr_pad_head_rows(const float* src, float* dst, int groups, int cols, int unpad, hipStream_t s) {
    EW(pad_head_rows_kernel, (long)groups * 64 * cols, src, dst, groups, cols, unpad);
}
int tr_pad_head_cols(const float* src, float* dst, int rows, int groups, int unpad, hipStream_t s) {
    EW(pad_head_cols_kernel, (long)rows * groups * 64, src, dst, rows, groups, unpad);
}
int tr_emb_relu(const float* table, const int32_t* ids, float* e, int B, int E, hipStream_t s) { EW(emb_relu_kernel, (long)B * E, table, ids, e, B, E); }
int tr_emb_relu_bwd(const float* table, const int32_t* ids, const float* de, float* dtable, int B, int E, hipStream_t s) {
    EW(emb_relu_bwd_kernel, (long)E, table, ids, de, dtable, B, E);
}
int tr_prepend_row(const float* first, const float* rest, float* joint, int B, int T, int C, int split, hipStream_t s) {
    EW(prepend_row_kernel, (long)B * (T + 1) * C, first, rest, joint, B, T, C, split);
}
int tr_prepend_tokens(const int32_t* z, const uint8_t* mask, int32_t* z_ext, uint8_t* m_ext, int B, int T, hipStream_t s) {
    EW(prepend_tokens_kernel, (long)B * (T + 1), z, mask, z_ext, m_ext, B, T);
}
int tr_gather128(const float* book, const int32_t* idx, float* out, int R, hipStream_t s) { EW(gather128_kernel, (long)R * 128, book, idx, out, R); }
int tr_copy_bt(const float* src, long src_bs, int src_ld, float* dst, long dst_bs, int dst_ld, int B, int T, int C, const float* addrow,
               int accumulate, hipStream_t s) {
    EW(copy_bt_kernel, (long)B * T * C, src, src_bs, src_ld, dst, dst_bs, dst_ld, B, T, C, addrow, accumulate);
}
int tr_copy_bt_u8(const uint8_t* src, long src_bs, uint8_t* dst, long dst_bs, int B, int T, int invert, hipStream_t s) {
    EW(copy_bt_u8_kernel, (long)B * T, src, src_bs, dst, dst_bs, B, T, invert);
}
int tr_mask_tokens(const int32_t* z, const uint8_t* sel, int32_t* out, long n, hipStream_t s) { EW(mask_tokens_kernel, n, z, sel, out, n); }
size_t tr_nce_scratch_floats(int B, int C) { return (size_t)8 * B * C + 4 * (size_t)B + 2 * (size_t)B * B + 64; }
// out[0] = nce, out[1] = c_acc; dxs / dxl [B, T, C] = d nce / d x_s, d x_l.  scr: tr_nce_scratch_floats(B, C) floats
int tr_nce(const float* xs, const float* xl, const uint8_t* mask, int B, int T, int C, float* scr, float* out, float* dxs, float* dxl,
           hipStream_t s) {
    float* mean = scr;                              // [2][B][C]
    float* dmean = mean + (size_t)2 * B * C;        // [2][B][C]
    int* lens = (int*)(dmean + (size_t)2 * B * C);  // [B]
    float* core = (float*)(lens + B + (B & 1));     // sn | nrm | total | dtot | dsn
    hipLaunchKernelGGL(nce_mean_kernel, dim3(B, 2), dim3(128), 0, s, xs, xl, mask, B, T, C, mean, lens);
    hipLaunchKernelGGL(nce_core_kernel, dim3(1), dim3(256), 0, s, mean, B, C, core, dmean, out);
    hipLaunchKernelGGL(nce_bcast_kernel, dim3(ew_grid((long)2 * B * T * C)), dim3(256), 0, s, dmean, lens, B, T, C, dxs, dxl);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}
#undef EW

}  // namespace dimx
