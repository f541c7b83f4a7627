// elementwise.hip -- the small HBM-bound kernels that glue the dense stages together: input cast/pad,
// embedding gathers, context assembly, token bookkeeping, cross-entropy/argmax over the 512-entry
// vocabulary and the top-k / softmax / noise-argmax sampler of the autoregressive loop.
#include "common.hpp"

namespace dimx {
namespace {

// y[m, 0:ldy) = x[m, 0:K) (+ coladd) zero-padded; x is f32 [M, ldx]
template <typename OutT>
__global__ void cast_pad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ coladd,
                                OutT* __restrict__ y, int ldy, int M, int K, const uint8_t* __restrict__ zero_rows) {
    const long total = (long)M * ldy;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ldy), k = (int)(i - (long)m * ldy);
        float v = 0.f;
        if (k < K && !(zero_rows && zero_rows[m])) {  // zero_rows: SLM's masked frames (seq2seq_pretrain.py:211-212)
            v = x[(size_t)m * ldx + k];
            if (coladd) v += coladd[k];
        }
        store_from_f32<OutT>(y + i, v);
    }
}

// y[m, :] = table[idx[m], :]   (codebook lookup = the one-hot matmul of
// code/seq2seq_pretrain.py:457-459; token embedding of the teacher-forced decoder)
template <typename OutT>
__global__ void gather_rows_kernel(const float* __restrict__ table, int ld_table, int rows,
                                   const int32_t* __restrict__ idx, OutT* __restrict__ y, int ldy, int M, int C) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        int r = idx[m];
        r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
        store_from_f32<OutT>(y + (size_t)m * ldy + c, table[(size_t)r * ld_table + c]);
    }
}

// ctx[m] = cat(x_s[m] + patch_embed_dec_s, audio[m])   (code/seq2seq_pretrain.py:445-446)
template <typename OutT>
__global__ void context_concat_kernel(const float* __restrict__ x_s, const float* __restrict__ patch,
                                      const float* __restrict__ audio, OutT* __restrict__ ctx, int M, int dim,
                                      int dim_a) {
    const int W = dim + dim_a;
    const long total = (long)M * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / W), c = (int)(i - (long)m * W);
        const float v = c < dim ? x_s[(size_t)m * dim + c] + patch[c] : audio[(size_t)m * dim_a + (c - dim)];
        store_from_f32<OutT>(ctx + i, v);
    }
}

// fqn codes per frame (1 for the listener / SLMFT VQ-VAEs, 8 for the legacy speaker VQ-VAE)
__global__ void finalize_idx_kernel(int32_t* idx, const int32_t* lens, int B, int T, int fqn, int32_t pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * fqn) return;
    const int b = i / (T * fqn), t = (i - b * T * fqn) / fqn;
    if (lens && t >= lens[b]) idx[i] = pad;
}

// AutoregressiveWrapper.forward: inp = z[:, :-1] with ignore_index -> pad_value 0, target = z[:, 1:]
__global__ void shift_tokens_kernel(const int32_t* z, int32_t* inp, int32_t* tgt, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = T - 1;
    if (i >= B * n) return;
    const int b = i / n, t = i - b * n;
    const int32_t v = z[b * T + t];
    inp[i] = v == -100 ? 0 : v;
    tgt[i] = z[b * T + t + 1];
}

// one wave per row of V = 512 logits: cross entropy against target (0 where target < 0) and argmax
__global__ __launch_bounds__(256) void ce_argmax_kernel(const float* __restrict__ logits,
                                                        const int32_t* __restrict__ target,
                                                        float* __restrict__ row_loss,
                                                        int32_t* __restrict__ argmax_tok, int R) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* lr = logits + (size_t)row * 512;
    float v[8];
    float mx = -3.0e38f;
    int mi = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v[c] = lr[lane + 64 * c];
        if (v[c] > mx) {
            mx = v[c];
            mi = lane + 64 * c;
        }
    }
    wave_argmax(mx, mi);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) se += expf(v[c] - mx);
    se = wave_sum(se);
    if (lane == 0) {
        if (argmax_tok) argmax_tok[row] = mi;
        if (row_loss) {
            const int t = target ? target[row] : -100;
            row_loss[row] = (t >= 0 && t < 512) ? (mx + logf(se)) - lr[t] : 0.f;
        }
    }
}

__device__ __forceinline__ uint32_t f32_order_key(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint64_t splitmix(uint64_t key, uint64_t ctr) {
    uint64_t z = key + (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Sampler of AutoregressiveWrapper.generate: top-k filter (k = 52), softmax(T), multinomial.
// torch.multinomial(p, 1) == argmax(p / q), q ~ Exp(1) (tests/golden/sampler_multinomial.npz), so the
// noise is an input: injected (parity) or drawn from a counter-based splitmix64 stream (production).
// One wave per row; the k-th largest logit is found by a 32-step bitwise search on the order-preserving
// integer image of the floats, counting with ballots.
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, int ld, int R, int top_k,
                                                     float temperature, const float* __restrict__ noise,
                                                     uint64_t seed, const int32_t* __restrict__ step_dev,
                                                     uint64_t step_host, int32_t* __restrict__ tokens, int tok_ld,
                                                     int tok_col_from_step, int nslab, long slab_stride,
                                                     float* __restrict__ logits_out, int logits_out_ld, int row0,
                                                     int rows_total,
                                                     const float* __restrict__ emb_table, int emb_C,
                                                     float* __restrict__ x_next, int32_t* __restrict__ step_rw,
                                                     unsigned* __restrict__ done_ctr,
                                                     const float* __restrict__ pos_table, float pos_scale,
                                                     int pos_rows, const int32_t* __restrict__ dev_params,
                                                     void* __restrict__ y_next, const float* __restrict__ y_gamma,
                                                     int y_bf16) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint64_t step = step_dev ? (uint64_t)*step_dev : step_host;
    if (dev_params) {  // generate(): temperature / seed live in device memory so that the captured step graph
                       // does not depend on them (a new seed per batch must not force a re-capture)
        temperature = __builtin_bit_cast(float, dev_params[0]);
        seed = *(const uint64_t*)(dev_params + 2);
    }
    // counter of the on-device generator: (step, GLOBAL sequence row, code) -- a rank that generates rows
    // [off, off + R) of a sharded batch of `tot` sequences draws what the single-process batch would draw
    const int rng_off = dev_params ? dev_params[4] : 0;
    const int rng_tot = dev_params && dev_params[5] > 0 ? dev_params[5] : rows_total;
    if (row < R) {
    const float* lr = logits + (size_t)row * ld;
    // the fused pre-norm's weight does not depend on the token: its loads go out with the logits' (behind the token's embedding
    // row they were one more exposed L2 round trip at the tail of every decode step)
    float2 gam[16];
    const bool fast_embed = x_next && emb_C / 2 <= 16 * 64;
    if (y_next && fast_embed) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = lane + 64 * k;
            gam[k] = ((const float2*)y_gamma)[i < emb_C / 2 ? i : emb_C / 2 - 1];
        }
    }
    float v[8];
    float mx = -3.0e38f;
    int mi = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v[c] = lr[lane + 64 * c];
        for (int sidx = 1; sidx < nslab; ++sidx) v[c] += lr[(size_t)sidx * slab_stride + lane + 64 * c];  // split-K slabs
        if (v[c] > mx) {
            mx = v[c];
            mi = lane + 64 * c;
        }
    }
    if (logits_out) {  // per-step dump of the raw logits: [rows, steps, 512]
        float* lo = logits_out + ((size_t)row * logits_out_ld + step) * 512;
#pragma unroll
        for (int c = 0; c < 8; ++c) lo[lane + 64 * c] = v[c];
    }
    wave_argmax(mx, mi);
    const bool greedy = temperature <= 0.f || (noise == nullptr && seed == 0);
    int tok = mi;
    if (!greedy) {
        uint32_t key[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) key[c] = f32_order_key(v[c]);
        uint32_t thr = 0;
        if (top_k > 0 && top_k < 512) {
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = thr | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c) cnt += __popcll(__ballot(key[c] >= cand));
                if (cnt >= top_k) {
                    thr = cand;
                    // exactly k keys at or above the candidate: that IS the top-k set (every key >= its minimum is
                    // already counted), the remaining bits could only tighten the threshold onto that minimum
                    if (cnt == top_k) break;
                }
            }
        }
        // Only the <= top_k (+ ties) survivors need a probability and a noise draw: compact them to one per lane
        // (2 rounds cover up to 128 survivors; more than that -- massive ties -- falls back to the per-element loop).
        const float inv_t = 1.0f / temperature;
        int nsurv_lane = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) nsurv_lane += key[c] >= thr ? 1 : 0;
        int incl = nsurv_lane;  // inclusive prefix over lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int total = __shfl(incl, 63);
        float best = -1.f;
        int bi = 0x7fffffff;
        auto score = [&](float pv, float zs, int i) {
            float q;
            if (noise) {
                q = noise[((size_t)step * rows_total + row0 + row) * 512 + i];
            } else {
                const uint64_t r = splitmix(seed, ((uint64_t)step * rng_tot + rng_off + row0 + row) * 512 + i);
                const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
                q = fmaxf(-log1pf(-u), 9.313225746154785e-10f);
            }
            const float sc = (pv / zs) / q;
            if (sc > best || (sc == best && i < bi)) {
                best = sc;
                bi = i;
            }
        };
        if (total <= 128) {
            __shared__ float cand_v[4][128];
            __shared__ int cand_i[4][128];
            float* cv = cand_v[threadIdx.x >> 6];
            int* ci = cand_i[threadIdx.x >> 6];
            int pos = incl - nsurv_lane;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (key[c] >= thr) {
                    cv[pos] = v[c];
                    ci[pos] = lane + 64 * c;
                    ++pos;
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float pv[2] = {0.f, 0.f};
            int pi[2] = {0, 0};
            float zsum = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = lane + 64 * r;
                if (k < total) {
                    pv[r] = expf((cv[k] - mx) * inv_t);
                    pi[r] = ci[k];
                    zsum += pv[r];
                }
            }
            zsum = wave_sum(zsum);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (lane + 64 * r < total) score(pv[r], zsum, pi[r]);
        } else {
            float p[8];
            float zsum = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                p[c] = key[c] >= thr ? expf((v[c] - mx) * inv_t) : 0.f;
                zsum += p[c];
            }
            zsum = wave_sum(zsum);
#pragma unroll
            for (int c = 0; c < 8; ++c) score(p[c], zsum, lane + 64 * c);
        }
        wave_argmax(best, bi);
        tok = bi;
    }
    if (lane == 0) tokens[(size_t)row * tok_ld + (tok_col_from_step ? (int)step : 0)] = tok;
    if (x_next) {  // next step's decoder input: token embedding row (fused embed_step)
        const float2* src = (const float2*)(emb_table + (size_t)tok * emb_C);
        float2* dst = (float2*)(x_next + (size_t)row * emb_C);
        const bool with_pos = pos_table && (int)step + 1 < pos_rows;   // legacy decoder: + pos_emb[next position] * dim^-0.5
        const float2* pr = with_pos ? (const float2*)(pos_table + (size_t)(step + 1) * emb_C) : nullptr;
        // all of the row's loads in flight together, then the stores (a plain copy loop keeps load -> store order: up to
        // 9 dependent L2 round trips at the tail of every decode step)
        const int nv = emb_C / 2;
        if (nv <= 16 * 64) {
            float2 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = lane + 64 * k;
                v[k] = src[i < nv ? i : nv - 1];
            }
            if (with_pos) {
                float2 q[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int i = lane + 64 * k;
                    q[k] = pr[i < nv ? i : nv - 1];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    v[k] = make_float2(__fadd_rn(v[k].x, __fmul_rn(q[k].x, pos_scale)), __fadd_rn(v[k].y, __fmul_rn(q[k].y, pos_scale)));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = lane + 64 * k;
                if (i < nv) dst[i] = v[k];
            }
            if (y_next) {
                // round 4: the first decoder layer's pre-norm of this row rides along (y = LayerNorm(x) * gamma, no bias) -- one
                // launch less per decode step.  Same lane <-> element map, summation order and wave reductions as
                // add_slabs_layernorm_kernel (norm.hip), so the f32 parity mode's bits do not change.
                float sm = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (lane + 64 * k < nv) sm += v[k].x + v[k].y;
                const float inv_c = 1.0f / (float)emb_C;
                const float mean = (y_bf16 ? wave_sum_sel<true>(sm) : wave_sum_sel<false>(sm)) * inv_c;
                float qs = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (lane + 64 * k < nv) {
                        const float a0 = v[k].x - mean, b0 = v[k].y - mean;
                        qs += a0 * a0 + b0 * b0;
                    }
                const float rstd = rsqrtf((y_bf16 ? wave_sum_sel<true>(qs) : wave_sum_sel<false>(qs)) * inv_c + 1e-5f);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int i = lane + 64 * k;
                    if (i < nv) {
                        const float2 g = gam[k];
                        const float o0 = (v[k].x - mean) * rstd * g.x, o1 = (v[k].y - mean) * rstd * g.y;
                        if (y_bf16)
                            *(uint32_t*)((bf16*)y_next + (size_t)row * emb_C + 2 * i) = pack_bf16x2(o0, o1);
                        else
                            *(float2*)((float*)y_next + (size_t)row * emb_C + 2 * i) = make_float2(o0, o1);
                    }
                }
            }
        } else {
            for (int i = lane; i < nv; i += 64) {
                float2 e = src[i];
                if (with_pos) {
                    const float2 q = pr[i];
                    e = make_float2(__fadd_rn(e.x, __fmul_rn(q.x, pos_scale)), __fadd_rn(e.y, __fmul_rn(q.y, pos_scale)));
                }
                dst[i] = e;
            }
        }
    }
    }  // row < R
    // the block that finishes last advances the device step counter (every block has read it by then)
    if (step_rw) {
        __syncthreads();
        if (threadIdx.x == 0) {  // no fence needed: only the READ of the step counter must precede the arrival
            const unsigned prev = atomicAdd(done_ctr, 1u);
            if (prev == gridDim.x - 1) {
                *done_ctr = 0u;
                *step_rw = (int32_t)step + 1;
            }
        }
    }
}

// x[b, :] = token_emb[tok_b], tok_b = start[b] at step 0 else the token sampled at the previous step
__global__ __launch_bounds__(256) void embed_step_kernel(const float* __restrict__ table, int C,
                                                         const int32_t* __restrict__ start,
                                                         const int32_t* __restrict__ tokens, int tok_ld,
                                                         const int32_t* __restrict__ step_dev, float* __restrict__ x,
                                                         int rows, int start_div,
                                                         const float* __restrict__ pos_table, float pos_scale) {
    const int b = blockIdx.x;
    const int step = *step_dev;
    int tok = step == 0 ? start[b / start_div] : tokens[(size_t)b * tok_ld + step - 1];
    tok = tok < 0 ? 0 : (tok >= rows ? rows - 1 : tok);
    const float4* src = (const float4*)(table + (size_t)tok * C);
    float4* dst = (float4*)(x + (size_t)b * C);
    if (pos_table) {
        const float4* pr = (const float4*)(pos_table + (size_t)step * C);
        for (int i = threadIdx.x; i < C / 4; i += blockDim.x) {
            const float4 e = src[i], q = pr[i];
            dst[i] = make_float4(__fadd_rn(e.x, __fmul_rn(q.x, pos_scale)), __fadd_rn(e.y, __fmul_rn(q.y, pos_scale)),
                                 __fadd_rn(e.z, __fmul_rn(q.z, pos_scale)), __fadd_rn(e.w, __fmul_rn(q.w, pos_scale)));
        }
    } else {
        for (int i = threadIdx.x; i < C / 4; i += blockDim.x) dst[i] = src[i];
    }
}

// x[b*n + t, :] += pos[t, :] * scale   (TransformerWrapper abs. positional embedding of the legacy decoder,
// code/seq2seq.py:39 -> x-transformers AbsolutePositionalEmbedding: emb(arange(n)) * dim^-0.5)
__global__ void add_pos_rows_kernel(float* __restrict__ x, const float* __restrict__ pos, int M, int n, int C,
                                    float scale) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        x[i] = __fadd_rn(x[i], __fmul_rn(pos[(size_t)(m % n) * C + c], scale));
    }
}

// lens[b] = number of set entries of mask[b, :]  (the reference indexes v_speaker[i][mask[i]], code/seq2seq.py:228)
__global__ __launch_bounds__(64) void mask_lens_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ lens,
                                                       int T) {
    const int b = blockIdx.x;
    int n = 0;
    for (int t = threadIdx.x; t < T; t += 64) n += mask[(size_t)b * T + t] ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) n += __shfl_xor(n, o);
    if (threadIdx.x == 0) lens[b] = n;
}

// ListenerGenerator's x_speaker (code/seq2seq.py:224-241): the per-clip code vectors live channel-major as
// [zdim, len*fqn], are zero-padded along the LAST axis to T*fqn, and that memory is then re-read through
// .view(B,-1,fqn,zdim).view(B,-1,fqn*zdim) -- a reinterpretation, not a transpose.  Output element f of clip b is
// therefore channel c = f / (T*fqn), position p = f % (T*fqn) of the padded tensor.
template <typename OutT>
__global__ void legacy_scramble_kernel(const float* __restrict__ E, const int32_t* __restrict__ idx,
                                       const int32_t* __restrict__ lens, OutT* __restrict__ out, int B, int T, int fqn,
                                       int zdim, int n_embed) {
    const long per = (long)T * fqn * zdim, total = (long)B * per;
    const int P = T * fqn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long f = i - (long)b * per;
        const int c = (int)(f / P), p = (int)(f - (long)c * P);
        float v = 0.f;
        if (p < lens[b] * fqn) {
            int r = idx[(size_t)b * P + p];
            r = r < 0 ? 0 : (r >= n_embed ? n_embed - 1 : r);
            v = E[(size_t)r * zdim + c];
        }
        store_from_f32<OutT>(out + i, v);
    }
}

__global__ void step_inc_kernel(int32_t* step) { *step += 1; }

// per clip group: [0] step counter = 0, [8] done counter = 0, [2] temperature bits, [4..5] seed, [6] global row
// offset and [7] global row count of the sampler's counter-based generator (generate())
__global__ void gen_params_kernel(int32_t* base, int groups, float temperature, uint64_t seed, int row_off,
                                  int rows_total) {
    const int g = threadIdx.x;
    if (g >= groups) return;
    int32_t* p = base + 16 * g;
    p[0] = 0;
    p[8] = 0;
    p[2] = __builtin_bit_cast(int32_t, temperature);
    *(uint64_t*)(p + 4) = seed;
    p[6] = row_off;
    p[7] = rows_total;
}

// dst[b, step, :] = src[b, :]  (optional per-step logits dump of generate)
__global__ void copy_rows_step_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int V, int n,
                                      const int32_t* __restrict__ step_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * V) return;
    const int b = i / V, c = i - b * V;
    dst[((size_t)b * n + *step_dev) * V + c] = src[i];
}

__global__ void mask_and_kernel(const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t va = a ? a[i] : 1, vb = b ? b[i] : 1;
    out[i] = (va && vb) ? 1 : 0;
}

inline int ew_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

int launch_cast_pad(int out_dtype, const float* x, int ldx, const float* coladd, void* y, int ldy, int M, int K,
                    hipStream_t s, const uint8_t* zero_rows) {
    DIMX_REQUIRE(x && y && M > 0 && K > 0 && ldy >= K, DIMX_ERR_ARG, "cast_pad: bad arguments");
    const int g = ew_blocks((long)M * ldy);
    if (out_dtype == DIMX_BF16)
        hipLaunchKernelGGL(cast_pad_kernel<bf16>, dim3(g), dim3(256), 0, s, x, ldx, coladd, (bf16*)y, ldy, M, K, zero_rows);
    else
        hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(g), dim3(256), 0, s, x, ldx, coladd, (float*)y, ldy, M, K, zero_rows);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_gather_rows(int out_dtype, const float* table, int ld_table, int rows, const int32_t* idx, void* y,
                       int ldy, int M, int C, hipStream_t s) {
    DIMX_REQUIRE(table && idx && y && M > 0 && C > 0, DIMX_ERR_ARG, "gather_rows: bad arguments");
    const int g = ew_blocks((long)M * C);
    if (out_dtype == DIMX_BF16)
        hipLaunchKernelGGL(gather_rows_kernel<bf16>, dim3(g), dim3(256), 0, s, table, ld_table, rows, idx, (bf16*)y,
                           ldy, M, C);
    else
        hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(g), dim3(256), 0, s, table, ld_table, rows, idx,
                           (float*)y, ldy, M, C);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_context_concat(int out_dtype, const float* x_s, const float* patch, const float* audio, void* ctx, int M,
                          int dim, int dim_a, hipStream_t s) {
    DIMX_REQUIRE(x_s && patch && audio && ctx && M > 0, DIMX_ERR_ARG, "context_concat: bad arguments");
    const int g = ew_blocks((long)M * (dim + dim_a));
    if (out_dtype == DIMX_BF16)
        hipLaunchKernelGGL(context_concat_kernel<bf16>, dim3(g), dim3(256), 0, s, x_s, patch, audio, (bf16*)ctx, M,
                           dim, dim_a);
    else
        hipLaunchKernelGGL(context_concat_kernel<float>, dim3(g), dim3(256), 0, s, x_s, patch, audio, (float*)ctx, M,
                           dim, dim_a);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_finalize_idx(int32_t* idx, const int32_t* lens, int B, int T, int32_t pad_value, hipStream_t s, int fqn) {
    if (!lens) return DIMX_OK;
    hipLaunchKernelGGL(finalize_idx_kernel, dim3(ceil_div(B * T * fqn, 256)), dim3(256), 0, s, idx, lens, B, T, fqn,
                       pad_value);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_shift_tokens(const int32_t* z, int32_t* inp, int32_t* tgt, int B, int T, hipStream_t s) {
    hipLaunchKernelGGL(shift_tokens_kernel, dim3(ceil_div(B * (T - 1), 256)), dim3(256), 0, s, z, inp, tgt, B, T);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_ce_argmax(const float* logits, const int32_t* target, float* row_loss, int32_t* argmax_tok, int R, int V,
                     hipStream_t s) {
    DIMX_REQUIRE(V == 512, DIMX_ERR_ARG, "ce_argmax: vocabulary %d != 512", V);
    hipLaunchKernelGGL(ce_argmax_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, logits, target, row_loss, argmax_tok,
                       R);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_sample(const float* logits, int ld_logits, int R, int top_k, float temperature, const float* noise,
                  uint64_t seed, const int32_t* step_dev, uint64_t step_host, int32_t* tokens, int tok_ld,
                  int tok_col_from_step, int nslab, long slab_stride, float* logits_out, int logits_out_ld, int row0,
                  int rows_total, const float* emb_table, int emb_C, float* x_next, int32_t* step_rw, unsigned* done_ctr,
                  hipStream_t s, const float* pos_table, float pos_scale, int pos_rows, const int32_t* dev_params,
                  void* y_next, const float* y_gamma, int y_dtype) {
    DIMX_REQUIRE(logits && tokens && R > 0, DIMX_ERR_ARG, "sample: bad arguments");
    DIMX_REQUIRE(!y_next || (x_next && y_gamma && emb_C % 128 == 0 && emb_C <= 2048), DIMX_ERR_ARG,
                 "sample: the fused pre-norm needs the fused embedding and a width of k * 128 <= 2048");
    const int wpb = R <= 1024 ? 1 : 4;  // one row per block for decode-sized batches: all CUs busy
    hipLaunchKernelGGL(sample_kernel, dim3(ceil_div(R, wpb)), dim3(64 * wpb), 0, s, logits, ld_logits, R, top_k,
                       temperature, noise, seed, step_dev, step_host, tokens, tok_ld, tok_col_from_step, nslab < 1 ? 1 : nslab,
                       slab_stride, logits_out, logits_out_ld, row0, rows_total, emb_table, emb_C, x_next, step_rw, done_ctr,
                       pos_table, pos_scale, pos_rows, dev_params, y_next, y_gamma, y_dtype == DIMX_BF16 ? 1 : 0);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_gen_params(int32_t* base, int groups, float temperature, uint64_t seed, int row_off, int rows_total,
                      hipStream_t s) {
    hipLaunchKernelGGL(gen_params_kernel, dim3(1), dim3(64), 0, s, base, groups, temperature, seed, row_off, rows_total);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_embed_step(const float* table, int C, int rows, const int32_t* start, const int32_t* tokens, int tok_ld,
                      const int32_t* step_dev, float* x, int B, int start_div, hipStream_t s, const float* pos_table,
                      float pos_scale) {
    DIMX_REQUIRE(C % 4 == 0, DIMX_ERR_ARG, "embed_step: C %% 4");
    hipLaunchKernelGGL(embed_step_kernel, dim3(B), dim3(256), 0, s, table, C, start, tokens, tok_ld, step_dev, x,
                       rows, start_div < 1 ? 1 : start_div, pos_table, pos_scale);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_add_pos_rows(float* x, const float* pos, int M, int n, int C, float scale, hipStream_t s) {
    DIMX_REQUIRE(x && pos && M > 0 && n > 0, DIMX_ERR_ARG, "add_pos_rows: bad arguments");
    const long total = (long)M * C;
    const int blocks = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
    hipLaunchKernelGGL(add_pos_rows_kernel, dim3(blocks), dim3(256), 0, s, x, pos, M, n, C, scale);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_mask_lens(const uint8_t* mask, int32_t* lens, int B, int T, hipStream_t s) {
    DIMX_REQUIRE(mask && lens && B > 0 && T > 0, DIMX_ERR_ARG, "mask_lens: bad arguments");
    hipLaunchKernelGGL(mask_lens_kernel, dim3(B), dim3(64), 0, s, mask, lens, T);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_legacy_scramble(int out_dtype, const float* E, const int32_t* idx, const int32_t* lens, void* out, int B,
                           int T, int fqn, int zdim, int n_embed, hipStream_t s) {
    DIMX_REQUIRE(E && idx && lens && out && B > 0 && T > 0, DIMX_ERR_ARG, "legacy_scramble: bad arguments");
    const long total = (long)B * T * fqn * zdim;
    const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    if (out_dtype == DIMX_BF16)
        hipLaunchKernelGGL((legacy_scramble_kernel<bf16>), dim3(blocks), dim3(256), 0, s, E, idx, lens, (bf16*)out, B,
                           T, fqn, zdim, n_embed);
    else
        hipLaunchKernelGGL((legacy_scramble_kernel<float>), dim3(blocks), dim3(256), 0, s, E, idx, lens, (float*)out,
                           B, T, fqn, zdim, n_embed);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_step_inc(int32_t* step_dev, hipStream_t s) {
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, step_dev);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_copy_rows_step(const float* src, float* dst, int B, int V, int n, const int32_t* step_dev, hipStream_t s) {
    hipLaunchKernelGGL(copy_rows_step_kernel, dim3(ceil_div(B * V, 256)), dim3(256), 0, s, src, dst, B, V, n, step_dev);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_mask_and(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(mask_and_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, a, b, out, n);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
