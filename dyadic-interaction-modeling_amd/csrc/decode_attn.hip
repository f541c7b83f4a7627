// decode_attn.hip -- one-query attention of the autoregressive decoder step (KV-cached
// AutoregressiveWrapper.generate, call site reference code/seq2seq_pretrain.py:450): for every
// (clip, head) a single 64-d query attends over the cached keys.  Self-attention additionally appends
// this step's key/value to the cache; cross-attention reads the context K/V projected once per clip and
// honours the context (padding) mask.
//
// M = 1 per (clip, head): no matrix-core work, the kernel is a pure HBM stream of the K/V caches
// ([B,H,Tmax,64], 128 B (bf16) / 256 B (f32) per key).  One wave per (clip, head); 8 (bf16) or 16 (f32)
// lanes cooperate on one key so that every wave-wide load instruction covers 1 KiB of contiguous
// cache, scores go through LDS (one float per key), and the P.V pass re-streams V the same way.
#include "common.hpp"

namespace dimx {
namespace {

// K/V rows are read exactly once per launch and the per-layer cache (236 MB at C3) exceeds every cache level:
// stream them with the non-temporal policy so they do not evict the weights / slabs the neighbouring kernels re-use
typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
template <typename P> __device__ __forceinline__ uint4 ld_stream(const P* p) {
    const u32x4_nt v = __builtin_nontemporal_load((const u32x4_nt*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

constexpr int kMaxKeys = 2048;  // x-transformers max_seq_len of the decoder (code/seq2seq_pretrain.py:381)
constexpr float kNegD = -3.0e38f;

template <typename T, int EPC> __device__ __forceinline__ void load_chunk(const T* p, float (&v)[EPC]);
template <> __device__ __forceinline__ void load_chunk<float, 4>(const float* p, float (&v)[4]) {
    const float4 f = *(const float4*)p;
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
template <> __device__ __forceinline__ void load_chunk<bf16, 8>(const bf16* p, float (&v)[8]) {
    const uint4 u = *(const uint4*)p;
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
template <typename T, int EPC> __device__ __forceinline__ void store_chunk(T* p, const float (&v)[EPC]);
template <> __device__ __forceinline__ void store_chunk<float, 4>(float* p, const float (&v)[4]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store_chunk<bf16, 8>(bf16* p, const float (&v)[8]) {
    *(uint4*)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                            pack_bf16x2(v[6], v[7]));
}

template <typename T, int EPC> __device__ __forceinline__ void cvt_chunk(const uint4& u, float (&v)[EPC]);
template <> __device__ __forceinline__ void cvt_chunk<float, 4>(const uint4& u, float (&v)[4]) {
    v[0] = __builtin_bit_cast(float, u.x); v[1] = __builtin_bit_cast(float, u.y);
    v[2] = __builtin_bit_cast(float, u.z); v[3] = __builtin_bit_cast(float, u.w);
}
template <> __device__ __forceinline__ void cvt_chunk<bf16, 8>(const uint4& u, float (&v)[8]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}

template <int EPC> __device__ __forceinline__ void load_f32_chunk(const float* p, float (&v)[EPC]) {
#pragma unroll
    for (int i = 0; i < EPC / 4; ++i) {
        const float4 f = *(const float4*)(p + 4 * i);
        v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
    }
}
// sum of the split-K slabs in slab order (deterministic)
template <int EPC> __device__ __forceinline__ void load_f32_slabs(const float* p, int nslab, long stride, float (&v)[EPC]) {
    load_f32_chunk<EPC>(p, v);
    for (int s = 1; s < nslab; ++s) {
        float t[EPC];
        load_f32_chunk<EPC>(p + (size_t)s * stride, t);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] += t[e];
    }
}

// SELF = true: self-attention form (append this step's k/v, keys = step counter + 1);
// SELF = false: cross-attention form (fixed n_keys, optional key mask).  Two instantiations so that the two
// launch shapes show up as separate rows of a rocprofv3 kernel trace.
// QF32: q (and knew/vnew) are f32 split-K slabs [nslab][B, ld] written by the projection GEMM; they are
// summed here in slab order.
// NSPLIT (1, 2 or 4): waves per (clip, head).  Small batches / long contexts (BASELINE C5: 64 clips x 1500
// keys per GPU) do not have B*H >= CUs * waves to saturate HBM, so the keys of one (clip, head) are split
// over NSPLIT waves of the block and the partial (max, sum, acc) are combined through LDS in a fixed order.
// The kernel is a latency-bound HBM stream: every wave keeps 2 x U independent 16-byte loads per lane in
// flight (the next batch of U key groups is issued before the current one is consumed).
template <typename T, bool SELF, bool QF32, int NSPLIT>
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecodeAttnArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int LPK = 64 / EPC;   // lanes per key: 8 (bf16) / 16 (f32)
    constexpr int KPI = 64 / LPK;   // keys per wave-wide load: 8 / 4
    constexpr int U = 4;            // key groups per batch (8 measured no faster on MI355X)
    constexpr int KB = U * KPI;     // keys per batch
    constexpr int PPB = 4 / NSPLIT; // (clip, head) pairs per block
    __shared__ float sc[PPB][kMaxKeys];
    __shared__ float red_m[4], red_l[4];
    __shared__ float red_acc[4][64];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pib = wave / NSPLIT, part = wave % NSPLIT;
    const int pair = blockIdx.x * PPB + pib;
    const bool active = pair < a.B * a.H;
    const int b = active ? pair / a.H : 0, h = active ? pair % a.H : 0;
    const int sub = lane / LPK, ch = lane % LPK;
    // Everything that does not depend on the step counter is requested first: q and (self attention) this step's k / v
    // rows.  At step 0 the kernel is nothing but its dependent-load chain (step counter -> cache address -> q -> new k/v ->
    // store: 8.2 us per launch in round 2, four launches per decode step); issued here the three loads overlap the
    // counter read and each other.
    float qv[EPC], knv[EPC], vnv[EPC];
    if (QF32)
        load_f32_slabs<EPC>((const float*)a.q + (size_t)b * a.q_ld + h * 64 + ch * EPC, a.nslab, a.slab_stride, qv);
    else
        load_chunk<T, EPC>((const T*)a.q + (size_t)b * a.q_ld + h * 64 + ch * EPC, qv);
    if (SELF) {
        if (QF32) {
            load_f32_slabs<EPC>((const float*)a.knew + (size_t)b * a.kv_ld + h * 64 + ch * EPC, a.nslab, a.slab_stride, knv);
            load_f32_slabs<EPC>((const float*)a.vnew + (size_t)b * a.kv_ld + h * 64 + ch * EPC, a.nslab, a.slab_stride, vnv);
        } else {
            load_chunk<T, EPC>((const T*)a.knew + (size_t)b * a.kv_ld + h * 64 + ch * EPC, knv);
            load_chunk<T, EPC>((const T*)a.vnew + (size_t)b * a.kv_ld + h * 64 + ch * EPC, vnv);
        }
    }
    int n = a.n_keys;
    if (SELF) n = *a.step;  // keys already in the cache
    const float scale2 = a.scale * 1.4426950408889634f;

    const T* kc = (const T*)a.kcache + ((size_t)(b * a.H + h) * a.Tmax) * 64 + ch * EPC;
    const T* vc = (const T*)a.vcache + ((size_t)(b * a.H + h) * a.Tmax) * 64 + ch * EPC;
    float* s = sc[pib];
    const bool masked = !SELF && a.kmask != nullptr;

    auto load_batch = [&](const T* base, int j0, uint4 (&r)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * KPI + sub;
            r[u] = ld_stream(base + (size_t)(j < n ? j : (n > 0 ? n - 1 : 0)) * 64);
        }
    };
    const int jfirst = part * KB, jstep = NSPLIT * KB;  // this wave's key batches: jfirst, jfirst + jstep, ...

    uint4 cur[U], nxt[U], vfirst[U];
    if (jfirst < n) {
        load_batch(kc, jfirst, cur);
        load_batch(vc, jfirst, vfirst);  // the first V batch rides along: short contexts are one latency hop shorter
    }
    // key mask -> additive bias in LDS (cross-attention): coalesced byte loads, once per launch
    if (masked) {
        for (int j = part * 64 + lane; j < n; j += NSPLIT * 64) s[j] = a.kmask[(size_t)b * a.kmask_ld + j] ? 0.f : kNegD;
    }
    // NSPLIT == 1: a wave works on its own (clip, head) and its own slice of the LDS score buffer -- nothing to wait for
    if (masked && NSPLIT > 1) __syncthreads();

    // ---- phase 1: scores
    float mx = kNegD;
    for (int j0 = jfirst; j0 < n; j0 += jstep) {
        if (j0 + jstep < n) load_batch(kc, j0 + jstep, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * KPI + sub;
            float kv[EPC];
            cvt_chunk<T, EPC>(cur[u], kv);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) d = fmaf(qv[e], kv[e], d);
            d = LPK == 8 ? group8_sum(d) : row16_sum(d);  // == the xor butterfly over the key's lanes (common.hpp)
            if (j < n) {
                float sv = d * scale2;
                if (masked && s[j] != 0.f) sv = kNegD;
                if (ch == 0) s[j] = sv;
                mx = fmaxf(mx, sv);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = vfirst[u];
    int total = n;
    if (SELF) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) d = fmaf(qv[e], knv[e], d);
        d = LPK == 8 ? group8_sum(d) : row16_sum(d);
        const float sv = d * scale2;
        total = n + 1;
        if (part == 0) {  // the new key belongs to the first wave of the pair
            if (lane == 0) s[n] = sv;
            mx = fmaxf(mx, sv);
            if (active && sub == 0) {  // append to the cache for the following steps
                store_chunk<T, EPC>((T*)a.kcache + ((size_t)(b * a.H + h) * a.Tmax + n) * 64 + ch * EPC, knv);
                store_chunk<T, EPC>((T*)a.vcache + ((size_t)(b * a.H + h) * a.Tmax + n) * 64 + ch * EPC, vnv);
            }
        }
    }
    mx = wave_max(mx);
    if (NSPLIT > 1) {
        if (lane == 0) red_m[wave] = mx;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) mx = fmaxf(mx, red_m[pib * NSPLIT + p]);
    }

    // ---- phase 2: probabilities (unnormalised) + row sum (each wave handles a slice of the pair's keys)
    float lsum = 0.f;
    for (int j = part * 64 + lane; j < total; j += NSPLIT * 64) {
        const float p = exp2f(s[j] - mx);
        s[j] = p;
        lsum += p;
    }
    lsum = wave_sum_sel<sizeof(T) == 2>(lsum);
    if (NSPLIT > 1) {
        if (lane == 0) red_l[wave] = lsum;
        __syncthreads();
        lsum = 0.f;
#pragma unroll
        for (int p = 0; p < NSPLIT; ++p) lsum += red_l[pib * NSPLIT + p];
    }

    // ---- phase 3: o = sum_j p_j v_j
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int j0 = jfirst; j0 < n; j0 += jstep) {
        if (j0 + jstep < n) load_batch(vc, j0 + jstep, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * KPI + sub;
            float vv[EPC];
            cvt_chunk<T, EPC>(cur[u], vv);
            const float p = j < n ? s[j] : 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fmaf(p, vv[e], acc[e]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    if (SELF && part == 0 && sub == 0) {
        const float p = s[n];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = fmaf(p, vnv[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {  // xor LPK .. 32 butterfly over the key sub-groups (common.hpp xor_lane: no LDS round trips)
        if (LPK <= 8) acc[e] += xor_lane_f32<8>(acc[e]);
        acc[e] += xor_lane_f32<16>(acc[e]);
        acc[e] += xor_lane_f32<32>(acc[e]);
    }
    if (NSPLIT > 1) {  // combine the partial outputs of the pair's waves in wave order
        if (sub == 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) red_acc[wave][ch * EPC + e] = acc[e];
        }
        __syncthreads();
        if (part == 0 && sub == 0) {
#pragma unroll
            for (int p = 1; p < NSPLIT; ++p)
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += red_acc[pib * NSPLIT + p][ch * EPC + e];
        }
    }
    if (active && part == 0 && sub == 0) {
        const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] *= inv;
        store_chunk<T, EPC>((T*)a.out + (size_t)b * a.o_ld + h * 64 + ch * EPC, acc);
    }
}


// Multi-sample cross attention: SQ query rows (independent samples of the same clip) share one pass over
// the clip's K/V cache -- the reference's best-of-10 protocol (code/x_engine_pt.py:257) re-reads the same
// context ten times; here the cache is streamed once per (clip, head) and every key is scored against all SQ
// queries.  q/out rows of clip b are b*SQ .. b*SQ+SQ-1; scores live in LDS as [SQ][n].
template <typename T, int SQ, bool QF32>
__global__ __launch_bounds__(256) void decode_attn_multi_kernel(const DecodeAttnArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int LPK = 64 / EPC;
    constexpr int KPI = 64 / LPK;
    constexpr int U = 4;
    constexpr int KB = U * KPI;
    extern __shared__ __attribute__((aligned(16))) float dyn_sc[];  // [4 waves][SQ][npad]

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + wave;
    const bool active = pair < a.B * a.H;
    const int b = active ? pair / a.H : 0, h = active ? pair % a.H : 0;   // b = clip
    const int sub = lane / LPK, ch = lane % LPK;
    const int n = a.n_keys;
    const int npad = (n + 3) & ~3;
    const float scale2 = a.scale * 1.4426950408889634f;
    const T* kc = (const T*)a.kcache + ((size_t)(b * a.H + h) * a.Tmax) * 64 + ch * EPC;
    const T* vc = (const T*)a.vcache + ((size_t)(b * a.H + h) * a.Tmax) * 64 + ch * EPC;
    float* s = dyn_sc + (size_t)wave * SQ * npad;
    const bool masked = a.kmask != nullptr;

    auto load_batch = [&](const T* base, int j0, uint4 (&r)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * KPI + sub;
            r[u] = ld_stream(base + (size_t)(j < n ? j : n - 1) * 64);
        }
    };
    uint4 cur[U], nxt[U];
    load_batch(kc, 0, cur);
    float qv[SQ][EPC];
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
        const size_t row = (size_t)b * SQ + q;
        if (QF32)
            load_f32_slabs<EPC>((const float*)a.q + row * a.q_ld + h * 64 + ch * EPC, a.nslab, a.slab_stride, qv[q]);
        else
            load_chunk<T, EPC>((const T*)a.q + row * a.q_ld + h * 64 + ch * EPC, qv[q]);
    }
    float mx[SQ];
#pragma unroll
    for (int q = 0; q < SQ; ++q) mx[q] = kNegD;
    for (int j0 = 0; j0 < n; j0 += KB) {
        if (j0 + KB < n) load_batch(kc, j0 + KB, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * KPI + sub;
            float kv[EPC];
            cvt_chunk<T, EPC>(cur[u], kv);
            const bool dead = masked && j < n && a.kmask[(size_t)b * a.kmask_ld + j] == 0;
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < EPC; ++e) d = fmaf(qv[q][e], kv[e], d);
                d = LPK == 8 ? group8_sum(d) : row16_sum(d);
                if (j < n) {
                    const float sv = dead ? kNegD : d * scale2;
                    if (ch == 0) s[q * npad + j] = sv;
                    mx[q] = fmaxf(mx[q], sv);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    load_batch(vc, 0, cur);
#pragma unroll
    for (int q = 0; q < SQ; ++q) mx[q] = wave_max(mx[q]);
    __syncthreads();
    float lsum[SQ];
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
        float l = 0.f;
        for (int j = lane; j < n; j += 64) {
            const float p = exp2f(s[q * npad + j] - mx[q]);
            s[q * npad + j] = p;
            l += p;
        }
        lsum[q] = wave_sum_sel<sizeof(T) == 2>(l);
    }
    __syncthreads();
    float acc[SQ][EPC];
#pragma unroll
    for (int q = 0; q < SQ; ++q)
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[q][e] = 0.f;
    for (int j0 = 0; j0 < n; j0 += KB) {
        if (j0 + KB < n) load_batch(vc, j0 + KB, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * KPI + sub;
            float vv[EPC];
            cvt_chunk<T, EPC>(cur[u], vv);
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                const float p = j < n ? s[q * npad + j] : 0.f;
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[q][e] = fmaf(p, vv[e], acc[q][e]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            if (LPK <= 8) acc[q][e] += xor_lane_f32<8>(acc[q][e]);
            acc[q][e] += xor_lane_f32<16>(acc[q][e]);
            acc[q][e] += xor_lane_f32<32>(acc[q][e]);
        }
        if (active && sub == 0) {
            const float inv = lsum[q] > 0.f ? 1.0f / lsum[q] : 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[q][e] *= inv;
            store_chunk<T, EPC>((T*)a.out + ((size_t)b * SQ + q) * a.o_ld + h * 64 + ch * EPC, acc[q]);
        }
    }
}

}  // namespace

int launch_decode_attn(const DecodeAttnArgs& a, hipStream_t s) {
    DIMX_REQUIRE(a.q && a.kcache && a.vcache && a.out, DIMX_ERR_ARG, "decode_attn: null operand");
    DIMX_REQUIRE((a.knew == nullptr) == (a.vnew == nullptr), DIMX_ERR_ARG, "decode_attn: knew/vnew mismatch");
    DIMX_REQUIRE(a.knew == nullptr || a.step != nullptr, DIMX_ERR_ARG, "decode_attn: self attention needs a step counter");
    DIMX_REQUIRE(a.Tmax <= kMaxKeys && a.n_keys <= kMaxKeys, DIMX_ERR_ARG, "decode_attn: more than %d keys", kMaxKeys);
    if (a.rows_per_clip > 1) {
        // multi-sample cross attention: a.B = clips, rows = B * rows_per_clip
        DIMX_REQUIRE(a.knew == nullptr, DIMX_ERR_ARG, "decode_attn: rows_per_clip applies to cross attention only");
        const int S = a.rows_per_clip;
        const int npad = (a.n_keys + 3) & ~3;
        const size_t lds = (size_t)4 * S * npad * sizeof(float);
        DIMX_REQUIRE(lds <= 150 * 1024, DIMX_ERR_ARG, "decode_attn: %d samples x %d keys do not fit the LDS score buffer", S,
                     a.n_keys);
        dim3 grid(ceil_div(a.B * a.H, 4)), block(256);
#define DM(TT, SS, QQ)                                                                                          \
    do {                                                                                                        \
        (void)hipFuncSetAttribute((const void*)decode_attn_multi_kernel<TT, SS, QQ>,                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        hipLaunchKernelGGL((decode_attn_multi_kernel<TT, SS, QQ>), grid, block, lds, s, a);                     \
    } while (0)
#define DM_S(TT, QQ)                         \
    do {                                     \
        switch (S) {                         \
            case 2: DM(TT, 2, QQ); break;    \
            case 4: DM(TT, 4, QQ); break;    \
            case 5: DM(TT, 5, QQ); break;    \
            case 8: DM(TT, 8, QQ); break;    \
            case 10: DM(TT, 10, QQ); break;  \
            default:                         \
                set_error("decode_attn: rows_per_clip %d not in {2,4,5,8,10}", S); \
                return DIMX_ERR_ARG;         \
        }                                    \
    } while (0)
        if (a.dtype == DIMX_BF16) { if (a.q_f32) DM_S(bf16, true); else DM_S(bf16, false); }
        else { if (a.q_f32) DM_S(float, true); else DM_S(float, false); }
#undef DM_S
#undef DM
        DIMX_HIP(hipGetLastError());
        return DIMX_OK;
    }
    // waves per (clip, head): enough waves to cover the chip (~12 per CU) without splitting large batches
    const int pairs = a.B * a.H;
    int nsplit = 1;
    if (pairs * 2 <= 3072) nsplit = 2;
    if (pairs * 4 <= 3072) nsplit = 4;
    if (a.force_nsplit == 1 || a.force_nsplit == 2 || a.force_nsplit == 4) nsplit = a.force_nsplit;
    dim3 grid(ceil_div(pairs, 4 / nsplit)), block(256);
    const bool self = a.knew != nullptr;
#define DA_LAUNCH(TT, SS, QQ, NS) hipLaunchKernelGGL((decode_attn_kernel<TT, SS, QQ, NS>), grid, block, 0, s, a)
#define DA_NS(TT, SS, QQ)                          \
    do {                                           \
        if (nsplit == 4) DA_LAUNCH(TT, SS, QQ, 4); \
        else if (nsplit == 2) DA_LAUNCH(TT, SS, QQ, 2); \
        else DA_LAUNCH(TT, SS, QQ, 1);             \
    } while (0)
    if (a.dtype == DIMX_BF16) {
        if (a.q_f32) { if (self) DA_NS(bf16, true, true); else DA_NS(bf16, false, true); }
        else { if (self) DA_NS(bf16, true, false); else DA_NS(bf16, false, false); }
    } else {
        if (a.q_f32) { if (self) DA_NS(float, true, true); else DA_NS(float, false, true); }
        else { if (self) DA_NS(float, true, false); else DA_NS(float, false, false); }
    }
#undef DA_NS
#undef DA_LAUNCH
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
