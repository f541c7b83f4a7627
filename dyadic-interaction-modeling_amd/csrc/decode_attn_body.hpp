// decode_attn_body.hpp -- the one-query decode attention as a device function (see decode_attn.hip for what it computes and
// why it is shaped the way it is).  In a header because two kernels run it: decode_attn_kernel (decode_attn.hip) and the
// attention / GEMM co-residency probe (gemm.hip, fused_probe_kernel).
#pragma once
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
// the same through L1-bypassing loads (sc1): rows another CU of this XCD wrote during the SAME launch (the layer kernel of
// chain.hip: the cross-attention query is the chain's q projection) -- a plain load could be served a stale L1 line
template <int EPC> __device__ __forceinline__ void load_f32_chunk_sc1(const float* p, float (&v)[EPC]) {
#pragma unroll
    for (int i = 0; i < EPC; ++i) v[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
// The kernel body as a device function: `block_id` is the block's index among the attention blocks, NW the waves of the
// block that take part (4 in decode_attn_kernel; 8 when a 512-thread launch shares its CUs with GEMM blocks, the fusion
// probe of gemm.hip); the LDS arrays come from the caller: sc [NW / NSPLIT][sc_stride] scores, red_m / red_l [NW],
// red_acc [NW][64].
template <typename T, bool SELF, bool QF32, int NSPLIT, int NW, bool QSC1 = false>
__device__ __forceinline__ void decode_attn_body(const DecodeAttnArgs& a, int block_id, float* sc, int sc_stride, float* red_m,
                                                 float* red_l, float* red_acc) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int LPK = 64 / EPC;   // lanes per key: 8 (bf16) / 16 (f32)
    constexpr int KPI = 64 / LPK;   // keys per wave-wide load: 8 / 4
    constexpr int U = 4;            // key groups per batch (8 measured no faster on MI355X)
    constexpr int KB = U * KPI;     // keys per batch
    constexpr int PPB = NW / NSPLIT; // (clip, head) pairs per block

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pib = wave / NSPLIT, part = wave % NSPLIT;
    const int pair = block_id * PPB + pib;
    const bool active = pair < a.B * a.H;
    const int b = active ? pair / a.H : 0, h = active ? pair % a.H : 0;
    const int sub = lane / LPK, ch = lane % LPK;
    // Everything that does not depend on the step counter is requested first: q and (self attention) this step's k / v
    // rows.  At step 0 the kernel is nothing but its dependent-load chain (step counter -> cache address -> q -> new k/v ->
    // store: 8.2 us per launch in round 2, four launches per decode step); issued here the three loads overlap the
    // counter read and each other.
    float qv[EPC], knv[EPC], vnv[EPC];
    if (QSC1)  // one f32 slab, written by other CUs of this XCD during this launch
        load_f32_chunk_sc1<EPC>((const float*)a.q + (size_t)b * a.q_ld + h * 64 + ch * EPC, qv);
    else if (QF32)
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
    float* s = sc + (size_t)pib * sc_stride;
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
            for (int e = 0; e < EPC; ++e) red_acc[wave * 64 + ch * EPC + e] = acc[e];
        }
        __syncthreads();
        if (part == 0 && sub == 0) {
#pragma unroll
            for (int p = 1; p < NSPLIT; ++p)
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += red_acc[(pib * NSPLIT + p) * 64 + ch * EPC + e];
        }
    }
    if (active && part == 0 && sub == 0) {
        const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] *= inv;
        store_chunk<T, EPC>((T*)a.out + (size_t)b * a.o_ld + h * 64 + ch * EPC, acc);
    }
}

}  // namespace
}  // namespace dimx
