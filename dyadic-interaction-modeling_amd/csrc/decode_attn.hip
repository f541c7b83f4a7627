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

// SELF = true: self-attention form (append this step's k/v, keys = step counter + 1);
// SELF = false: cross-attention form (fixed n_keys, optional key mask).  Two instantiations so that the two
// launch shapes show up as separate rows of a rocprofv3 kernel trace.
template <typename T, bool SELF>
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecodeAttnArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int LPK = 64 / EPC;   // lanes per key: 8 (bf16) / 16 (f32)
    constexpr int KPI = 64 / LPK;   // keys per wave-wide load: 8 / 4
    __shared__ float sc[4][kMaxKeys];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + wave;
    const bool active = pair < a.B * a.H;
    const int b = active ? pair / a.H : 0, h = active ? pair % a.H : 0;
    const int sub = lane / LPK, ch = lane % LPK;
    constexpr bool self = SELF;
    int n = a.n_keys;
    if (self) n = *a.step;  // keys already in the cache
    const float scale2 = a.scale * 1.4426950408889634f;

    const T* kc = (const T*)a.kcache + ((size_t)(b * a.H + h) * a.Tmax) * 64 + ch * EPC;
    const T* vc = (const T*)a.vcache + ((size_t)(b * a.H + h) * a.Tmax) * 64 + ch * EPC;
    float* s = sc[wave];

    float qv[EPC];
    load_chunk<T, EPC>((const T*)a.q + (size_t)b * a.q_ld + h * 64 + ch * EPC, qv);

    // ---- phase 1: scores
    float mx = kNegD;
    for (int j0 = 0; j0 < n; j0 += 4 * KPI) {  // wave-uniform trip count, 4 loads in flight per lane
        float kv[4][EPC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * KPI + sub;
            load_chunk<T, EPC>(kc + (size_t)(j < n ? j : n - 1) * 64, kv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * KPI + sub;
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) d = fmaf(qv[e], kv[u][e], d);
#pragma unroll
            for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o);
            if (j < n) {
                float sv = d * scale2;
                if (a.kmask && a.kmask[(size_t)b * a.kmask_ld + j] == 0) sv = kNegD;
                if (ch == 0) s[j] = sv;
                mx = fmaxf(mx, sv);
            }
        }
    }
    float knv[EPC], vnv[EPC];
    int total = n;
    if (self) {
        load_chunk<T, EPC>((const T*)a.knew + (size_t)b * a.kv_ld + h * 64 + ch * EPC, knv);
        load_chunk<T, EPC>((const T*)a.vnew + (size_t)b * a.kv_ld + h * 64 + ch * EPC, vnv);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) d = fmaf(qv[e], knv[e], d);
#pragma unroll
        for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o);
        const float sv = d * scale2;
        if (lane == 0) s[n] = sv;
        mx = fmaxf(mx, sv);
        total = n + 1;
        if (active && sub == 0) {  // append to the cache for the following steps
            store_chunk<T, EPC>((T*)a.kcache + ((size_t)(b * a.H + h) * a.Tmax + n) * 64 + ch * EPC, knv);
            store_chunk<T, EPC>((T*)a.vcache + ((size_t)(b * a.H + h) * a.Tmax + n) * 64 + ch * EPC, vnv);
        }
    }
    mx = wave_max(mx);
    __syncthreads();

    // ---- phase 2: probabilities (unnormalised) + row sum
    float lsum = 0.f;
    for (int j = lane; j < total; j += 64) {
        const float p = exp2f(s[j] - mx);
        s[j] = p;
        lsum += p;
    }
    lsum = wave_sum(lsum);
    __syncthreads();

    // ---- phase 3: o = sum_j p_j v_j
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 < n; j0 += 4 * KPI) {
        float vv[4][EPC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * KPI + sub;
            load_chunk<T, EPC>(vc + (size_t)(j < n ? j : n - 1) * 64, vv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * KPI + sub;
            const float p = j < n ? s[j] : 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fmaf(p, vv[u][e], acc[e]);
        }
    }
    if (self && sub == 0) {
        const float p = s[n];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = fmaf(p, vnv[e], acc[e]);
    }
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += __shfl_xor(acc[e], o);
    if (active && sub == 0) {
        const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] *= inv;
        store_chunk<T, EPC>((T*)a.out + (size_t)b * a.o_ld + h * 64 + ch * EPC, acc);
    }
}

}  // namespace

int launch_decode_attn(const DecodeAttnArgs& a, hipStream_t s) {
    DIMX_REQUIRE(a.q && a.kcache && a.vcache && a.out, DIMX_ERR_ARG, "decode_attn: null operand");
    DIMX_REQUIRE((a.knew == nullptr) == (a.vnew == nullptr), DIMX_ERR_ARG, "decode_attn: knew/vnew mismatch");
    DIMX_REQUIRE(a.knew == nullptr || a.step != nullptr, DIMX_ERR_ARG, "decode_attn: self attention needs a step counter");
    DIMX_REQUIRE(a.Tmax <= kMaxKeys && a.n_keys <= kMaxKeys, DIMX_ERR_ARG, "decode_attn: more than %d keys", kMaxKeys);
    dim3 grid(ceil_div(a.B * a.H, 4)), block(256);
    const bool self = a.knew != nullptr;
    if (a.dtype == DIMX_BF16) {
        if (self) hipLaunchKernelGGL((decode_attn_kernel<bf16, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((decode_attn_kernel<bf16, false>), grid, block, 0, s, a);
    } else {
        if (self) hipLaunchKernelGGL((decode_attn_kernel<float, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((decode_attn_kernel<float, false>), grid, block, 0, s, a);
    }
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
