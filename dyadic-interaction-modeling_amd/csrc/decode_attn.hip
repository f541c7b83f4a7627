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
#include "decode_attn_body.hpp"

namespace dimx {
namespace {

template <typename T, bool SELF, bool QF32, int NSPLIT>
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecodeAttnArgs a) {
    constexpr int PPB = 4 / NSPLIT;  // (clip, head) pairs per block
    __shared__ float sc[PPB][kMaxKeys];
    __shared__ float red_m[4], red_l[4];
    __shared__ float red_acc[4][64];
    decode_attn_body<T, SELF, QF32, NSPLIT, 4>(a, blockIdx.x, &sc[0][0], kMaxKeys, red_m, red_l, &red_acc[0][0]);
}


// Round 6, two-engine experiment (DIMX_GEN_EXCL=1): ONE 12-wave block per clip (a wave per head, no block barrier) whose LDS
// request is more than half a CU's, so that a block owns its CU: B blocks occupy B CUs and the other engine's one-block-per-CU
// kernels take the rest -- the placement a whole-XCD CU mask would have given, which this platform does not honour
// (profiles/r06_xcdmask_probe.txt).
template <bool SELF>
__global__ __launch_bounds__(768) void decode_attn_clip_kernel(const DecodeAttnArgs a, int sc_stride) {
    extern __shared__ __attribute__((aligned(16))) float sc_clip[];
    decode_attn_body<bf16, SELF, true, 1, 12>(a, blockIdx.x, sc_clip, sc_stride, nullptr, nullptr, nullptr);
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
    if (a.clip_blocks && a.dtype == DIMX_BF16 && a.q_f32 && a.H == 12) {
        const int nk = a.knew ? a.Tmax : a.n_keys;
        const int sc_stride = (nk + 15) / 16 * 16;
        size_t lds = (size_t)12 * sc_stride * sizeof(float);
        if (lds < 84 * 1024) lds = 84 * 1024;   // two of these (or one and a 64 x 72 decode GEMM block) never share a CU
        DIMX_REQUIRE(lds <= 160 * 1024, DIMX_ERR_ARG, "decode_attn: %d keys do not fit the one-block-per-clip form", nk);
        if (a.knew) {
            (void)hipFuncSetAttribute((const void*)decode_attn_clip_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((decode_attn_clip_kernel<true>), dim3(a.B), dim3(768), lds, s, a, sc_stride);
        } else {
            (void)hipFuncSetAttribute((const void*)decode_attn_clip_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((decode_attn_clip_kernel<false>), dim3(a.B), dim3(768), lds, s, a, sc_stride);
        }
        DIMX_HIP(hipGetLastError());
        return DIMX_OK;
    }
    // waves per (clip, head): enough waves to cover the chip (~12 per CU) without splitting large batches
    const int pairs = a.B * a.H;
    int nsplit = 1;
    if (pairs * 2 <= 3072) nsplit = 2;
    if (pairs * 4 <= 3072) nsplit = 4;
    // f32 parity mode: the wave count decides the summation order, and a rank's shard of a batch must reproduce the rows of the
    // whole batch bit for bit -- so it may depend on the cache length only, never on the batch (round 4; until then
    // a 128-clip shard ran 2 waves per pair, the 256-clip batch 1, and equal tokens were a matter of no near-tie being hit)
    if (a.dtype != DIMX_BF16) nsplit = a.Tmax >= 1024 ? 4 : (a.Tmax >= 512 ? 2 : 1);
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
