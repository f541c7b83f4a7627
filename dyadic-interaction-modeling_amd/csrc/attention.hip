// attention.hip -- flash-style softmax(Q K^T * scale + mask) V for the parallel (non-autoregressive)
// attention blocks of the path:
//   * VQ-VAE blocks, 8 heads x 48 (listener / SLMFT speaker) or 8 x 96 (legacy speaker, hidden 768),
//     scale hidden^-0.5, keys limited to the clip length
//     (reference code/models/lib/base_models.py:125-146);
//   * x-transformers encoder self-attention (causal + key padding mask), teacher-forced decoder
//     self-attention (causal + random key mask) and cross-attention over the speaker context
//     (12 heads x 64, scale 64^-0.5; ctor sites code/seq2seq_pretrain.py:388-418).
// Masked scores take a large negative finite value exactly like x-transformers' -finfo.max fill, so a
// row with at least one visible key gives identical probabilities.
//
// Everything is computed in the "swapped" form so that softmax statistics are lane-local:
//   S^T[key][query] = K . Q^T   (MFMA A = K tile rows from LDS, B = Q fragment kept in registers)
//   O^T[d][query]   = V^T . P^T (MFMA A = V^T tile rows from LDS, B = P straight from the S registers)
// A lane owns one query (lane & 31) and, per 32-key tile, the 16 keys (r&3) + 8*(r>>2) + 4*(lane>>5);
// the same key order is used for the V^T operand, so P never moves between lanes.  The V operand is
// consumed transposed ([B,H,D,Lk]), which is how the QKV GEMM epilogue writes it.
// One wave = 32 queries, NW waves per block share the K / V^T tiles (64 keys) staged in LDS.
#include <stdlib.h>

#include "common.hpp"

namespace dimx {

namespace {

template <typename T> struct MmaT;
template <> struct MmaT<bf16> {
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct MmaT<float> {
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        const float4 fa = __builtin_bit_cast(float4, a), fb = __builtin_bit_cast(float4, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
    }
};

constexpr float kNeg = -3.0e38f;

// bf16 perf mode: the bare v_exp_f32 (arguments are <= 0, results that underflow flush to 0, which is what a
// softmax wants); f32 parity mode keeps exp2f
template <typename T> __device__ __forceinline__ float fast_exp2(float x) {
    if (sizeof(T) == 2) return __builtin_amdgcn_exp2f(x);
    return exp2f(x);
}

// DK = head-dim extent the kernel reduces over (64 for D in {48,64}; 96 for the 8 x 96 heads of the legacy
// speaker VQ-VAE, hidden 768).  K rows are padded to DKP = 64 / 128 elements so the XOR swizzle stays a power
// of two; the V^T tile has DK rows (NB = DK/32 output blocks) of 64 keys.
// VROW (bf16, DK 64): V arrives row-major [key][d] like K (the fused q/k/v projection then has three row-contiguous destinations
// and runs on the two-phase 256 x 256 GEMM); the transposition happens on the way into LDS -- a thread owns a column PAIR of
// 4-key groups (one 4-byte load per key, coalesced over the 32 pairs of a key row), separates the two columns with v_perm_b32 and
// writes two 8-byte pieces into the same swizzled V^T tile the rest of the kernel reads.  Same registers as the 16-byte staging.
template <typename T, int NW, int DK, bool VROW = false>
__global__ __launch_bounds__(NW * 64) void attn_kernel(const AttnArgs a) {
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;     // elements per 16-byte chunk
    constexpr int DKP = DK == 64 ? 64 : 128;
    constexpr int ROWBK = DKP * ES;  // bytes per K row in LDS
    constexpr int CPRK = ROWBK / 16; // 16-byte chunks per K row: 8 / 16 / 32
    constexpr int ROWB = 64 * ES;    // bytes per V^T row (64 keys)
    constexpr int CPR = ROWB / 16;   // chunks per V^T row: 8 (bf16) / 16 (f32)
    constexpr int NKS = DK * ES / 32;  // 32-byte k-steps over the head dim
    constexpr int NB = DK / 32;      // 32-row output blocks
    constexpr int NT = NW * 64;
    constexpr int UB = 4 * ES;       // bytes of a 4-key unit in the V^T tile
    __shared__ __attribute__((aligned(16))) unsigned char sK[64 * ROWBK];
    __shared__ __attribute__((aligned(16))) unsigned char sV[NB * 32 * ROWB];
    auto swz = [](int row) { return CPRK == 8 ? ((row >> 1) & 7) : (row & (CPRK - 1)); };

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk0 = blockIdx.x * (32 * NW);
    const int qi = qblk0 + wave * 32 + l31;
    const int qc = qi < a.Lq ? qi : a.Lq - 1;

    const T* __restrict__ Q = (const T*)a.q + (size_t)b * a.q_sb + (size_t)h * a.q_sh;
    const T* __restrict__ K = (const T*)a.k + (size_t)b * a.k_sb + (size_t)h * a.k_sh;
    const T* __restrict__ Vt = (const T*)a.vt + (size_t)b * a.v_sb + (size_t)h * a.v_sh;

    // Q fragment: chunk 2*ks + half of row qc, zero beyond D
    uint4 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int c = 2 * ks + half;
        qf[ks] = (c * EPC < a.D) ? *(const uint4*)(Q + (size_t)qc * a.q_st + c * EPC) : make_uint4(0, 0, 0, 0);
    }

    int kmax = a.Lk;
    const int len_b = a.lens ? a.lens[b] : a.Lk;
    kmax = len_b < kmax ? len_b : kmax;
    if (a.causal) {
        int last = qblk0 + 32 * NW - 1;
        last = last < a.Lq - 1 ? last : a.Lq - 1;
        kmax = (last + 1) < kmax ? (last + 1) : kmax;
    }
    const int ntiles = (kmax + 63) / 64;

    float m_run = kNeg, l_run = 0.f;
    f32x16_t ot[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    const float scale2 = a.scale * 1.4426950408889634f;

    // Staging is software-pipelined through registers (round 3): the global loads of tile t + 1 are issued right after tile
    // t's operands went to LDS and land while tile t is being computed (S^T, softmax, P . V: ~1.5 k cycles per wave) -- the
    // loop used to load, wait, store and only then compute, with nothing but other blocks to cover the L2 latency.
    constexpr int KI = 64 * CPRK / NT, VI = NB * 32 * CPR / NT;
    uint4 kreg[KI], vreg[VI];
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPRK, c = idx % CPRK;
            int key = j0 + row;
            key = key < a.Lk ? key : a.Lk - 1;
            kreg[i] = (c * EPC < a.D) ? *(const uint4*)(K + (size_t)key * a.k_st + c * EPC) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (VROW) {
            static_assert(!VROW || (sizeof(T) == 2 && DK == 64 && NW == 2), "row-major V: bf16, 64 head columns, 2 waves");
            // thread = (column pair dp, key-group selector): 4 groups of 4 keys, one 4-byte load per key (2 columns of it)
            const int dp = tid & 31, ksel = tid >> 5;
            const uint16_t* vp = (const uint16_t*)Vt + 2 * dp;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // predicated loads, NOT load + select: a select consumes the value at once and the wave would wait for the
                    // prefetch instead of computing under it (measured: 202 -> 397 us per call)
                    const int key = j0 + 16 * p + 4 * ksel + i;
                    e[i] = (2 * dp < a.D && key < a.Lk) ? *(const uint32_t*)(vp + (size_t)key * a.v_st) : 0u;
                }
                vreg[p] = make_uint4(e[0], e[1], e[2], e[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < VI; ++i) {
                const int idx = tid + i * NT;
                const int d = idx / CPR, c = idx % CPR;
                const int jj = j0 + c * EPC;
                vreg[i] = (d < a.D && a.Lk - jj > 0) ? *(const uint4*)(Vt + (size_t)d * a.v_sd + jj) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    // bf16 only: in the f32 parity mode the 16 extra 16-byte registers cost a wave of occupancy (248 -> 314 VGPRs)
    constexpr bool PIPE = ES == 2;
    if (PIPE && ntiles > 0) load_tile(0);

    for (int tile = 0; tile < ntiles; ++tile) {
        const int j0 = tile * 64;
        if (!PIPE) load_tile(j0);
        __syncthreads();
        // ---- K tile [64 keys][64 d] and V^T tile [64 d][64 keys] from the registers to LDS
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPRK, c = idx % CPRK;
            *(uint4*)(sK + row * ROWBK + ((c ^ swz(row)) << 4)) = kreg[i];
        }
        if constexpr (VROW) {
            // keys 4 g4 .. 4 g4 + 3 of columns 2 dp (low halves) and 2 dp + 1 (high halves): two 8-byte pieces of the V^T tile, at the
            // place the 16-byte staging of the transposed form puts them (chunk g4 / 2 swizzled by f / 2, halves swapped for odd f)
            const int dp = tid & 31, ksel = tid >> 5, f = dp & 15;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int g4 = 4 * p + ksel;
                const uint4 w = vreg[p];
                const uint2 ev = make_uint2(__builtin_amdgcn_perm(w.y, w.x, 0x05040100u), __builtin_amdgcn_perm(w.w, w.z, 0x05040100u));
                const uint2 od = make_uint2(__builtin_amdgcn_perm(w.y, w.x, 0x07060302u), __builtin_amdgcn_perm(w.w, w.z, 0x07060302u));
                const int off = (((g4 >> 1) ^ (f >> 1)) << 4) + (((g4 & 1) ^ (f & 1)) << 3);
                *(uint2*)(sV + (2 * dp) * ROWB + off) = ev;
                *(uint2*)(sV + (2 * dp + 1) * ROWB + off) = od;
            }
        } else
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPR, c = idx % CPR;
            {
                const int d = row;
                const int jj = j0 + c * EPC;
                const int nvalid = a.Lk - jj;
                uint4 v = vreg[i];
                if (d < a.D && nvalid > 0 && nvalid < EPC) {  // zero the keys beyond Lk: 0 * garbage must stay 0
                    if (ES == 2) {
                        uint16_t e[8];
                        *(uint4*)e = v;
#pragma unroll
                        for (int q = 0; q < 8; ++q) e[q] = q < nvalid ? e[q] : (uint16_t)0;
                        v = *(uint4*)e;
                    } else {
                        uint32_t e[4];
                        *(uint4*)e = v;
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = q < nvalid ? e[q] : 0u;
                        v = *(uint4*)e;
                    }
                }
                if (ES == 2) {
                    const int f = (d >> 1) & 15;
                    if (f & 1) v = make_uint4(v.z, v.w, v.x, v.y);
                    *(uint4*)(sV + d * ROWB + ((c ^ (f >> 1)) << 4)) = v;
                } else {
                    *(uint4*)(sV + d * ROWB + ((c ^ (d & 15)) << 4)) = v;
                }
            }
        }
        if (PIPE && tile + 1 < ntiles) load_tile(j0 + 64);  // in flight during this tile's compute
        // ---- key validity bits of this tile (wave-uniform 64-bit mask)
        const int jl = j0 + lane;
        bool kv = jl < a.Lk && jl < len_b;
        if (a.kmask && jl < a.Lk) kv = kv && a.kmask[(size_t)b * a.kmask_ld + jl] != 0;
        const unsigned long long kbits = __ballot(kv);
        __syncthreads();

        // ---- S^T = K . Q^T
        f32x16_t st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
            const int row = 32 * kt + l31;
            const int sw = swz(row);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int c = 2 * ks + half;
                const uint4 kf = *(const uint4*)(sK + row * ROWBK + ((c ^ sw) << 4));
                MmaT<T>::run(st[kt], kf, qf[ks]);
            }
        }
        // ---- mask, online softmax (lane-local: this lane's query, 32 of the tile's 64 keys).  This part, not the
        // MFMAs, bounds the kernel (VALU: ~3 k cycles per tile and wave against 512 MFMA cycles), so interior
        // tiles -- every key valid and, if causal, wholly below the diagonal of this wave's queries -- skip the
        // masking, and the masked path tests compile-time bit positions of a pre-shifted 32-bit validity word.
        const int qlo = qblk0 + wave * 32;  // smallest query index of this wave
        const bool interior = kbits == ~0ull && (!a.causal || j0 + 63 <= qlo);
        float mx = kNeg;
        if (interior) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = st[kt][r] * scale2;
                    st[kt][r] = sv;
                    mx = fmaxf(mx, sv);
                }
        } else {
            const int dq = a.causal ? j0 + 4 * half - qi : -(1 << 30);  // key (32kt + c) is visible iff 32kt + c + dq <= 0
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const uint32_t w = (uint32_t)(kbits >> (32 * kt)) >> (4 * half);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (r & 3) + 8 * (r >> 2);
                    const bool ok = (w & (1u << c)) != 0u && (32 * kt + c + dq <= 0);
                    const float sv = ok ? st[kt][r] * scale2 : kNeg;
                    st[kt][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
        }
        mx = fmaxf(mx, xor_lane_f32<32>(mx));
        const float m_new = fmaxf(m_run, mx);
        // bf16 mode: the running maximum (and with it the 32 accumulator rescales + one exp per tile) is only moved when some
        // query's maximum grew by more than 2^8 -- probabilities relative to a slightly stale maximum are at most 256, well
        // inside f32 / bf16 range, and the final normalisation divides it out.  The f32 parity mode keeps the exact form.
        bool rescale = true;
        if (ES == 2) rescale = __any((m_new - m_run) > 8.0f);
        if (rescale) {
            const float alpha = fast_exp2<T>(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2<T>(st[kt][r] - m_run);
                st[kt][r] = p;
                psum += p;
            }
        l_run += psum;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const int d = 32 * blk + l31;
            const unsigned char* vrow = sV + d * ROWB;
            if (ES == 2) {
                const int f = (d >> 1) & 15;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int u0 = 8 * kt + 4 * s + half, u1 = u0 + 2;
                        const uint2 v0 = *(const uint2*)(vrow + ((u0 ^ f) * UB));
                        const uint2 v1 = *(const uint2*)(vrow + ((u1 ^ f) * UB));
                        const uint4 va = make_uint4(v0.x, v0.y, v1.x, v1.y);
                        uint4 pb;
                        pb.x = pack_bf16x2(st[kt][8 * s + 0], st[kt][8 * s + 1]);
                        pb.y = pack_bf16x2(st[kt][8 * s + 2], st[kt][8 * s + 3]);
                        pb.z = pack_bf16x2(st[kt][8 * s + 4], st[kt][8 * s + 5]);
                        pb.w = pack_bf16x2(st[kt][8 * s + 6], st[kt][8 * s + 7]);
                        MmaT<T>::run(ot[blk], va, pb);
                    }
            } else {
                const int f = d & 15;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int u = 8 * kt + 2 * q + half;
                        const uint4 va = *(const uint4*)(vrow + ((u ^ f) * UB));
                        const float4 pf = make_float4(st[kt][4 * q], st[kt][4 * q + 1], st[kt][4 * q + 2],
                                                      st[kt][4 * q + 3]);
                        MmaT<T>::run(ot[blk], va, __builtin_bit_cast(uint4, pf));
                    }
            }
        }
    }

    // ---- finish: combine the two lane halves' row sums, normalise, store O[q][d]
    const float l_tot = l_run + xor_lane_f32<32>(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qi < a.Lq) {
        T* orow = (T*)a.o + (size_t)b * a.o_sb + (size_t)qi * a.o_st + (size_t)h * a.o_sh;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * blk + 8 * g + 4 * half;
                if (d0 < a.D) {
                    const float o0 = ot[blk][4 * g] * inv, o1 = ot[blk][4 * g + 1] * inv;
                    const float o2 = ot[blk][4 * g + 2] * inv, o3 = ot[blk][4 * g + 3] * inv;
                    if (ES == 2) {
                        *(uint2*)(orow + d0) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
                    } else {
                        *(float4*)(orow + d0) = make_float4(o0, o1, o2, o3);
                    }
                }
            }
    }
}

}  // namespace

int launch_attention(const AttnArgs& a, hipStream_t s) {
    DIMX_REQUIRE(a.q && a.k && a.vt && a.o, DIMX_ERR_ARG, "attention: null operand");
    DIMX_REQUIRE(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0, DIMX_ERR_ARG, "attention: empty shape");
    DIMX_REQUIRE(a.D == 48 || a.D == 64 || a.D == 96, DIMX_ERR_ARG, "attention: head dim %d not in {48,64,96}", a.D);
    const int epc = a.dtype == DIMX_BF16 ? 8 : 4;
    constexpr int NW = 2;
    dim3 grid(ceil_div(a.Lq, 32 * NW), a.H, a.B), block(NW * 64);
    if (a.v_rows) {
        // round 4: row-major V with 48- / 64-wide heads takes the shared-tile kernel of attention_tr.hip (DIMX_ATTN_OLD=1: this file's
        // VROW form, for A/B runs)
        static const bool old_form = getenv("DIMX_ATTN_OLD") != nullptr;
        if (!old_form && a.dtype == DIMX_BF16 && (a.D == 48 || a.D == 64)) return launch_attention_tr(a, s);
        DIMX_REQUIRE(a.dtype == DIMX_BF16 && a.D <= 64 && a.v_st > 0 && a.q_st % epc == 0 && a.k_st % epc == 0 && a.o_st % 4 == 0,
                     DIMX_ERR_ARG, "attention: row-major V needs bf16 operands and a head dim <= 64");
        hipLaunchKernelGGL((attn_kernel<bf16, NW, 64, true>), grid, block, 0, s, a);
        DIMX_HIP(hipGetLastError());
        return DIMX_OK;
    }
    DIMX_REQUIRE(a.v_sd % epc == 0 && a.v_sd >= a.Lk && a.q_st % epc == 0 && a.k_st % epc == 0 && a.o_st % 4 == 0,
                 DIMX_ERR_ARG, "attention: strides must keep 16-byte alignment (v_sd=%ld)", a.v_sd);
    if (a.D == 96) {
        if (a.dtype == DIMX_BF16)
            hipLaunchKernelGGL((attn_kernel<bf16, NW, 96>), grid, block, 0, s, a);
        else
            hipLaunchKernelGGL((attn_kernel<float, NW, 96>), grid, block, 0, s, a);
    } else if (a.dtype == DIMX_BF16) {
        hipLaunchKernelGGL((attn_kernel<bf16, NW, 64>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((attn_kernel<float, NW, 64>), grid, block, 0, s, a);
    }
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
