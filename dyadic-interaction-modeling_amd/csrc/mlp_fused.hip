// mlp_fused.hip -- one launch for a whole pre-norm feed-forward sublayer of the prefill (bf16 perf mode):
//     x <- x + W2 . gelu(W1 . LayerNorm(x) + b1) + b2          x [M, C] f32 in place, C = 384, hidden F = 1536
// Reference: the MLP sublayers of the VQ-VAE transformers (code/models/lib/base_models.py:56-68, 148-170: pre-LN, tanh-GELU) and
// x-transformers' FeedForward inside the encoders (pre-LN without bias, erf-GELU) -- 20 of them per forward of the
// headline workload at M = 76 800 rows.
//
// Why: as two library GEMMs + a LayerNorm pass the sublayer took 28 + 223 + 189 us (in situ, profiles/r04): K = 384 gives a 256 x 256
// tile only 6 k-tiles between a prologue and a GELU epilogue (16 % of the MFMA peak), and the [M, 1536] activation makes a round trip
// through HBM (472 MB) that nothing needs.  Here the hidden activation never leaves the registers -- the structure of a flash attention
// with GELU in the place of the softmax:
//   * a wave owns 32 rows.  It loads them once, computes the LayerNorm statistics in registers (a row lives in lanes l and l + 32) and
//     keeps bf16(LN(x))^T as the B operands of the first product for the whole kernel (25 fragments: the 25th is the constant 1 that
//     carries b1 as a bf16 hi + lo pair in two k-slots of W1);
//   * per chunk of 32 hidden units:  H^T [32 hidden x 32 rows] = W1_chunk . LN(x)^T  (25 v_mfma_f32_32x32x16_bf16, A fragments from
//     LDS), GELU lane-locally on the 16 accumulator registers, and -- the swapped-operand trick of csrc/train_attn.hip -- those
//     registers ARE the two B fragments of  Out^T [384 x 32 rows] += W2[:, chunk] . gelu(H^T)  (24 MFMAs into 12 resident
//     accumulators = 192 registers) once W2's k-slots are stored in accumulator-row order;
//   * the weights are packed ONCE on the host (dimx_load_weights) into exactly the LDS image of a chunk -- 25 + 24 fragments of
//     1 KiB, lane-linear -- so a chunk is 49 linear LDS-DMA pieces (global_load_lds, 16 B per lane) and every fragment read is a
//     conflict-free ds_read_b128 at base + lane * 16;
//   * epilogue: + b2 + the residual x, f32, in place.
// One block = 4 waves = 128 rows, one wave per SIMD (~400 registers); W1 and W2 in 3-slot LDS rings (147 KiB); the chunk loop is
// software-pipelined three deep (first product of chunk c + 2, GELU of c + 1, second product of c) so that the GELU and the LDS
// reads issue between MFMAs.
// Algorithmic traffic: x read twice + written once (3 x 118 MB at M = 76 800) + 2.4 MB of weights per 128 rows from L2; 181 GFLOP.
#include "common.hpp"

namespace dimx {
namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int kC = 384;            // model width (K of the first product, N of the second)
constexpr int kKS = kC / 16;       // k-steps of the first product (+ 1 for the bias step)
constexpr int kOB = kC / 32;       // 32-row blocks of Out^T
constexpr int kW1Frags = kKS + 1;
constexpr int kW2Frags = 2 * kOB;
constexpr int kChunkBytes = (kW1Frags + kW2Frags) * 1024;   // 49 KiB: the LDS image of one chunk of 32 hidden units
constexpr int kStages = 3;   // ring slots per weight (LDS: 3 x 49 KiB = 147 KiB of the CU's 160)

// GELU in as few VALU instructions as the forms allow (they share the SIMD's issue slots with the MFMAs): error far below the bf16
// rounding of the result.
//   tanh form: x sigma(2u), 2u = x (1.59576912 + 0.07135482 x^2); sigma through v_exp_f32 (= 2^x) and v_rcp_f32: 7 instructions
//   erf form:  x Phi(x), Phi = 1 - q (x >= 0) or q (x < 0), q = 0.5 poly(t) exp(-x^2 / 2), t = 1 / (1 + 0.47047 |x| / sqrt 2)
//              (Abramowitz-Stegun 7.1.25: |error of Phi| < 1.3e-5, a 300th of the bf16 rounding of the result): 11 instructions
template <int ACT> __device__ __forceinline__ float gelu_fast(float x) {
    if (ACT == ACT_GELU_TANH) {
        const float t = x * x;
        const float p = __builtin_fmaf(t, -0.10294324f, -2.3022082f);   // -(2u / x) log2(e)
        const float e = __builtin_amdgcn_exp2f(x * p);                   // e^{-2u}
        return x * __builtin_amdgcn_rcpf(1.0f + e);
    }
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x), 0.33267253f, 1.0f));
    const float poly = t * (0.1740121f + t * (-0.0479399f + t * 0.3739278f));
    const float q = poly * __builtin_amdgcn_exp2f(x * x * -0.72134752f);
    return __builtin_fmaf(-fabsf(x), q, fmaxf(x, 0.f));   // x >= 0: x - x q;  x < 0: x q
}

template <int OFF> __device__ __forceinline__ void ds_read128(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
}

struct MlpArgs {
    float* x;              // [M, C] f32, updated in place
    const void* wp;        // packed weights: F / 32 chunk images of kChunkBytes
    const float* b2;       // [C]
    const float* ln_g;     // [C]
    const float* ln_b;     // [C] or null
    int M, nchunk;
    int stagger;           // start delay step in units of 64 shader cycles: block b of the first wave of blocks waits (b % 8) * stagger
    int abl;               // tuning (DIMX_MLP_ABL): 1 = no DMA inside the loop, 2 = no GELU, 4 = no chunk loop at all
};

// ABLC (tuning, DIMX_MLP_ABL bits 8 / 16 on the tanh + beta instantiation): 8 = no fragment reads in the loop, 16 = no MFMAs in the loop
// (The out-projection-in-the-prologue variant of round 4 measured break-even against the separate launch and was removed in
// round 5: profiles/r04_mlp_fused_outproj.txt, git history 572449b.)
template <int ACT, bool BETA, int ABLC = 0> __global__ __launch_bounds__(256) void mlp_fused_kernel(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hf = lane >> 5, l31 = lane & 31;
    const int row = blockIdx.x * 128 + wave * 32 + l31;
    const int rowc = row < a.M ? row : a.M - 1;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    const unsigned char* wp = (const unsigned char*)a.wp;
    // Every CU's first block would run its prologue (196 KB of x per block) at the same moment, and then all the epilogues collide:
    // the first wave of blocks starts staggered so that the memory phases of different CUs fall beside each other's compute.
    if (a.stagger > 0 && blockIdx.x < 256) {
        const int n = (int)(blockIdx.x & 7) * a.stagger;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);   // 64 cycles each
    }

    // The weight stream: per chunk 25 W1 pieces + 24 W2 pieces of 1 KiB, each a linear LDS-DMA (lane l moves bytes [16 l, 16 l + 16)).
    // Wave w issues pieces w, w + 4, ...; W1 and W2 have their own 2-slot rings because the pipeline below reads W1 of chunk c + 2
    // and W2 of chunk c in the same iteration.
    unsigned char* const ring1 = smem;                               // kStages x 25 KiB
    unsigned char* const ring2 = smem + kStages * kW1Frags * 1024;   // kStages x 24 KiB
    const unsigned char* const wlane = wp + lane * 16;
    // piece j (0 .. 12) of this wave's share of {W1(chunk1) -> slot1, W2(chunk2) -> slot2}: 6 + 6 pieces, wave 0 also the 25th of W1
    auto issue_piece = [&](int j, int chunk1, int slot1, int chunk2, int slot2) {
        if (j < 6) {
            const int p = wave + 4 * j;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wlane + (size_t)chunk1 * kChunkBytes + (size_t)p * 1024),
                                             (lds_void_t*)(ring1 + slot1 * (kW1Frags * 1024) + p * 1024), 16, 0, 0);
        } else if (j < 12) {
            const int p = wave + 4 * (j - 6);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wlane + (size_t)chunk2 * kChunkBytes + (size_t)(kW1Frags + p) * 1024),
                                             (lds_void_t*)(ring2 + slot2 * (kW2Frags * 1024) + p * 1024), 16, 0, 0);
        } else if (wave == 0) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wlane + (size_t)chunk1 * kChunkBytes + (size_t)(kW1Frags - 1) * 1024),
                                             (lds_void_t*)(ring1 + slot1 * (kW1Frags * 1024) + (kW1Frags - 1) * 1024), 16, 0, 0);
        }
    };
    const int nch = a.nchunk;
    auto wrap = [&](int c) { return c < nch ? c : (c - nch < nch ? c - nch : 0); };
    // pre-loop: W1(0), W1(1), W1(2) and W2(0), W2(1)
    auto issue_preloop = [&]() {
#pragma unroll
        for (int j = 0; j < 13; ++j) issue_piece(j, 0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 13; ++j) issue_piece(j, wrap(1), 1, wrap(1), 1);
#pragma unroll
        for (int j = 0; j < 13; ++j)
            if (j < 6 || j == 12) issue_piece(j, wrap(2), 2, 0, 0);
    };
    issue_preloop();

    bf16x8_t xb[kKS + 1];
    f32x16_t o[kOB];   // Out^T accumulators: register r of block ob is out column 32 ob + 8 (r / 4) + 4 hf + r % 4
    // ---- this lane's half of its row: k = 16 s + 8 hf + j, read ONCE into registers (192 f32: the accumulators are not live yet).
    // LayerNorm statistics over the row (lanes l and l + 32), two-pass in registers, then the B fragments
    // bf16((x - mean) rstd gamma + beta) replace the f32 values fragment by fragment.
    {
        const float* xr = a.x + (size_t)rowc * kC + 8 * hf;
        f32x4_t v[kKS][2];
#pragma unroll
        for (int s = 0; s < kKS; ++s) {
            v[s][0] = *(const f32x4_t*)(xr + 16 * s);
            v[s][1] = *(const f32x4_t*)(xr + 16 * s + 4);
        }
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < kKS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) s1 += v[s][0][j] + v[s][1][j];
        s1 += __shfl_xor(s1, 32);
        const float mean = s1 * (1.0f / kC);
        float s2 = 0.f;
#pragma unroll
        for (int s = 0; s < kKS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d0 = v[s][0][j] - mean, d1 = v[s][1][j] - mean;
                s2 += d0 * d0 + d1 * d1;
            }
        s2 += __shfl_xor(s2, 32);
        const float rstd = rsqrtf(s2 * (1.0f / kC) + 1e-5f);
        const float* g = a.ln_g + 8 * hf;
        const float* be = a.ln_b + 8 * hf;
#pragma unroll
        for (int s = 0; s < kKS; ++s) {
            const f32x4_t g0 = *(const f32x4_t*)(g + 16 * s), g1 = *(const f32x4_t*)(g + 16 * s + 4);
            f32x4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
            if (BETA) {
                b0 = *(const f32x4_t*)(be + 16 * s);
                b1 = *(const f32x4_t*)(be + 16 * s + 4);
            }
            const f32x4_t v0 = v[s][0], v1 = v[s][1];
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                w[j] = pack_bf16x2((v0[2 * j] - mean) * rstd * g0[2 * j] + b0[2 * j], (v0[2 * j + 1] - mean) * rstd * g0[2 * j + 1] + b0[2 * j + 1]);
                w[2 + j] = pack_bf16x2((v1[2 * j] - mean) * rstd * g1[2 * j] + b1[2 * j], (v1[2 * j + 1] - mean) * rstd * g1[2 * j + 1] + b1[2 * j + 1]);
            }
            const u32x4_t u = {w[0], w[1], w[2], w[3]};
            xb[s] = __builtin_bit_cast(bf16x8_t, u);
        }
        // the bias step: k-slots 0 and 1 of the 25th k-step multiply bf16 hi / lo of b1 (packed into W1) by 1
        const u32x4_t one = {hf == 0 ? 0x3f803f80u : 0u, 0u, 0u, 0u};
        xb[kKS] = __builtin_bit_cast(bf16x8_t, one);
        __builtin_amdgcn_sched_barrier(0);   // the accumulators come to life only now

        // ---- Out^T accumulators, started at b2
#pragma unroll
        for (int ob = 0; ob < kOB; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4_t b = *(const f32x4_t*)(a.b2 + 32 * ob + 8 * q + 4 * hf);
#pragma unroll
            for (int t = 0; t < 4; ++t) o[ob][4 * q + t] = b[t];
        }

    }

    // ---- the pipeline.  Iteration c carries three independent instruction streams, so the GELU's VALU work and the fragment reads ride
    // in the issue slots between MFMAs instead of standing alone:
    //     MFMA   h_next  = W1(c + 2) . LN(x)^T            (25, one dependent chain ...
    //     MFMA   Out^T  += W2(c) . hB_cur                 ... interleaved with these 24 on 12 independent accumulators)
    //     VALU   hB_next = bf16(gelu(h_mid)),  h_mid = H^T of chunk c + 1 (finished in iteration c - 1)
    // Chunk indices past the end wrap around: the last two iterations compute two H^T tiles nobody reads (2 % of the MFMAs) and the
    // loop body has no branch.
    const unsigned rd1 = lds0 + lane * 16, rd2 = rd1 + kStages * kW1Frags * 1024;
    auto gemm1 = [&](int slot, f32x16_t& h) {   // pre-loop form: the whole first product of one chunk
        const unsigned base = rd1 + slot * (kW1Frags * 1024);
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = 0.f;
        u32x4_t fr[2][5];
#pragma unroll
        for (int q = 0; q < 5; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[0][q]) : "v"(base), "i"(q * 1024));
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            if (g + 1 < 5) {
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(g + 1) & 1][q]) : "v"(base), "i"(((g + 1) * 5 + q) * 1024));
                asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 5; ++q)
                h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[g & 1][q]), xb[g * 5 + q], h, 0, 0, 0);
        }
    };
    auto gelu_pair = [&](const f32x16_t& h, int r) { return pack_bf16x2(gelu_fast<ACT>(h[2 * r]), gelu_fast<ACT>(h[2 * r + 1])); };

    f32x16_t h_mid, h_next;
    uint32_t hbc[8], hbn[8];    // hB_cur / hB_next: two B fragments each
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gemm1(0, h_next);
#pragma unroll
    for (int r = 0; r < 8; ++r) hbc[r] = gelu_pair(h_next, r);
    gemm1(1, h_mid);

    // Ring discipline (3 slots per weight): iteration c reads W1(c + 2) from slot (c + 2) % 3 and W2(c) from slot c % 3 and issues
    // the DMA of W1(c + 3) -> slot c % 3 and W2(c + 2) -> slot (c + 2) % 3, one piece per MFMA group.  Both target slots were last
    // read in iteration c - 1 (W1(c), pre-loop for c = 0 .. 1; W2(c - 1)), which every wave has left once it passed this iteration's
    // barrier.  What iteration c reads was issued in the FIRST half of iteration c - 1 (W1: groups 0 .. 5) and in iteration c - 2 (W2:
    // groups 6 .. 11), so the wait at the top leaves the 6 youngest pieces in flight.
    const int loops = (a.abl & 4) ? 0 : nch;
    int s0 = 0, s2 = 2;    // c % 3, (c + 2) % 3
    for (int c = 0; c < loops; ++c) {
        // in-order return: all but this wave's 6 youngest pieces -- the W2(c + 1) pieces issued in the second half of iteration
        // c - 1, not read before iteration c + 1 -- have landed
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int cw1 = wrap(c + 3), cw2 = wrap(c + 2);
        const bool dma = !(a.abl & 1);
        const unsigned b1 = rd1 + s2 * (kW1Frags * 1024);   // W1(c + 2)
        const unsigned b2 = rd2 + s0 * (kW2Frags * 1024);   // W2(c)
#pragma unroll
        for (int r = 0; r < 16; ++r) h_next[r] = 0.f;
        // 12 groups: {2 k-steps of the first product, one out block of the second}; fragments are read one group ahead.  Group 0
        // also takes the bias k-step.  GELU pairs 0..7 ride in groups 2..9, one DMA piece per group.
        u32x4_t fr[3][4], fbias;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbias) : "v"(b1), "i"(kKS * 1024));
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[gg][q]) : "v"(b1), "i"((2 * gg + q) * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[gg][2 + q]) : "v"(b2), "i"((2 * gg + q) * 1024));
            }
        if (dma) issue_piece(12, cw1, s0, cw2, s2);
#pragma unroll
        for (int g = 0; g < kOB; ++g) {
            // fragments travel two groups ahead of their MFMAs (LDS returns in order)
            if (g + 2 < kOB) {
                if (!(ABLC & 8)) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(g + 2) % 3][q]) : "v"(b1), "i"((2 * (g + 2) + q) * 1024));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(g + 2) % 3][2 + q]) : "v"(b2), "i"((2 * (g + 2) + q) * 1024));
                    }
                    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                }
            } else if (g + 1 < kOB) {
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4_t c0 = {hbc[0], hbc[1], hbc[2], hbc[3]}, c1v = {hbc[4], hbc[5], hbc[6], hbc[7]};
            if (!(ABLC & 16)) {
                if (g == 0) h_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fbias), xb[kKS], h_next, 0, 0, 0);
                h_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[g % 3][0]), xb[2 * g], h_next, 0, 0, 0);
                h_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[g % 3][1]), xb[2 * g + 1], h_next, 0, 0, 0);
            } else {
                h_next[g] += __builtin_bit_cast(float, fr[g % 3][0][0]) + __builtin_bit_cast(float, fr[g % 3][1][1]);
            }
            if (dma) issue_piece(g, cw1, s0, cw2, s2);
            if (!(ABLC & 16)) {
                o[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[g % 3][2]), __builtin_bit_cast(bf16x8_t, c0), o[g], 0, 0, 0);
                o[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[g % 3][3]), __builtin_bit_cast(bf16x8_t, c1v), o[g], 0, 0, 0);
            } else {
                o[g][0] += __builtin_bit_cast(float, fr[g % 3][2][0]) + __builtin_bit_cast(float, fr[g % 3][3][1]) + __builtin_bit_cast(float, hbc[g & 7]);
            }
            if (g >= 2 && g < 10) hbn[g - 2] = (a.abl & 2) ? hbc[g - 2] : gelu_pair(h_mid, g - 2);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) hbc[r] = hbn[r];
        h_mid = h_next;
        s0 = s0 == 2 ? 0 : s0 + 1;
        s2 = s2 == 2 ? 0 : s2 + 1;
    }

    // ---- epilogue: x[row, col] += Out^T[col, row]
    if (row < a.M) {
        float* xr = a.x + (size_t)row * kC + 4 * hf;
#pragma unroll
        for (int ob = 0; ob < kOB; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* p = xr + 32 * ob + 8 * q;
                f32x4_t v = {0.f, 0.f, 0.f, 0.f};
                v = *(const f32x4_t*)p;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] += o[ob][4 * q + t];
                *(f32x4_t*)p = v;
            }
    }
}

}  // namespace

size_t mlp_fused_packed_bytes(int C, int F) { return (C == kC && F % 32 == 0) ? (size_t)(F / 32) * kChunkBytes : 0; }

// host: the chunk images.  w1 [F, C], b1 [F], w2 [C, F] f32 row-major -> out (mlp_fused_packed_bytes)
int mlp_fused_pack(const float* w1, const float* b1, const float* w2, int C, int F, uint16_t* out) {
    DIMX_REQUIRE(C == kC && F % 32 == 0, DIMX_ERR_ARG, "mlp_fused_pack: C = %d F = %d not supported", C, F);
    auto bf = [](uint16_t b) {
        const uint32_t u = (uint32_t)b << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    for (int c = 0; c < F / 32; ++c) {
        uint16_t* img = out + (size_t)c * (kChunkBytes / 2);
        for (int s = 0; s < kW1Frags; ++s)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int h = 32 * c + (l & 31), half = l >> 5;
                    uint16_t v = 0;
                    if (s < kKS) {
                        v = host_f32_to_bf16(w1[(size_t)h * C + 16 * s + 8 * half + j]);
                    } else if (half == 0 && j < 2) {   // b1 as hi + lo
                        const uint16_t hi = host_f32_to_bf16(b1 ? b1[h] : 0.f);
                        v = j == 0 ? hi : host_f32_to_bf16((b1 ? b1[h] : 0.f) - bf(hi));
                    }
                    img[(size_t)s * 512 + l * 8 + j] = v;
                }
        for (int ob = 0; ob < kOB; ++ob)
            for (int s2 = 0; s2 < 2; ++s2)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int o = 32 * ob + (l & 31), half = l >> 5;
                        const int hid = 32 * c + (2 * s2 + j / 4) * 8 + 4 * half + j % 4;   // accumulator-row order of H^T
                        img[(size_t)(kW1Frags + 2 * ob + s2) * 512 + l * 8 + j] = host_f32_to_bf16(w2[(size_t)o * F + hid]);
                    }
    }
    return DIMX_OK;
}

// act: ACT_GELU_TANH / ACT_GELU_ERF
int launch_mlp_fused(float* x, const void* packed, const float* b2, const float* ln_g, const float* ln_b, int M, int C, int F, int act,
                     hipStream_t s) {
    DIMX_REQUIRE(x && packed && b2 && ln_g && M > 0, DIMX_ERR_ARG, "mlp_fused: null argument");
    DIMX_REQUIRE(C == kC && F % 32 == 0 && (act == ACT_GELU_TANH || act == ACT_GELU_ERF), DIMX_ERR_ARG, "mlp_fused: C = %d F = %d act = %d", C, F, act);
    DIMX_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)b2 % 16) == 0 && ((uintptr_t)ln_g % 16) == 0 &&
                     (!ln_b || ((uintptr_t)ln_b % 16) == 0),
                 DIMX_ERR_ARG, "mlp_fused: operands must be 16-byte aligned");
    MlpArgs a;
    a.x = x; a.wp = packed; a.b2 = b2; a.ln_g = ln_g; a.ln_b = ln_b; a.M = M; a.nchunk = F / 32;
    static const int abl = getenv("DIMX_MLP_ABL") ? atoi(getenv("DIMX_MLP_ABL")) : 0;
    a.abl = abl;
    static const int stagger = getenv("DIMX_MLP_STAGGER") ? atoi(getenv("DIMX_MLP_STAGGER")) : 0;
    a.stagger = stagger;
    DIMX_REQUIRE(a.nchunk >= 3, DIMX_ERR_ARG, "mlp_fused: F = %d is too small for the three-deep pipeline", F);
    const size_t lds = (size_t)kStages * kChunkBytes;
    const dim3 grid((M + 127) / 128);
    if (abl & 24) {
        DIMX_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<ACT_GELU_TANH, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DIMX_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<ACT_GELU_TANH, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
#define MLP_LAUNCH(A, B)                                                                                                              \
    do {                                                                                                                              \
        /* per launch, not once per process: the attribute is per device and the call is cheap */                     \
        DIMX_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((mlp_fused_kernel<A, B>), grid, dim3(256), lds, s, a);                                                     \
    } while (0)
    if (act == ACT_GELU_TANH) {
        if (ln_b && (abl & 8)) hipLaunchKernelGGL((mlp_fused_kernel<ACT_GELU_TANH, true, 8>), grid, dim3(256), lds, s, a);
        else if (ln_b && (abl & 16)) hipLaunchKernelGGL((mlp_fused_kernel<ACT_GELU_TANH, true, 16>), grid, dim3(256), lds, s, a);
        else if (ln_b) MLP_LAUNCH(ACT_GELU_TANH, true); else MLP_LAUNCH(ACT_GELU_TANH, false);
    } else {
        if (ln_b) MLP_LAUNCH(ACT_GELU_ERF, true); else MLP_LAUNCH(ACT_GELU_ERF, false);
    }
#undef MLP_LAUNCH
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
