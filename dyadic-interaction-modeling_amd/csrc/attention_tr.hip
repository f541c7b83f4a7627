// attention_tr.hip -- the bf16 prefill attention of round 4: softmax(Q K^T * scale + mask) V for the parallel
// (non-autoregressive) attention blocks with 48- or 64-wide heads, ALL THREE operands row-major ([.., key, d]: what the
// fused q/k/v projection of the two-phase 256 x 256 GEMM stores, and what the head-major cross-attention caches are).
// Same mathematics and masking rules as attention.hip (reference code/models/lib/base_models.py:125-146 for the VQ-VAE
// blocks; x-transformers' Attend via code/seq2seq_pretrain.py:388-418,439-448 for the encoder / decoder stacks); the
// f32 parity mode, the 96-wide heads of the legacy speaker VQ-VAE and transposed-V callers stay on attention.hip.
//
// What changed against attn_kernel<bf16, 2, 64, VROW> (205 us per call at B 256, H 12, L 300; 220 VGPRs -> 2 waves per SIMD):
//   * NW (5 at L = 300: 2 x 160 queries) waves of 32 queries share every staged K / V tile instead of 2, and the blocks of
//     one (clip, head) are placed on ONE XCD (block id mod 8 is the XCD), so the second block's K / V come from that L2;
//   * V is staged row-major exactly like K (16-byte loads / ds_write_b128 -- the 16 predicated 4-byte loads + v_perm per
//     thread and tile are gone) and read as the A operand of O^T = V^T . P^T with gfx950's transposing LDS read
//     (ds_read_b64_tr_b16: lane i of a 16-lane group receives element i & 3 of the 8-byte pieces addressed by lanes
//     4 j + (i >> 2), j = 0..3 -- column i of a [4 keys][16 d] block);
//   * two LDS buffers, ONE barrier per tile: tile t + 1 is written (from the registers its loads were issued into during
//     tile t - 1) while tile t is multiplied, tile t + 2's loads are issued right behind the write;
//   * the 64^-0.5 log2(e) scale is folded into the exponent's fma, D = 48 contracts over 3 k-steps instead of a padded 4,
//     waves above the causal diagonal skip the tile, the key mask byte of the next tile is prefetched, the output leaves
//     as 16-byte stores (v_permlane32_swap pairs the two lane halves' 8-byte pieces).
#include "common.hpp"

namespace dimx {

namespace {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // native vectors: a HIP uint4 (a struct) copied through an
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));   // address-space cast kept the staging array in scratch
constexpr float kNegT = -0x1p126f;  // a power of two: kNegT * scale2 is exact, so a fully masked row sees exp2(0) = 1 like attention.hip

__device__ __forceinline__ u32x2_t lds_read_tr16(uint32_t addr) {
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)addr);
    return __builtin_bit_cast(u32x2_t, v);
}
__device__ __forceinline__ u32x4_t lds_read_b128(uint32_t addr) {
    return *(__attribute__((address_space(3))) const u32x4_t*)(uintptr_t)addr;
}
__device__ __forceinline__ void lds_write_b128(uint32_t addr, u32x4_t v) {
    *(__attribute__((address_space(3))) u32x4_t*)(uintptr_t)addr = v;
}

// DH: head width (48 / 64).  NW waves x 32 queries per block.
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64) void attn_tr_kernel(const AttnArgs a, const int nqb) {
    constexpr int NT = NW * 64;
    constexpr int CR = DH / 8;                     // 16-byte chunks per K / V row
    constexpr int NKS = DH / 16;                   // k-steps of S^T = K . Q^T
    constexpr int UNITS = 2 * CR;                  // 64-chunk units per tile: CR of K, then CR of V
    constexpr int NI = (UNITS + NW - 1) / NW;      // staging chunks per thread and tile
    constexpr int BUF = 16384, VOFF = 8192;        // per buffer: K [64][128 B] | V [64][128 B]
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // blocks n, n + 8, .. of one XCD walk (clip, head) pairs; the nqb query blocks of a pair are consecutive slots of that XCD
    const int n = blockIdx.x, xcd = n & 7, slot = n >> 3;
    const int bh = (slot / nqb) * 8 + xcd;
    if (bh >= a.B * a.H) return;
    const int qb = nqb - 1 - slot % nqb;           // the block with the most causal tiles first
    const int b = bh / a.H, h = bh - b * a.H;
    const int qblk0 = qb * (32 * NW);
    const int qlo = qblk0 + wave * 32;
    const int qi = qlo + l31;
    const int qc = qi < a.Lq ? qi : a.Lq - 1;
    const bool wave_on = qlo < a.Lq;

    const bf16* __restrict__ Q = (const bf16*)a.q + (size_t)b * a.q_sb + (size_t)h * a.q_sh;
    const bf16* __restrict__ K = (const bf16*)a.k + (size_t)b * a.k_sb + (size_t)h * a.k_sh;
    const bf16* __restrict__ V = (const bf16*)a.vt + (size_t)b * a.v_sb + (size_t)h * a.v_sh;

    u32x4_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const u32x4_t*)(Q + (size_t)qc * a.q_st + (2 * ks + half) * 8);

    int kmax = a.Lk;
    const int len_b = a.lens ? a.lens[b] : a.Lk;
    kmax = len_b < kmax ? len_b : kmax;
    if (a.causal) {
        int last = qblk0 + 32 * NW - 1;
        last = last < a.Lq - 1 ? last : a.Lq - 1;
        kmax = (last + 1) < kmax ? (last + 1) : kmax;
    }
    const int ntiles = (kmax + 63) / 64;
    const int qlast = (qlo + 31) < a.Lq ? (qlo + 31) : (a.Lq - 1);   // this wave's last query (causal tile skip)

    // ---- staging map: thread -> NI 16-byte chunks (unit = wave + i * NW; units 0 .. CR-1 are K, CR .. 2 CR-1 are V).  A slot
    // beyond the last unit repeats the last unit (same bytes to the same LDS address as the wave that owns it): the staging
    // code stays branch-free, so the staging registers stay registers (conditional slots sent the array to scratch).
    uint32_t goff[NI], loff[NI], gstr[NI];
    int srow[NI];
    const char* gbase[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int unit = wave + i * NW;
        unit = unit < UNITS ? unit : UNITS - 1;
        const bool isv = unit >= CR;
        const int rem = (unit - (isv ? CR : 0)) * 64 + lane;
        const int row = rem / CR, c = rem - row * CR;
        srow[i] = row;
        gstr[i] = (uint32_t)(isv ? a.v_st : a.k_st) * 2u;                 // wave-uniform
        gbase[i] = (const char*)(isv ? V : K);                           // wave-uniform
        goff[i] = (uint32_t)row * gstr[i] + (uint32_t)c * 16u;
        const int cs = isv ? (c ^ (((row >> 1) & 1) << 2)) : (c ^ ((row >> 1) & 7));
        loff[i] = lds0 + (isv ? VOFF : 0) + row * 128 + (cs << 4);
    }
    u32x4_t sreg[NI];
    // rows beyond Lk repeat the last row (finite values; their keys are masked)
#define DIMX_LOAD_TILE(J0)                                                                          \
    do {                                                                                            \
        const int j0_ = (J0);                                                                       \
        const int lim_ = a.Lk - 1 - j0_;                                                            \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                            \
            uint32_t off_ = goff[i];                                                                \
            if (lim_ < 63) off_ -= (uint32_t)(srow[i] > lim_ ? srow[i] - lim_ : 0) * gstr[i];       \
            sreg[i] = *(const u32x4_t*)(gbase[i] + (size_t)j0_ * gstr[i] + off_);                     \
        }                                                                                           \
    } while (0)
#define DIMX_WRITE_TILE(BUFOFF)                                                                     \
    do {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) lds_write_b128(loff[i] + (BUFOFF), sreg[i]); \
    } while (0)
    // key validity of tile j0 for this lane's key j0 + lane: the mask byte is only LOADED here (consumed a tile later, so
    // the wave never waits for it), the predicate is formed by key_valid() and combined by __ballot
    const uint8_t* kmrow = a.kmask ? a.kmask + (size_t)b * a.kmask_ld : nullptr;
#define DIMX_KM_LOAD(J0) ((kmrow && (J0) + lane < a.Lk) ? (uint32_t)kmrow[(J0) + lane] : 1u)
#define DIMX_KEY_VALID(J0, BYTE) ((J0) + lane < a.Lk && (J0) + lane < len_b && (BYTE) != 0u)

    // ---- per-lane LDS read addresses (buffer 0; the buffer toggles with ^ BUF)
    // K as A operand: row 32 kt + l31 (kt by immediate 4096), chunk 2 ks + half: (c ^ swz) = ((ks ^ sw >> 1) << 1) | (half ^ sw & 1)
    const int swk = (l31 >> 1) & 7;
    uint32_t kaddr = lds0 + l31 * 128 + (((half ^ (swk & 1)) | ((swk >> 1) << 1)) << 4);   // ks = 0; ks by ^ (ks << 5)
    // V through the transposing read: 16-lane group g16 covers d = 32 blk + 16 g16 .. + 15; lane p of it addresses key
    // 4 half + (p >> 2) (+ 16 s + 32 kt + 8 second, by immediate), 8 bytes at d = 32 blk + 16 g16 + 4 (p & 3)
    const int p16 = lane & 15, g16 = (lane >> 4) & 1;
    const int vrow = 4 * half + (p16 >> 2);
    const int vsw = ((p16 >> 3) & 1) << 6;          // = ((row >> 1) & 1) << 6 for every row this lane addresses
    uint32_t vaddr0 = lds0 + VOFF + vrow * 128 + ((32 * g16 + 8 * (p16 & 3)) ^ vsw);
    uint32_t vaddr1 = lds0 + VOFF + vrow * 128 + ((64 + 32 * g16 + 8 * (p16 & 3)) ^ vsw);

    float m_run = kNegT, l_run = 0.f;
    f32x16_t ot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    const float scale2 = a.scale * 1.4426950408889634f;

    // ---- prologue: tile 0 into buffer 0, tile 1's loads in flight.  An empty clip (no key at all) writes zeros like attention.hip.
    if (ntiles == 0) {
        if (qi < a.Lq) {
            bf16* orow0 = (bf16*)a.o + (size_t)b * a.o_sb + (size_t)qi * a.o_st + (size_t)h * a.o_sh;
#pragma unroll
            for (int c = 0; c < CR; ++c) *(u32x4_t*)(orow0 + 8 * c) = u32x4_t{0u, 0u, 0u, 0u};
        }
        return;
    }
    DIMX_LOAD_TILE(0);
    uint32_t km_byte = DIMX_KM_LOAD(0);
    DIMX_WRITE_TILE(0u);
    if (ntiles > 1) DIMX_LOAD_TILE(64);
    __syncthreads();

    // the next tile's operands go to the other buffer (their loads were issued a whole tile ago), then tile t + 2's loads leave
#define DIMX_STAGE_NEXT()                                          \
    do {                                                           \
        if (tile + 1 < ntiles) {                                   \
            DIMX_WRITE_TILE(cur ^ BUF);                            \
            km_byte = DIMX_KM_LOAD(j0 + 64);                       \
            if (tile + 2 < ntiles) DIMX_LOAD_TILE(j0 + 128);       \
        }                                                          \
    } while (0)

    for (int tile = 0; tile < ntiles; ++tile) {
        const int j0 = tile * 64;
        const uint32_t cur = (tile & 1) ? BUF : 0;
        const unsigned long long kbits = __ballot(DIMX_KEY_VALID(j0, km_byte));
        const bool on = wave_on && (!a.causal || j0 <= qlast);
        // the next tile's operands go to the other buffer first (their loads were issued a whole tile ago; every wave left
        // that buffer at the last barrier), then tile t + 2's loads leave -- unconditional, ahead of the wave's own matrix
        // work: staged in one place, the staging registers are never copied between branches
        DIMX_STAGE_NEXT();
        if (on) {
            // ---- S^T = K . Q^T: all fragment reads first, one wait, then the eight matrix instructions back to back
            u32x4_t kf[2][NKS];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) kf[kt][ks] = lds_read_b128(((kaddr ^ (ks << 5)) ^ cur) + kt * 4096);
            __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise re-serialises read -> wait -> MFMA to save registers
            f32x16_t st[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf[kt][ks]),
                                                                     __builtin_bit_cast(bf16x8_t, qf[ks]), st[kt], 0, 0, 0);
            }
            // ---- mask + online softmax, lane-local (this lane's query, 32 of the tile's 64 keys); raw scores, the scale rides
            // in the exponent's fma
            const bool interior = kbits == ~0ull && (!a.causal || j0 + 63 <= qlo);
            float mx = kNegT;
            if (!interior) {
                // one validity word per lane, bit 16 kt + r <-> key 32 kt + (r & 3) + 8 (r >> 2) + 4 half: the tile's key bits
                // shifted by this lane's half, every second nibble kept and packed; causal: keys j0 + x <= qi
                const uint32_t klo = half ? (uint32_t)(kbits >> 4) : (uint32_t)kbits;
                const uint32_t khi = half ? (uint32_t)(kbits >> 36) : (uint32_t)(kbits >> 32);
                uint32_t y0 = klo & 0x0F0F0F0Fu, y1 = khi & 0x0F0F0F0Fu;
                y0 = (y0 | (y0 >> 4)) & 0x00FF00FFu;
                y1 = (y1 | (y1 >> 4)) & 0x00FF00FFu;
                y0 = (y0 | (y0 >> 8)) & 0xFFFFu;
                y1 = (y1 | (y1 >> 8)) & 0xFFFFu;
                uint32_t vm = y0 | (y1 << 16);
                if (a.causal) {
                    const int nvis = qi - j0 - 4 * half + 1;              // visible key offsets x (of 0..63) are x < nvis
                    int rem = nvis & 7;
                    rem = rem < 4 ? rem : 4;
                    int nb = 4 * (nvis >> 3) + rem;
                    nb = nvis <= 0 ? 0 : nb;
                    vm &= nb >= 32 ? ~0u : ((1u << nb) - 1u);
                }
                const uint32_t negb = __builtin_bit_cast(uint32_t, kNegT);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t keep = (uint32_t)((int)(vm << (31 - (16 * kt + r))) >> 31);   // all ones: valid
                        const uint32_t sb = __builtin_bit_cast(uint32_t, st[kt][r]);
                        st[kt][r] = __builtin_bit_cast(float, (sb & keep) | (negb & ~keep));
                    }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
            mx = fmaxf(mx, xor_lane_f32<32>(mx)) * scale2;
            const float m_new = fmaxf(m_run, mx);
            // the running maximum (and with it the 32 accumulator rescales) only moves when some query's maximum grew by more
            // than 2^8: probabilities relative to a slightly stale maximum are at most 256, the final normalisation divides it out
            if (__any((m_new - m_run) > 8.0f)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
            }
            const float nm = -m_run;
            float psum = 0.f;
            u32x4_t pk[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float p[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][8 * s + e], scale2, nm));
                        psum += p[e];
                    }
                    pk[kt][s] = u32x4_t{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]),
                                        pack_bf16x2(p[6], p[7])};
                }
            l_run += psum;

            // ---- O^T += V^T . P^T (one output block's eight transposing reads, then its four matrix instructions)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const uint32_t vb = (blk ? vaddr1 : vaddr0) ^ cur;
                u32x4_t va[2][2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const u32x2_t v0 = lds_read_tr16(vb + (32 * kt + 16 * s) * 128);
                        const u32x2_t v1 = lds_read_tr16(vb + (32 * kt + 16 * s + 8) * 128);
                        va[kt][s] = u32x4_t{v0.x, v0.y, v1.x, v1.y};
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        ot[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, va[kt][s]),
                                                                          __builtin_bit_cast(bf16x8_t, pk[kt][s]), ot[blk], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef DIMX_STAGE_NEXT

    // ---- finish: combine the two lane halves' row sums, normalise, store O[q][d] as 16-byte pieces
    const float l_tot = l_run + xor_lane_f32<32>(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    bf16* orow = (bf16*)a.o + (size_t)b * a.o_sb + (size_t)qc * a.o_st + (size_t)h * a.o_sh;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            if (32 * blk + 16 * gp < DH) {
                // groups g = 2 gp (d0 = 32 blk + 16 gp + 4 half) and g + 1 (d0 + 8): after the swap the lower half holds
                // d 32 blk + 16 gp + 0..7, the upper half d .. + 8..15
                const int g = 2 * gp;
                uint32_t x0 = pack_bf16x2(ot[blk][4 * g] * inv, ot[blk][4 * g + 1] * inv);
                uint32_t x1 = pack_bf16x2(ot[blk][4 * g + 2] * inv, ot[blk][4 * g + 3] * inv);
                uint32_t y0 = pack_bf16x2(ot[blk][4 * g + 4] * inv, ot[blk][4 * g + 5] * inv);
                uint32_t y1 = pack_bf16x2(ot[blk][4 * g + 6] * inv, ot[blk][4 * g + 7] * inv);
                const auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                if (qi < a.Lq) *(u32x4_t*)(orow + 32 * blk + 16 * gp + 8 * half) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
            }
        }
}

#undef DIMX_LOAD_TILE
#undef DIMX_WRITE_TILE
#undef DIMX_KM_LOAD
#undef DIMX_KEY_VALID

}  // namespace

// bf16, D in {48, 64}, q / k / v row-major with 16-byte aligned rows.  Returns DIMX_OK after launching.
int launch_attention_tr(const AttnArgs& a, hipStream_t s) {
    DIMX_REQUIRE(a.dtype == DIMX_BF16 && (a.D == 48 || a.D == 64) && a.v_rows, DIMX_ERR_ARG, "attention_tr: bf16, D 48 / 64, row-major V");
    DIMX_REQUIRE(a.q_st % 8 == 0 && a.k_st % 8 == 0 && a.v_st % 8 == 0 && a.o_st % 8 == 0 && a.q_sh % 8 == 0 && a.k_sh % 8 == 0 &&
                     a.v_sh % 8 == 0 && a.o_sh % 8 == 0 && a.q_sb % 8 == 0 && a.k_sb % 8 == 0 && a.v_sb % 8 == 0 && a.o_sb % 8 == 0,
                 DIMX_ERR_ARG, "attention_tr: strides must keep 16-byte alignment");
    DIMX_REQUIRE((long)64 * (a.k_st > a.v_st ? a.k_st : a.v_st) * 2 < (1l << 31), DIMX_ERR_ARG, "attention_tr: row stride too large");
    const int nwaves = ceil_div(a.Lq, 32);
    // waves per block: the fewest blocks of at most 8 waves, evenly filled (L = 300: 10 waves -> 2 x 5; L = 1500: 47 -> 6 x 8)
    const int nblk = ceil_div(nwaves, 8);
    const int nw = ceil_div(nwaves, nblk);
    const int NWsel = nw <= 1 ? 1 : nw <= 2 ? 2 : nw <= 4 ? 4 : nw <= 5 ? 5 : 8;
    const int nqb = ceil_div(a.Lq, 32 * NWsel);
    const int pairs8 = ceil_div(a.B * a.H, 8);
    dim3 grid(pairs8 * 8 * nqb), block(NWsel * 64);
#define DIMX_ATR(DH, NWV) hipLaunchKernelGGL((attn_tr_kernel<DH, NWV>), grid, block, 0, s, a, nqb)
#define DIMX_ATR_D(NWV)      \
    do {                     \
        if (a.D == 48)       \
            DIMX_ATR(48, NWV); \
        else                 \
            DIMX_ATR(64, NWV); \
    } while (0)
    switch (NWsel) {
        case 1: DIMX_ATR_D(1); break;
        case 2: DIMX_ATR_D(2); break;
        case 4: DIMX_ATR_D(4); break;
        case 5: DIMX_ATR_D(5); break;
        default: DIMX_ATR_D(8); break;
    }
#undef DIMX_ATR_D
#undef DIMX_ATR
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
