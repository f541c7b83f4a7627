// attention_tr.hip -- the bf16 prefill attention of round 4: softmax(Q K^T * scale + mask) V for the parallel
// (non-autoregressive) attention blocks with 48- or 64-wide heads, ALL THREE operands row-major ([.., key, d]: what the
// fused q/k/v projection of the two-phase 256 x 256 GEMM stores, and what the head-major cross-attention caches are).
// Same mathematics and masking rules as attention.hip (reference code/models/lib/base_models.py:125-146 for the VQ-VAE
// blocks; x-transformers' Attend via code/seq2seq_pretrain.py:388-418,439-448 for the encoder / decoder stacks); the
// f32 parity mode, the 96-wide heads of the legacy speaker VQ-VAE and transposed-V callers stay on attention.hip.
//
// Round 3's attn_kernel<bf16, 2, 64, VROW> took 205 us per call at B 256, H 12, L 300 (14 % of the bf16 MFMA peak): 2-wave
// blocks at 2 waves per SIMD, 16 predicated 4-byte loads + v_perm per thread and tile to transpose V.  Measured on the way here
// (profiles/r04_attn_*): a block that only loads Q and two tiles and exits costs 92 us at 6144 blocks -- the price is the
// per-block launch + first-byte latency, not the arithmetic -- and a tile step whose next tile is requested one step ahead
// waits 1.5 - 2.3 us for it whatever the step computes.  Hence:
//   * PERSISTENT blocks (2 per CU, 4 waves x 32 queries = one 128-query item at a time) walk the (clip, head) pairs; nothing
//     is ever requested and then waited for in the same breath: K / V tiles travel global -> LDS by LDS-DMA (global_load_lds,
//     16 B per lane, the XOR swizzle applied to the SOURCE chunk) into a FOUR-slot ring, three tiles ahead of the one being
//     multiplied, across item boundaries; the next item's Q rows land in the wave's own LDS rows while the current item runs;
//     the end-of-step wait is a COUNTED vmcnt (only the next tile must have landed; the two younger tiles, the next item's Q rows
//     and the last item's output stores stay in flight); key masks arrive as 64-bit validity words through the scalar cache
//     (a pack kernel ahead of the launch), clip lengths an item ahead;
//   * every LDS read is inline asm: behind a C++ LDS access hipcc waits vmcnt(0) -- it cannot tell the read from the DMA in
//     flight -- which would park every wave until the youngest tile has landed;
//   * V is staged row-major exactly like K and read as the A operand of O^T = V^T . P^T with gfx950's transposing LDS read
//     (ds_read_b64_tr_b16: lane i of a 16-lane group receives element i & 3 of the 8-byte pieces addressed by lanes
//     4 j + (i >> 2), j = 0..3 -- column i of a [4 keys][16 d] block; checked by tools/ubench/tr16_probe.hip);
//   * the 64^-0.5 log2(e) scale is folded into the exponent's fma, D = 48 contracts over 3 k-steps instead of a padded 4,
//     waves above the causal diagonal skip the tile, masked tiles test one validity word per lane, the output leaves as
//     16-byte stores (v_permlane32_swap pairs the two lane halves' 8-byte pieces).
#include <stdlib.h>

#include "common.hpp"

namespace dimx {

namespace {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // native vectors (a HIP uint4 is a struct)
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
constexpr float kNegT = -0x1p126f;  // a power of two: kNegT * scale2 is exact, so a fully masked row sees exp2(0) = 1 like attention.hip

// Fragment reads, results waited for inside the statement (see the header).
__device__ __forceinline__ void lds_read_4x2_b128(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, u32x4_t (&lo)[4], u32x4_t (&hi)[4]) {
    asm volatile(
        "ds_read_b128 %0, %8\n\t"
        "ds_read_b128 %1, %9\n\t"
        "ds_read_b128 %2, %10\n\t"
        "ds_read_b128 %3, %11\n\t"
        "ds_read_b128 %4, %8 offset:4096\n\t"
        "ds_read_b128 %5, %9 offset:4096\n\t"
        "ds_read_b128 %6, %10 offset:4096\n\t"
        "ds_read_b128 %7, %11 offset:4096\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
        : "memory");
}
__device__ __forceinline__ void lds_read_4_b128(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, u32x4_t (&o)[4]) {
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %5\n\t"
        "ds_read_b128 %2, %6\n\t"
        "ds_read_b128 %3, %7\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
        : "memory");
}
// eight transposing reads: key rows 0, 8, 16, .. 56 (+ the lane's own row) of one 32-column block
__device__ __forceinline__ void lds_read_8_tr16(uint32_t addr, u32x2_t (&t)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:2048\n\t"
        "ds_read_b64_tr_b16 %3, %8 offset:3072\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:4096\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:5120\n\t"
        "ds_read_b64_tr_b16 %6, %8 offset:6144\n\t"
        "ds_read_b64_tr_b16 %7, %8 offset:7168\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
        : "v"(addr)
        : "memory");
}
// wait until at most n (rounded down to a multiple of 4) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void wait_vm_at_most(int n) {
    if (n >= 63)
        return;
    else if (n >= 16)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// key validity words for attention_tr: word[b][t] bit j <-> key 64 t + j of clip b is a key (below Lk, below the clip's
// length, kept by the mask).  One thread per key; a wave writes one word.
__global__ __launch_bounds__(256) void pack_key_words_kernel(const uint8_t* __restrict__ kmask, int kmask_ld, const int32_t* __restrict__ lens,
                                                             int B, int Lk, int nwords, unsigned long long* __restrict__ words) {
    const int gw = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (gw >= B * nwords) return;
    const int b = gw / nwords, t = gw - b * nwords;
    const int j = 64 * t + lane;
    bool ok = j < Lk && (!lens || j < lens[b]);
    if (ok && kmask) ok = kmask[(size_t)b * kmask_ld + j] != 0;
    const unsigned long long w = __ballot(ok);
    if (lane == 0) words[gw] = w;
}

// One work item = 128 queries (4 waves x 32) of one (clip, head) pair.
struct AttnItem {
    int pair, b, qblk0, kmax, ntiles, len_b;
    const bf16 *Q, *K, *V;
    bf16* O;
};


// DH: head width (48 / 64).  4 waves x 32 queries per item, persistent blocks (2 per CU).
// kRing tile slots: one being multiplied, kRing - 1 landing; 160 KiB / (16 KiB kRing + 16 KiB) blocks per CU.
template <int DH, int kRing>
__global__ __launch_bounds__(256, kRing == 2 ? 3 : 2) void attn_tr_kernel(const AttnArgs a, const int nqb, const int32_t* __restrict__ lens,
                                                         const unsigned long long* __restrict__ kwords, const int nwords) {
    constexpr int NW = 4;
    constexpr int CR = DH / 8;                     // 16-byte chunks per q / k / v row
    constexpr int NKS = DH / 16;                   // k-steps of S^T = K . Q^T
    constexpr int SLOT = 16384, VOFF = 8192;       // a tile slot: K [64][128 B] | V [64][128 B]
    constexpr int QOFF = kRing * SLOT;             // then the item's Q rows, [4 waves][32][128 B], each wave's part private to it
    constexpr int NST = DH == 64 ? 4 : 3;          // output stores per item and lane
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kRing * SLOT + NW * 4096];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int BH = a.B * a.H;
    // Which items a block takes.  XCD x (= block id mod 8, where a block of that id runs: an observed placement used for locality
    // only, any block may compute any item) owns the (clip, head) pairs x, x + 8, ...; its item list is pair-major -- item i is
    // query block nqb - 1 - i % nqb of pair x + 8 (i / nqb) -- and in round k the XCD's M blocks take the M consecutive items
    // k M .. k M + M - 1 (block m the one at offset (m + k) % M: the rotation spreads the causal query blocks' unequal tile counts
    // over the blocks).  The query blocks of ONE pair are thus multiplied at about the same time on ONE XCD: their K / V reach that
    // L2 from the fabric once and the other query blocks' reads hit it.  (A per-block list of whole pairs put 64 pairs x 77 KB
    // through a 4 MB L2 between two uses: 69 % of the re-reads missed and the kernel ran at the fabric's 5.6 TB/s.)
    const int xq = (int)(blockIdx.x & 7), mloc = (int)(blockIdx.x >> 3), M = (int)(gridDim.x >> 3);
    const int n_items = ((BH - xq + 7) >> 3) * nqb;             // items of this XCD
    auto item_index = [&](int k) -> int { return k * M + (mloc + k) % M; };
    auto pair_of = [&](int k) -> int {
        const int i = item_index(k);
        return i < n_items ? xq + 8 * (i / nqb) : -1;
    };
    auto len_of = [&](int pair) -> int { return (lens && pair >= 0) ? lens[pair / a.H] : a.Lk; };   // scalar load
    auto make_item = [&](int k, int len_b) -> AttnItem {
        AttnItem it;
        const int pair = pair_of(k);
        const int qb = nqb - 1 - item_index(k) % nqb;
        it.pair = pair;
        if (pair < 0) return it;
        const int b = pair / a.H, h = pair - b * a.H;
        it.b = b;
        it.qblk0 = qb * (32 * NW);
        it.Q = (const bf16*)a.q + (size_t)b * a.q_sb + (size_t)h * a.q_sh;
        it.K = (const bf16*)a.k + (size_t)b * a.k_sb + (size_t)h * a.k_sh;
        it.V = (const bf16*)a.vt + (size_t)b * a.v_sb + (size_t)h * a.v_sh;
        it.O = (bf16*)a.o + (size_t)b * a.o_sb + (size_t)h * a.o_sh;
        it.len_b = len_b;
        int kmax = it.len_b < a.Lk ? it.len_b : a.Lk;
        if (a.causal) {
            int last = it.qblk0 + 32 * NW - 1;
            last = last < a.Lq - 1 ? last : a.Lq - 1;
            kmax = (last + 1) < kmax ? (last + 1) : kmax;
        }
        it.kmax = kmax;
        it.ntiles = kmax > 0 ? (kmax + 63) / 64 : 1;    // an empty clip still takes one (fully masked) step; its output is zeroed
        return it;
    };

    // ---- LDS-DMA maps (global_load_lds, 16 B per lane: one instruction fills 1 KiB = 8 rows x 128 B, lane l -> row l >> 3,
    // slot l & 7; the XOR swizzle is applied to the SOURCE chunk).  Per tile this wave brings K pieces 2w, 2w+1 and V pieces
    // 2w, 2w+1 (FOUR instructions, whatever the head width: the counted waits rely on it); per item its own 32 query rows (four).
    const int r8 = lane >> 3, c8 = lane & 7;
    uint32_t koff[2], voff[2];
    int trow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (2 * wave + j) * 8 + r8;
        trow[j] = row;
        // a lane whose source chunk lies beyond a 48-wide row re-reads chunk 0 into its (never read) slot: no lane is predicated
        // off, the instruction count per tile stays four
        int ck = c8 ^ ((row >> 1) & 7), cv = c8 ^ (((row >> 1) & 1) << 2);
        ck = ck < CR ? ck : 0;
        cv = cv < CR ? cv : 0;
        koff[j] = ((uint32_t)row * (uint32_t)a.k_st + (uint32_t)ck * 8u) * 2u;
        voff[j] = ((uint32_t)row * (uint32_t)a.v_st + (uint32_t)cv * 8u) * 2u;
    }
    const uint32_t kstr = (uint32_t)a.k_st * 2u, vstr = (uint32_t)a.v_st * 2u;
    auto dma_tile = [&](const AttnItem& it, int j0, int slot) {
        const int lim = a.Lk - 1 - j0;                 // rows beyond Lk repeat the last row (finite values; their keys are masked)
        const char* kb = (const char*)it.K + (size_t)j0 * kstr;
        const char* vb = (const char*)it.V + (size_t)j0 * vstr;
        unsigned char* dst = smem + slot * SLOT + (2 * wave) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t ko = koff[j], vo = voff[j];
            if (lim < 63) {
                const uint32_t over = (uint32_t)(trow[j] > lim ? trow[j] - lim : 0);
                ko -= over * kstr;
                vo -= over * vstr;
            }
            __builtin_amdgcn_global_load_lds((glb_void_t*)(kb + ko), (lds_void_t*)(dst + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(vb + vo), (lds_void_t*)(dst + VOFF + j * 1024), 16, 0, 0);
        }
    };
    auto dma_q = [&](const AttnItem& it) {
        unsigned char* dst = smem + QOFF + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * j + r8;
            int cq = c8 ^ ((row >> 1) & 7);
            cq = cq < CR ? cq : 0;
            int q = it.qblk0 + 32 * wave + row;
            q = q < a.Lq ? q : a.Lq - 1;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(it.Q + (size_t)q * a.q_st + cq * 8), (lds_void_t*)(dst + j * 1024), 16, 0, 0);
        }
    };
    // validity word of tile t of the item's clip (a scalar load when packed words exist)
    auto key_word = [&](const AttnItem& it, int t) -> unsigned long long {
        if (kwords) return kwords[(size_t)it.b * nwords + t];
        const int lim = it.len_b < a.Lk ? it.len_b : a.Lk;
        const int nv = lim - 64 * t;
        return nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : ((1ull << nv) - 1ull));
    };

    // ---- per-lane LDS read addresses (slot 0; slot s adds s * SLOT)
    // K (and Q) as MFMA operands: row l31 (+ 32 kt by immediate), chunk 2 ks + half: (c ^ swz) = ((ks ^ sw >> 1) << 1) | (half ^ sw & 1)
    const int swk = (l31 >> 1) & 7;
    const uint32_t frag0 = l31 * 128 + (((half ^ (swk & 1)) | ((swk >> 1) << 1)) << 4);   // ks = 0; ks by ^ (ks << 5)
    const uint32_t kaddr = lds0 + frag0;
    const uint32_t qaddr = lds0 + QOFF + wave * 4096 + frag0;
    // V through the transposing read: 16-lane group g16 covers d = 32 blk + 16 g16 .. + 15; lane p of it addresses key
    // 4 half + (p >> 2) (+ 16 s + 32 kt + 8 second, by immediate), 8 bytes at d = 32 blk + 16 g16 + 4 (p & 3)
    const int p16 = lane & 15, g16 = (lane >> 4) & 1;
    const int vrow = 4 * half + (p16 >> 2);
    const int vsw = ((p16 >> 3) & 1) << 6;          // = ((row >> 1) & 1) << 6 for every row this lane addresses
    const uint32_t vaddr0 = lds0 + VOFF + vrow * 128 + ((32 * g16 + 8 * (p16 & 3)) ^ vsw);
    const uint32_t vaddr1 = lds0 + VOFF + vrow * 128 + ((64 + 32 * g16 + 8 * (p16 & 3)) ^ vsw);
    const float scale2 = a.scale * 1.4426950408889634f;

    if (pair_of(0) < 0) return;
    // ---- the two walkers over the block's (item, tile) sequence: `pf` requests tiles, three steps ahead of the one that computes
    int pf_k = 0, pf_tile = 0;
    int pf_len_next = len_of(pair_of(1));
    AttnItem pf = make_item(0, len_of(pair_of(0)));
    int gstep = 0;                                     // tiles computed so far (ring slot = step & 3)
    int issued = 0;                                    // tiles requested so far
    // Counted waits.  n_ops = vector-memory operations this wave has issued (an UNDER-count is safe: it only makes a wait
    // stricter); pos1 / pos2 / pos3 = n_ops right after the request of the tile one / two / three steps ahead (-1: none).
    // "Tile g + 1 has landed" <=> at most n_ops - pos1 operations are outstanding (they complete in order).
    int n_ops = 0, qpos = 0;
    int pos[3] = {-1, -1, -1};                         // pos[d - 1]: the tile d steps ahead
    auto request_next_tile = [&]() -> int {            // one tile (4 DMA instructions of this wave), if any is left
        if (pf.pair < 0) return -1;
        dma_tile(pf, 64 * pf_tile, issued & (kRing - 1));
        ++issued;
        n_ops += 4;
        if (++pf_tile == pf.ntiles) {
            pf_tile = 0;
            ++pf_k;
            pf = make_item(pf_k, pf_len_next);
            pf_len_next = len_of(pair_of(pf_k + 1));
        }
        return n_ops;
    };

    int k_item = 0;
    int len_nxt = len_of(pair_of(1));
    AttnItem cur = make_item(0, len_of(pair_of(0)));
    dma_q(cur);
    n_ops += 4;
    const int pos0 = request_next_tile();              // the first item has at least one tile
#pragma unroll
    for (int d = 1; d < kRing - 1; ++d) pos[d - 1] = request_next_tile();
    unsigned long long kw_next = key_word(cur, 0);
    wait_vm_at_most(n_ops - pos0);                     // tile 0 (and the Q rows, requested before it) have landed
    __builtin_amdgcn_s_barrier();

    for (;;) {
        // ---- item start: Q fragments from this wave's own LDS rows (their DMA was waited for at the last barrier of the previous
        // item), then the NEXT item's rows may overwrite them
        u32x4_t qf[4];                                  // (NKS used; the 48-wide heads read a fourth, unused chunk)
        lds_read_4_b128(qaddr, qaddr ^ 32u, qaddr ^ 64u, qaddr ^ 96u, qf);
        const AttnItem nxt = make_item(k_item + 1, len_nxt);
        len_nxt = len_of(pair_of(k_item + 2));
        if (nxt.pair >= 0) {
            dma_q(nxt);
            n_ops += 4;
            qpos = n_ops;
        }

        const int qlo = cur.qblk0 + wave * 32;
        const int qi = qlo + l31;
        const bool wave_on = qlo < a.Lq;
        const int qlast = (qlo + 31) < a.Lq ? (qlo + 31) : (a.Lq - 1);   // this wave's last query (causal tile skip)
        float m_run = kNegT, l_run = 0.f;
        f32x16_t ot[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;

        for (int tile = 0; tile < cur.ntiles; ++tile) {
            const int j0 = tile * 64;
            const uint32_t so = (uint32_t)(gstep & (kRing - 1)) * SLOT;
            const unsigned long long kbits = kw_next;
            const bool last_tile = tile + 1 == cur.ntiles;
            if (!last_tile)
                kw_next = key_word(cur, tile + 1);
            else if (nxt.pair >= 0)
                kw_next = key_word(nxt, 0);
            // ---- request the tile three steps ahead into the slot every wave left at the last barrier
            pos[kRing - 2] = (a.dbg & 1) ? -1 : request_next_tile();
            const bool on = wave_on && (!a.causal || j0 <= qlast);
            if (on && !(a.dbg & 8)) {
                // ---- S^T = K . Q^T
                u32x4_t kf[2][4];
                const uint32_t kb = kaddr + so;
                lds_read_4x2_b128(kb, kb ^ 32u, kb ^ 64u, kb ^ 96u, kf[0], kf[1]);
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
                float sc[32];                 // this lane's 32 scores: sc[16 kt + r] <-> key 32 kt + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[16 * kt + r] = st[kt][r];
                if (!interior) {
                    // one validity word per lane, bit 16 kt + r: the tile's key bits shifted by this lane's half, every second
                    // nibble kept and packed; causal: key offsets x (of 0..63, in this lane's numbering) with j0 + x <= qi
                    const uint32_t klo = half ? (uint32_t)(kbits >> 4) : (uint32_t)kbits;
                    const uint32_t khi = half ? (uint32_t)(kbits >> 36) : (uint32_t)(kbits >> 32);
                    uint32_t y0 = klo & 0x0F0F0F0Fu, y1 = khi & 0x0F0F0F0Fu;
                    y0 = (y0 | (y0 >> 4)) & 0x00FF00FFu;
                    y1 = (y1 | (y1 >> 4)) & 0x00FF00FFu;
                    y0 = (y0 | (y0 >> 8)) & 0xFFFFu;
                    y1 = (y1 | (y1 >> 8)) & 0xFFFFu;
                    uint32_t vm = y0 | (y1 << 16);
                    if (a.causal) {
                        const int nvis = qi - j0 - 4 * half + 1;              // visible key offsets x are x < nvis
                        int rem = nvis & 7;
                        rem = rem < 4 ? rem : 4;
                        int nb = 4 * (nvis >> 3) + rem;
                        nb = nvis <= 0 ? 0 : nb;
                        vm &= nb >= 32 ? ~0u : ((1u << nb) - 1u);
                    }
#pragma unroll
                    for (int i = 0; i < 32; ++i) sc[i] = ((vm >> i) & 1u) ? sc[i] : kNegT;
                }
                // (four independent chains each: one chain of 32 dependent instructions is what a wave with one partner on its
                // SIMD cannot hide)
                float mx4[4] = {kNegT, kNegT, kNegT, kNegT};
#pragma unroll
                for (int i = 0; i < 32; ++i) mx4[i & 3] = fmaxf(mx4[i & 3], sc[i]);
                float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
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
                float ps4[4] = {0.f, 0.f, 0.f, 0.f};
                u32x4_t pk[2][2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float p[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            p[e] = (a.dbg & 2) ? sc[16 * kt + 8 * s + e] : __builtin_amdgcn_exp2f(__builtin_fmaf(sc[16 * kt + 8 * s + e], scale2, nm));
                            ps4[e & 3] += p[e];
                        }
                        pk[kt][s] = u32x4_t{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]),
                                            pack_bf16x2(p[6], p[7])};
                    }
                l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);

                // ---- O^T += V^T . P^T (one output block's eight transposing reads, then its four matrix instructions)
                if (!(a.dbg & 4))
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    u32x2_t t[8];
                    lds_read_8_tr16((blk ? vaddr1 : vaddr0) + so, t);
                    // rows 16 (2 kt + s) + 4 half + {0..3} and + 8: t[2 (2 kt + s)], t[2 (2 kt + s) + 1]
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const int i = 2 * (2 * kt + s);
                            const u32x4_t va = {t[i].x, t[i].y, t[i + 1].x, t[i + 1].y};
                            ot[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, va),
                                                                              __builtin_bit_cast(bf16x8_t, pk[kt][s]), ot[blk], 0, 0, 0);
                        }
                }
            }
            // ---- end of the step: the next step's tile has landed (and, before an item's last barrier, the next item's Q rows);
            // what was requested after it stays in flight
            {
                int allow = pos[0] >= 0 ? n_ops - pos[0] : 64;
                if (last_tile && nxt.pair >= 0 && n_ops - qpos < allow) allow = n_ops - qpos;
                wait_vm_at_most(allow);
            }
            if (!(a.dbg & 16)) __builtin_amdgcn_s_barrier();
            ++gstep;
            pos[0] = pos[1];
            pos[1] = pos[2];
            pos[2] = -1;
        }

        // ---- finish the item: combine the two lane halves' row sums, normalise, store O[q][d] as 16-byte pieces
        const float l_tot = l_run + xor_lane_f32<32>(l_run);
        const float inv = (l_tot > 0.f && cur.kmax > 0) ? 1.0f / l_tot : 0.f;
        const int qc = qi < a.Lq ? qi : a.Lq - 1;
        bf16* orow = cur.O + (size_t)qc * a.o_st;
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
                    if (qi < a.Lq && !(a.dbg & 32)) *(u32x4_t*)(orow + 32 * blk + 16 * gp + 8 * half) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
                }
            }
        // the stores are younger than everything requested so far (a wave without a valid query issues none; under-counting what
        // may stay in flight is safe, over-counting is not)
        if (wave_on && !(a.dbg & 32)) n_ops += NST;
        if (nxt.pair < 0) break;
        cur = nxt;
        ++k_item;
    }
}

}  // namespace

int launch_pack_key_words(const uint8_t* kmask, int kmask_ld, const int32_t* lens, int B, int Lk, unsigned long long* words, size_t cap,
                          hipStream_t s) {
    const int nwords = ceil_div(Lk, 64);
    const size_t need = (size_t)B * nwords;
    DIMX_REQUIRE(kmask && words && cap >= need, DIMX_ERR_ARG, "pack_key_words: %zu words needed, %zu given", need, cap);
    hipLaunchKernelGGL(pack_key_words_kernel, dim3(ceil_div((int)need * 64, 256)), dim3(256), 0, s, kmask, kmask_ld, lens, B, Lk, nwords, words);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

// bf16, D in {48, 64}, q / k / v row-major with 16-byte aligned rows.  Returns DIMX_OK after launching.
int launch_attention_tr(const AttnArgs& a_in, hipStream_t s) {
    static const int dbg = getenv("DIMX_ATTN_DBG") ? atoi(getenv("DIMX_ATTN_DBG")) : 0;   // ablations (tools/bench_attn.py)
    static const int blocks_per_cu = getenv("DIMX_ATTN_BPC") ? atoi(getenv("DIMX_ATTN_BPC")) : 0;
    static const int ring = getenv("DIMX_ATTN_RING") ? atoi(getenv("DIMX_ATTN_RING")) : 2;
    AttnArgs a = a_in;
    a.dbg = dbg;
    DIMX_REQUIRE(a.dtype == DIMX_BF16 && (a.D == 48 || a.D == 64) && a.v_rows, DIMX_ERR_ARG, "attention_tr: bf16, D 48 / 64, row-major V");
    DIMX_REQUIRE(a.q_st % 8 == 0 && a.k_st % 8 == 0 && a.v_st % 8 == 0 && a.o_st % 8 == 0 && a.q_sh % 8 == 0 && a.k_sh % 8 == 0 &&
                     a.v_sh % 8 == 0 && a.o_sh % 8 == 0 && a.q_sb % 8 == 0 && a.k_sb % 8 == 0 && a.v_sb % 8 == 0 && a.o_sb % 8 == 0,
                 DIMX_ERR_ARG, "attention_tr: strides must keep 16-byte alignment");
    DIMX_REQUIRE((long)64 * (a.k_st > a.v_st ? a.k_st : a.v_st) * 2 < (1l << 31), DIMX_ERR_ARG, "attention_tr: row stride too large");
    const int nqb = ceil_div(a.Lq, 128);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        DIMX_HIP(hipGetDevice(&dev));
        DIMX_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    // key masks travel as 64-bit validity words (clip lengths folded in), packed on the launch stream into scratch the CALLER
    // provides (AttnArgs.kwords: a slice of the handle's workspace arena; until round 4 a process-wide static buffer that every
    // handle, stream and device shared and that was re-allocated inside the launch path)
    const unsigned long long* kwords = nullptr;
    const int nwords = ceil_div(a.Lk, 64);
    if (a.kmask) {
        const size_t need = (size_t)a.B * nwords;
        DIMX_REQUIRE(a.kwords && a.kwords_cap >= need, DIMX_ERR_ARG,
                     "attention_tr: a key mask needs %zu words of caller scratch (AttnArgs.kwords), got %zu", need, a.kwords ? a.kwords_cap : (size_t)0);
        if (!a.kwords_ready)
            hipLaunchKernelGGL(pack_key_words_kernel, dim3(ceil_div((int)need * 64, 256)), dim3(256), 0, s, a.kmask, a.kmask_ld, a.lens, a.B, a.Lk,
                               nwords, a.kwords);
        kwords = a.kwords;
    }
    // persistent blocks, 2 per CU (80 KiB of LDS each): each walks (clip, head) pairs g, g + grid, ...
    const int items = a.B * a.H * nqb;
    const int cap_blocks = n_cu * (blocks_per_cu > 0 ? blocks_per_cu : (ring == 2 ? 3 : 2));
    int nblk = items < cap_blocks ? items : cap_blocks;
    nblk = (nblk + 7) / 8 * 8;                       // the same number of blocks on every XCD
    dim3 grid(nblk), block(256);
    if (ring == 2) {
        if (a.D == 48)
            hipLaunchKernelGGL((attn_tr_kernel<48, 2>), grid, block, 0, s, a, nqb, a.lens, kwords, nwords);
        else
            hipLaunchKernelGGL((attn_tr_kernel<64, 2>), grid, block, 0, s, a, nqb, a.lens, kwords, nwords);
    } else {
        if (a.D == 48)
            hipLaunchKernelGGL((attn_tr_kernel<48, 4>), grid, block, 0, s, a, nqb, a.lens, kwords, nwords);
        else
            hipLaunchKernelGGL((attn_tr_kernel<64, 4>), grid, block, 0, s, a, nqb, a.lens, kwords, nwords);
    }
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
