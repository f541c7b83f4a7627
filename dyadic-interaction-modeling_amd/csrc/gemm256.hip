// gemm256.hip -- the prefill GEMM: C = epilogue(A[M,K] . W[N,K]^T), bf16 operands, M = clips x frames (tens of thousands of
// rows).  Every token-parallel Linear of the path with a row-contiguous destination takes it (reference call sites:
// the cross-attention to_k / to_v of the x-transformers Decoder built at code/seq2seq_pretrain.py:413-419 on the
// context of :445-446, the VQ-VAE / encoder MLPs and output projections of code/models/lib/base_models.py:43-146).
//
// Why a second GEMM kernel: the one-barrier-per-k-tile loop of gemm.hip tops out near 30-34 % of the bf16 MFMA peak
// at every tile size and ring depth (tools/bench_prefill.py); what lifts that ceiling on CDNA4 is a
// phase structure in which the two waves of a SIMD alternate between a matrix segment and a load segment:
//   * 256 x 256 block tile, BK = 64, 8 waves as 2 (M) x 4 (N); a k-tile is four 16 KiB half-tiles in LDS (A rows
//     0-127 / 128-255, W rows 0-127 / 128-255), two k-tile buffers = 128 KiB.
//   * a k-tile is four phases, one BLOCK quadrant (128 x 128) each; in a phase every wave multiplies its 64 x 32 piece
//     of that quadrant over K = 64 (8 x v_mfma_f32_32x32x16_bf16).  Quadrant order (0,0) (0,1) (1,1) (1,0): the A
//     fragments are re-read only when the row half changes and the first W half stays in registers, so a half-tile
//     is released one per phase and re-staged one per phase (2 global_load_lds per wave), five phases ahead of its
//     first read: 64 KiB of DMA in flight all the time, `s_waitcnt vmcnt(8)` once per phase, never vmcnt(0).
//   * waves 4-7 run half a phase behind waves 0-3 (one extra s_barrier at the start): on every SIMD one wave is in
//     its MFMA segment while its partner issues ds_reads / DMA -- two raw s_barriers per phase keep them in step.
//   * persistent blocks: a block walks its output tiles back to back, the staging cursor simply runs on into the next
//     tile, so there is no pipeline fill / drain per tile and the epilogue stores overlap the next tile's DMA.
//   * MFMA operands are swapped (weights as srcA, activations as srcB): a lane then holds 4 consecutive output
//     COLUMNS of one row, so the epilogue stores 8-byte (bf16) / 16-byte (f32) row-contiguous pieces with no LDS
//     staging (the LDS is full of the next tile's operands at that point).
//   * tile order: XCD x takes the row tiles tm = x (mod 8); inside an XCD consecutive blocks share a row tile (its A
//     panel is fetched once per XCD) and sweep `n_group` column tiles before moving on, so that W stays in L2.
#include "common.hpp"

namespace dimx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

constexpr int kHalf = 16384;        // one half-tile: 128 rows x 128 B
constexpr int kBuf = 4 * kHalf;     // one k-tile: A0h A1h B0h B1h
constexpr int kDefaultVar = 4;  // 4 = the two-phase loop (gemm256p2_kernel); 0..3 = the four-phase loop's DMA placements (tuning)
constexpr bool kOverlapEpi = true;
constexpr bool kPrio = true;  // s_setprio(1) around the MFMA segments: +7 % (without it 35.8 -> 33.3 % on the fused K/V projection)
constexpr int kLds = 2 * kBuf;
constexpr int kLdsTotal = kLds + 8 * 4096;  // + one 4 KiB epilogue scratch per wave = all 160 KiB

template <int OFF> __device__ __forceinline__ void ds_read128(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void raw_barrier() { __builtin_amdgcn_s_barrier(); }

struct Big {
    const bf16* A;
    const bf16* W;
    int lda, ldw, M, N, K;
    const float* bias;
    int act;
    const float* residual;
    int ldr;
    const float* rowadd;
    int rowadd_mode;
    float rowadd_scale;
    int rowadd_off, ld_rowadd, rowadd_div, rowT;
    OutSeg seg[8];
    int nseg, seg_width;
    int tiles_m, tiles_n, n_group, nkt;
    unsigned long long* prof;  // tuning only (PROF instantiation): interval sums of block 0, waves 0 and 4
    int nt_max;                // tiles of the busiest block
    unsigned desync_slack, desync_full;  // start delay (shader cycles) spread over blocks with / without a spare tile slot
    int abl;                   // tuning (DIMX_G256_ABL): 1 = no global stores in the epilogue, 2 = non-temporal bf16 stores
};

template <int ACT> __device__ __forceinline__ float act256(float x) {
    if (ACT == ACT_LEAKY) return x > 0.f ? x : 0.2f * x;
    if (ACT == ACT_GELU_TANH) {  // same fast forms as gemm.hip's bf16 epilogue
        const float u = 0.7978845608028654f * (x + 0.044715f * (x * x * x));
        const float e = __expf(2.0f * u);
        return x * (0.5f * (1.0f + (1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e))));
    }
    if (ACT == ACT_GELU_ERF) {
        const float z = fabsf(x) * 0.7071067811865476f;
        const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float er = 1.0f - poly * __expf(-z * z);
        return 0.5f * x * (1.0f + (x < 0.f ? -er : er));
    }
    return x;
}

// Tile order (one global sequence): column-tile groups of n_group (the last one may be narrower); inside a group row tile
// major, column tile minor -- consecutive positions share a row tile (its A panel is fetched once) and a W group that
// fits the L2.  XCD x owns the contiguous positions [T x / 8, T (x + 1) / 8) (round 3: balanced to +-1 tile; the modulo-8
// row split of round 2 left half the XCDs a whole row of tiles short), its blocks take them round-robin.
// XCD-local tile q of XCD x -> (tile_m, tile_n); false when past the XCD's last tile.
__device__ __forceinline__ bool tile_of(const Big& a, int x, int q, int& tm, int& tn) {
    const int total = a.tiles_m * a.tiles_n;
    const int lo = (int)((long)total * x >> 3), hi = (int)((long)total * (x + 1) >> 3);
    const int p = lo + q;
    if (p >= hi) return false;
    const int per_group = a.tiles_m * a.n_group;
    const int full = a.tiles_n / a.n_group;
    int gi, r, w;
    if (p < full * per_group) {
        gi = p / per_group;
        r = p - gi * per_group;
        w = a.n_group;
    } else {
        gi = full;
        r = p - full * per_group;
        w = a.tiles_n - full * a.n_group;
    }
    const int ti = r / w;
    tm = ti;
    tn = gi * a.n_group + (r - ti * w);
    return true;
}

// Epilogue of one 32-row x 64-column chunk of a wave (both column halves of row block rb of row half mh), through a
// private 4 KiB LDS scratch: the accumulators arrive "one row per lane, 4 consecutive columns per register group" and
// leave as 16-byte row-contiguous pieces, 8 lanes per 128-byte line (a direct store of the accumulator layout
// touches 32 lines per instruction and measured 24 us per tile).  bf16 output: one pass over 32 x 64; f32 output: two
// passes over 32 x 32.  Bias / activation run before the transpose, positional rows / residual after it.
//
// Everything the 64 lanes share is kept on the scalar unit (round 3): the chunk's first row / column are wave-uniform,
// so the output segment is ONE scalar lookup (a per-lane index made hipcc fetch the OutSeg with global loads and wait
// vmcnt(0) -- draining the 64 KiB of LDS-DMA the main loop keeps in flight, four times per tile), the bias comes through
// scalar loads for the same reason, and the (clip, frame) split of a row is one division per chunk instead of one
// 35-instruction sequence per row.
typedef const __attribute__((address_space(4))) float cfloat_t;
typedef float f32x8_t __attribute__((ext_vector_type(8)));

// What the four chunks of a wave's tile share, computed once per tile: the 64 output columns of a wave lie in one segment
// (seg_width % 64 == 0), so segment, head and column of this lane's 16-byte piece are fixed for the tile.
template <typename OutT> struct EpiTile {
    OutT* pcol;   // segment base + head offset + column of this lane's piece (per lane)
    long sb, st;  // element strides of (clip, frame) in the segment (wave-uniform)
    int n;        // first column of this lane's piece in pass 0 (f32 output: pass 1 is 32 columns further)
};
template <typename OutT>
__device__ __forceinline__ EpiTile<OutT> epi_tile(const Big& a, int ncol0, int lane) {
    constexpr bool BF = sizeof(OutT) == 2;
    EpiTile<OutT> e;
    int s = 0, nbase = ncol0 < a.N ? ncol0 : 0;
    if (a.nseg > 1) {
        s = nbase / a.seg_width;
        nbase -= s * a.seg_width;
    }
    const OutSeg sg = a.seg[s];  // wave-uniform index: scalar loads (a per-lane index made this a vmcnt(0) global load)
    const int c16 = lane & 7;
    const int nn = nbase + (BF ? c16 * 8 : c16 * 4);
    int hh, dd;
    if (sg.D == 64) {
        hh = nn >> 6;
        dd = nn & 63;
    } else {
        hh = nn / sg.D;
        dd = nn - hh * sg.D;
    }
    e.pcol = (OutT*)sg.ptr + (long)hh * sg.sh + dd;
    e.sb = sg.sb;
    e.st = sg.st;
    e.n = ncol0 + (BF ? c16 * 8 : c16 * 4);
    return e;
}

// The residual of one chunk of an f32 destination, requested ahead of the chunk's transpose (round 4): eight 16-byte pieces per lane
// (two column passes x four row groups, the pieces store_chunk adds after its read-back).  Loaded inside store_chunk, each pass
// stalled on its own four loads -- eight HBM round trips per tile and wave, 190-225 us for every 384-wide out / ff2 projection of
// the prefill whatever its K; one chunk ahead, the round trip hides behind the previous chunk's transpose and stores.
struct ResPiece {
    float4 v[8];
};
template <typename OutT, bool PLAIN>
__device__ __forceinline__ void load_residual(const Big& a, const EpiTile<OutT>& e, int mrow0, int ncol0, int lane, ResPiece& rp) {
    if (PLAIN || sizeof(OutT) == 2) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) rp.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!a.residual || mrow0 >= a.M || ncol0 >= a.N) return;
    const int r = lane >> 3;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mrow0 + r + 8 * i, n = e.n + ps * 32;
            if (m < a.M && n < a.N) rp.v[ps * 4 + i] = *(const float4*)(a.residual + (size_t)m * a.ldr + n);
        }
}

template <typename OutT, int ACT, bool PLAIN>
__device__ __forceinline__ void store_chunk(const Big& a, const EpiTile<OutT>& e, const f32x16_t& acc0, const f32x16_t& acc1,
                                            int mrow0, int ncol0, unsigned char* scratch, int lane, const ResPiece& rp) {
    const int half = lane >> 5, l31 = lane & 31;
    constexpr bool BF = sizeof(OutT) == 2;
    constexpr int PASSES = BF ? 1 : 2;
    if (mrow0 >= a.M || ncol0 >= a.N) return;  // wave-uniform: nothing of this chunk is inside the matrix
    // LDS traffic of the epilogue goes through inline asm: for a C++ access hipcc waits vmcnt(0) first (a pending LDS-DMA is
    // a pending LDS write to it) and drains the prefetch pipeline the next tile is about to need
    const unsigned sbase = (unsigned)(size_t)(lds_void_t*)scratch;
    // (clip, frame) of the chunk's first row: one wave-uniform division per chunk; the lane's four rows (lane / 8 + 8 i)
    // follow by additions (rowT == 1 or rowT >= 32, checked by the launcher: at most one wrap per step)
    int b0 = mrow0, t0 = 0;
    if (a.rowT > 1) {
        b0 = mrow0 / a.rowT;
        t0 = mrow0 - b0 * a.rowT;
    }
    const int r = lane >> 3;
    int bl = b0, tl = t0 + r;
    if (a.rowT > 1) {
        if (tl >= a.rowT) {
            tl -= a.rowT;
            ++bl;
        }
    } else {
        bl = b0 + r;
        tl = 0;
    }
    const long step8 = a.rowT > 1 ? 8 * e.st : 8 * e.sb;        // eight rows further inside a clip
    const long wrap = a.rowT > 1 ? e.sb - (long)a.rowT * e.st : 0;  // ... and across a clip boundary
    const long off0 = (long)bl * e.sb + (long)tl * e.st;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        // ---- write: row l31 of the chunk
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            if (!BF && nh != ps) continue;
            const f32x16_t& acc = nh == 0 ? acc0 : acc1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                if (!PLAIN && a.bias) {  // 8 consecutive columns through the scalar cache; this lane's half of them
                    const int n0 = ncol0 + nh * 32 + 8 * q;
                    const f32x8_t bv = *(const __attribute__((address_space(4))) f32x8_t*)(cfloat_t*)(a.bias + (n0 + 8 <= a.N ? n0 : 0));
                    v.x += half ? bv[4] : bv[0];
                    v.y += half ? bv[5] : bv[1];
                    v.z += half ? bv[6] : bv[2];
                    v.w += half ? bv[7] : bv[3];
                }
                v.x = act256<ACT>(v.x); v.y = act256<ACT>(v.y); v.z = act256<ACT>(v.z); v.w = act256<ACT>(v.w);
                if (BF) {
                    const int c16 = nh * 4 + q;  // 16-byte chunk of the 128-byte row; this lane's half of it
                    u32x2_t pk;
                    pk[0] = pack_bf16x2(v.x, v.y);
                    pk[1] = pack_bf16x2(v.z, v.w);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(sbase + (unsigned)(l31 * 128 + ((c16 ^ (l31 & 7)) << 4) + half * 8)), "v"(pk) : "memory");
                } else {
                    const int c16 = 2 * q + half;
                    asm volatile("ds_write_b128 %0, %1" ::"v"(sbase + (unsigned)(l31 * 128 + ((c16 ^ (l31 & 7)) << 4))), "v"(__builtin_bit_cast(u32x4_t, v)) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave reads it back: no barrier needed
        // ---- read back row-contiguous: 8 lanes per row, 4 row groups
        const int c16 = lane & 7;
        const int n = e.n + (BF ? 0 : ps * 32);
        OutT* const pcol = e.pcol + (BF ? 0 : ps * 32);
        u32x4_t raw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r + 8 * i;
            asm volatile("ds_read_b128 %0, %1" : "=v"(raw[i]) : "v"(sbase + (unsigned)(row * 128 + ((c16 ^ (row & 7)) << 4))) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        long off = off0;
        int b = bl, t = tl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mrow0 + r + 8 * i;
            if (i > 0) {
                off += step8;
                if (a.rowT > 1) {
                    t += 8;
                    if (t >= a.rowT) {
                        t -= a.rowT;
                        ++b;
                        off += wrap;
                    }
                } else {
                    b += 8;
                }
            }
            if (m >= a.M || n >= a.N || (a.abl & 1)) continue;
            OutT* p = pcol + off;
            if (BF) {
                uint4 o = __builtin_bit_cast(uint4, raw[i]);
                if (!PLAIN && (a.rowadd_mode || a.residual)) {  // rare for bf16 destinations: unpack, add, repack
                    float f[8];
                    const uint32_t w4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f[2 * k] = __builtin_bit_cast(float, w4[k] << 16);
                        f[2 * k + 1] = __builtin_bit_cast(float, w4[k] & 0xffff0000u);
                    }
                    if (a.rowadd_mode) {
                        const int ri = a.rowadd_mode == 1 ? t : (a.rowadd_mode == 2 ? b / a.rowadd_div + a.rowadd_off : a.rowadd_off);
                        const float* pr = a.rowadd + (size_t)ri * a.ld_rowadd + n;
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] += pr[k] * a.rowadd_scale;
                    }
                    if (a.residual) {
                        const float* pr = a.residual + (size_t)m * a.ldr + n;
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] += pr[k];
                    }
                    o = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
                }
                *(uint4*)p = o;
            } else {
                float4 v = __builtin_bit_cast(float4, raw[i]);
                if (!PLAIN && a.rowadd_mode) {
                    const int ri = a.rowadd_mode == 1 ? t : (a.rowadd_mode == 2 ? b / a.rowadd_div + a.rowadd_off : a.rowadd_off);
                    const float4 q4 = *(const float4*)(a.rowadd + (size_t)ri * a.ld_rowadd + n);
                    v.x += q4.x * a.rowadd_scale; v.y += q4.y * a.rowadd_scale; v.z += q4.z * a.rowadd_scale; v.w += q4.w * a.rowadd_scale;
                }
                if (!PLAIN && a.residual) {
                    const float4 q4 = rp.v[ps * 4 + i];   // requested one chunk ahead (load_residual)
                    v.x += q4.x; v.y += q4.y; v.z += q4.z; v.w += q4.w;
                }
                *(float4*)p = v;
            }
        }
    }
}

// VAR = where the two LDS-DMA pieces of a phase's half-tile re-staging are issued (tools/g256_var.py measures them):
//   0  both in the load segment (round 2)
//   1  both inside the wave's own MFMA segment of the PREVIOUS phase (same landing distance, no DMA in the load segment)
//   2  first piece in the load segment, second after the fourth MFMA of the same phase
//   3  both inside the MFMA segment of the same phase (after the 2nd and the 6th MFMA)
// A load segment that carries two DMA issues next to 4-12 ds_read_b128 is longer than the partner's 8-MFMA segment (the
// texture addresser takes 8 wave instructions from the four loading waves at once); among bare MFMAs an issue is cheap.
// PROF: s_memtime stamps at the three points of a phase where lgkmcnt(0) holds anyway (MFMA segment start / end, after the
// closing barrier); block 0, waves 0 and 4 write 4 x 3 interval sums + the k-tile count to a.prof.
template <typename OutT, int ACT, bool PLAIN, int VAR, bool PROF>
__global__ __launch_bounds__(512) void gemm256_kernel(const Big a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations / M0 stay on the SALU
    const int wr = wave >> 2, wc = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;

    // ---- this block's tiles: XCD-local indices j, j + nb, ...
    int ntiles = 0;
    {
        int tm, tn;
        for (int q = j; tile_of(a, x, q, tm, tn); q += nb) ++ntiles;
    }
    if (ntiles == 0) return;
    const int nkt = a.nkt;
    const long total = (long)ntiles * nkt;  // k-tiles this block computes
    // De-phase the blocks.  Persistent blocks of equal work run in lock step, so all 256 reach their epilogue together:
    // a 32 MB burst of stores (every CU's 128 KiB tile) that the memory side takes ~15k cycles to absorb while no CU
    // computes, then silence on the write path for a whole tile.  A start delay spread over one tile time turns that
    // into a steady stream under the other CUs' MFMA work.  Blocks with a spare tile slot (fewer tiles than the busiest
    // block) take the delay for free; the busiest blocks take desync_full (0 unless no block has slack).
    {
        const unsigned span = ntiles < a.nt_max ? a.desync_slack : a.desync_full;
        if (span) {
            const unsigned frac = ((blockIdx.x * 0x9E3779B1u) >> 16) & 0xffffu;  // well-spread over the blocks of every XCD
            const unsigned long long wait = ((unsigned long long)span * frac) >> 16;
            unsigned long long t0, t1;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            do {
                __builtin_amdgcn_s_sleep(32);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            } while (t1 - t0 < wait);
        }
    }

    // ---- DMA addressing: wave w moves local rows [16w, 16w + 16) of every half-tile as two 8-row pieces.  The W
    // half-tiles hold PERMUTED columns: local row lr of column half nh is column (lr / 32) * 64 + nh * 32 + lr % 32 of
    // the tile, so that a wave's two 32-column blocks are adjacent in the output (64 contiguous columns per row).
    const int r0 = 16 * wave + (lane >> 3), r1 = r0 + 8;
    const int dc0 = ((lane & 7) ^ ((r0 >> 1) & 7)) * 8, dc1 = ((lane & 7) ^ ((r1 >> 1) & 7)) * 8;
    const int wcol0 = (r0 >> 5) * 64 + (r0 & 31), wcol1 = (r1 >> 5) * 64 + (r1 & 31);
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    // A staging cursor covers one row half and one column half of the k-tile it points at (hA, hB fixed per cursor);
    // the four source pointers advance by one k-tile (128 B) per step and are rebuilt when the tile changes.
    struct Cur {
        int q, kt, hA, hB;
        unsigned a0, a1, b0, b1;  // element offsets from a.A / a.W (< 2^32: checked by the launcher)
    };
    auto bind_tile = [&](Cur& c) {
        int tm, tn;
        if (!tile_of(a, x, c.q, tm, tn)) return false;  // past the end: keep the previous tile (dummy re-staging)
        const int mo = tm * 256 + c.hA * 128, no = tn * 256 + c.hB * 32;
        int ra = mo + r0, rb = mo + r1, na = no + wcol0, nbb = no + wcol1;
        ra = ra < a.M ? ra : a.M - 1;
        rb = rb < a.M ? rb : a.M - 1;
        na = na < a.N ? na : a.N - 1;
        nbb = nbb < a.N ? nbb : a.N - 1;
        c.a0 = (unsigned)ra * (unsigned)a.lda + dc0;
        c.a1 = (unsigned)rb * (unsigned)a.lda + dc1;
        c.b0 = (unsigned)na * (unsigned)a.ldw + dc0;
        c.b1 = (unsigned)nbb * (unsigned)a.ldw + dc1;
        return true;
    };
    auto advance = [&](Cur& c) {
        if (++c.kt == nkt) {
            c.kt = 0;
            c.q += nb;
            if (!bind_tile(c)) {  // rewind the pointers of the kept tile
                c.a0 -= (unsigned)(nkt - 1) * 64; c.a1 -= (unsigned)(nkt - 1) * 64;
                c.b0 -= (unsigned)(nkt - 1) * 64; c.b1 -= (unsigned)(nkt - 1) * 64;
            }
        } else {
            c.a0 += 64; c.a1 += 64; c.b0 += 64; c.b1 += 64;
        }
    };
    // piece 0 / 1 of a half-tile's re-staging (PC = -1: both)
    auto stage_a = [&](const Cur& c, int buf, int pc) {
        unsigned char* dst = smem + buf * kBuf + c.hA * kHalf + wave * 2048;
        if (pc != 1) __builtin_amdgcn_global_load_lds((glb_void_t*)(a.A + c.a0), (lds_void_t*)dst, 16, 0, 0);
        if (pc != 0) __builtin_amdgcn_global_load_lds((glb_void_t*)(a.A + c.a1), (lds_void_t*)(dst + 1024), 16, 0, 0);
    };
    auto stage_b = [&](const Cur& c, int buf, int pc) {
        unsigned char* dst = smem + buf * kBuf + (2 + c.hB) * kHalf + wave * 2048;
        if (pc != 1) __builtin_amdgcn_global_load_lds((glb_void_t*)(a.W + c.b0), (lds_void_t*)dst, 16, 0, 0);
        if (pc != 0) __builtin_amdgcn_global_load_lds((glb_void_t*)(a.W + c.b1), (lds_void_t*)(dst + 1024), 16, 0, 0);
    };

    // ---- fragment addressing (local row inside a half-tile; the swizzle term depends on l31 only)
    const unsigned sw = (l31 >> 1) & 7;
    unsigned aaddr[4], baddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned ko = ((unsigned)(2 * ks + half) ^ sw) << 4;
        aaddr[ks] = lds0 + (wr * 64 + l31) * 128 + ko;
        baddr[ks] = lds0 + 2 * kHalf + (wc * 32 + l31) * 128 + ko;
    }

    f32x16_t acc[2][2][2];  // [row half][col half][row block]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) (&acc[0][0][0])[i][r] = 0.f;

    // ---- prologue: A0h B0h B1h A1h of k-tile 0, A0h B0h of k-tile 1 (the steady-state issue order)
    int cu_q = j, cu_kt = 0, cu_m0 = 0, cu_n0 = 0;  // compute cursor
    {
        int tm, tn;
        tile_of(a, x, cu_q, tm, tn);
        cu_m0 = tm * 256;
        cu_n0 = tn * 256;
    }
    Cur c2{j, 0, 0, 0, 0u, 0u, 0u, 0u};  // halves (A0h, B0h): runs two k-tiles ahead
    Cur c1{j, 0, 1, 1, 0u, 0u, 0u, 0u};  // halves (A1h, B1h): runs one k-tile ahead
    bind_tile(c2);
    bind_tile(c1);
    stage_a(c2, 0, -1);
    stage_b(c2, 0, -1);
    stage_b(c1, 0, -1);
    stage_a(c1, 0, -1);
    advance(c2);  // -> k-tile 1
    advance(c1);  // -> k-tile 1
    stage_a(c2, 1, -1);
    stage_b(c2, 1, -1);
    advance(c2);  // -> k-tile 2
    if (VAR == 1) {  // phase 1's re-staging is issued one phase early in this form
        stage_b(c1, 1, -1);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    raw_barrier();
    if (wr == 1) raw_barrier();  // waves 4-7 run half a phase behind

    unsigned long long psum[17];
    unsigned long long tprev = 0;
    if (PROF) {
#pragma unroll
        for (int i = 0; i < 17; ++i) psum[i] = 0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
    }
#define G256_STAMP(SLOT)                                                                    \
    do {                                                                                    \
        if (PROF) {                                                                         \
            unsigned long long tn_;                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");     \
            psum[SLOT] += tn_ - tprev;                                                      \
            tprev = tn_;                                                                    \
        }                                                                                   \
    } while (0)
#define G256_WAIT_LOADSEG()                                                   \
    do {                                                                      \
        if (PROF && (a.abl & 8)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (VAR == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");        \
        else if (VAR == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                 \
    } while (0)
// one MFMA segment: quadrant accumulators ACC[0..1] += FB[ks] x fa[rb][ks]; H1 / H2 / H3 run after the 2nd / 4th / 6th MFMA
#define G256_MFMA_SEG(ACC, FB, H1, H2, H3)                                                                              \
    do {                                                                                                                \
        if (kPrio) __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                              \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                                            \
                ACC[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FB[ks]),                 \
                                                                  __builtin_bit_cast(bf16x8_t, fa[rb][ks]), ACC[rb], 0, 0, 0); \
            if (VAR != 0) {                                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
                if (ks == 0) { H1; }                                                                                    \
                if (ks == 1) { H2; }                                                                                    \
                if (ks == 2) { H3; }                                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
            }                                                                                                           \
        }                                                                                                               \
        if (kPrio) __builtin_amdgcn_s_setprio(0);                                                                       \
    } while (0)
#define G256_NOP ((void)0)

    u32x4_t fa[2][4], fb0[4], fb1[4];
    unsigned bo = 0;  // byte offset of the k-tile buffer being computed
    for (long g = 0; g < total; ++g) {
        const int buf = (int)(bo != 0);
        // ================= phase 1: quadrant (0,0) -- read A(row half 0) and W(col half 0); stage B1h of g + 1
        {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (!(PROF && (a.abl & 4))) ds_read128<0>(fb0[ks], baddr[ks] + bo);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (PROF && (a.abl & 16)) continue;
                ds_read128<0>(fa[0][ks], aaddr[ks] + bo);
                ds_read128<4096>(fa[1][ks], aaddr[ks] + bo);
            }
        }
        if (VAR == 0) stage_b(c1, buf ^ 1, -1);
        if (VAR == 2) stage_b(c1, buf ^ 1, 0);
        if (PROF && (a.abl & 8)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G256_STAMP(13); }
        G256_WAIT_LOADSEG();
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        G256_STAMP(0);
        if (VAR == 1) G256_MFMA_SEG(acc[0][0], fb0, stage_a(c1, buf ^ 1, 0), G256_NOP, stage_a(c1, buf ^ 1, 1));
        else if (VAR == 2) G256_MFMA_SEG(acc[0][0], fb0, G256_NOP, stage_b(c1, buf ^ 1, 1), G256_NOP);
        else if (VAR == 3) G256_MFMA_SEG(acc[0][0], fb0, stage_b(c1, buf ^ 1, 0), G256_NOP, stage_b(c1, buf ^ 1, 1));
        else G256_MFMA_SEG(acc[0][0], fb0, G256_NOP, G256_NOP, G256_NOP);
        G256_STAMP(1);
        raw_barrier();
        G256_STAMP(2);
        // ================= phase 2: quadrant (0,1) -- read W(col half 1); stage A1h of g + 1
        {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ds_read128<kHalf>(fb1[ks], baddr[ks] + bo);
        }
        if (VAR == 0) stage_a(c1, buf ^ 1, -1);
        if (VAR == 2) stage_a(c1, buf ^ 1, 0);
        if (PROF && (a.abl & 8)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G256_STAMP(14); }
        G256_WAIT_LOADSEG();
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        G256_STAMP(3);
        if (VAR == 1) G256_MFMA_SEG(acc[0][1], fb1, stage_a(c2, buf, 0), G256_NOP, stage_a(c2, buf, 1));
        else if (VAR == 2) G256_MFMA_SEG(acc[0][1], fb1, G256_NOP, stage_a(c1, buf ^ 1, 1), G256_NOP);
        else if (VAR == 3) G256_MFMA_SEG(acc[0][1], fb1, stage_a(c1, buf ^ 1, 0), G256_NOP, stage_a(c1, buf ^ 1, 1));
        else G256_MFMA_SEG(acc[0][1], fb1, G256_NOP, G256_NOP, G256_NOP);
        G256_STAMP(4);
        raw_barrier();
        G256_STAMP(5);
        advance(c1);
        // ================= phase 3: quadrant (1,1) -- read A(row half 1); stage A0h of g + 2
        {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                ds_read128<kHalf>(fa[0][ks], aaddr[ks] + bo);
                ds_read128<kHalf + 4096>(fa[1][ks], aaddr[ks] + bo);
            }
        }
        if (VAR == 0) stage_a(c2, buf, -1);
        if (VAR == 2) stage_a(c2, buf, 0);
        if (PROF && (a.abl & 8)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G256_STAMP(15); }
        G256_WAIT_LOADSEG();
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        G256_STAMP(6);
        if (VAR == 1) G256_MFMA_SEG(acc[1][1], fb1, stage_b(c2, buf, 0), G256_NOP, stage_b(c2, buf, 1));
        else if (VAR == 2) G256_MFMA_SEG(acc[1][1], fb1, G256_NOP, stage_a(c2, buf, 1), G256_NOP);
        else if (VAR == 3) G256_MFMA_SEG(acc[1][1], fb1, stage_a(c2, buf, 0), G256_NOP, stage_a(c2, buf, 1));
        else G256_MFMA_SEG(acc[1][1], fb1, G256_NOP, G256_NOP, G256_NOP);
        G256_STAMP(7);
        raw_barrier();
        G256_STAMP(8);
        // ================= phase 4: quadrant (1,0) -- everything is in registers; stage B0h of g + 2
        if (VAR == 0) stage_b(c2, buf, -1);
        if (VAR == 2) stage_b(c2, buf, 0);
        if (PROF && (a.abl & 8)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G256_STAMP(16); }
        G256_WAIT_LOADSEG();
        raw_barrier();
        __builtin_amdgcn_sched_barrier(0);
        G256_STAMP(9);
        if (VAR == 1) {  // c1 already points one k-tile further (advanced after phase 2): next phase 1's B1h into THIS buffer
            G256_MFMA_SEG(acc[1][0], fb0, stage_b(c1, buf, 0), G256_NOP, stage_b(c1, buf, 1));
        } else if (VAR == 2) G256_MFMA_SEG(acc[1][0], fb0, G256_NOP, stage_b(c2, buf, 1), G256_NOP);
        else if (VAR == 3) G256_MFMA_SEG(acc[1][0], fb0, stage_b(c2, buf, 0), G256_NOP, stage_b(c2, buf, 1));
        else G256_MFMA_SEG(acc[1][0], fb0, G256_NOP, G256_NOP, G256_NOP);
        G256_STAMP(10);
        raw_barrier();
        G256_STAMP(11);
        advance(c2);
        bo ^= (unsigned)kBuf;

        // ================= end of an output tile: store, clear, move on (no barriers in here)
        if (++cu_kt == nkt) {
            unsigned char* scratch = smem + kLds + wave * 4096;
            // Both wave groups store at the same time: group 0 lets group 1 finish its last MFMA segment first (one extra
            // barrier here), group 1 pays its extra barrier after the stores -- the half-phase stagger is the same afterwards.
            // Without the pair the stagger barriers serialise the two epilogues (group 1 waits for group 0's stores at its
            // closing barrier, then group 0 waits for group 1's).
            if (kOverlapEpi && wr == 0) raw_barrier();
            const EpiTile<OutT> et = epi_tile<OutT>(a, cu_n0 + wc * 64, lane);
            ResPiece res[2];
            load_residual<OutT, PLAIN>(a, et, cu_m0 + wr * 64, cu_n0 + wc * 64, lane, res[0]);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int mh = ch >> 1, rb = ch & 1;
                if (ch < 3)
                    load_residual<OutT, PLAIN>(a, et, cu_m0 + ((ch + 1) >> 1) * 128 + wr * 64 + ((ch + 1) & 1) * 32, cu_n0 + wc * 64, lane,
                                               res[(ch + 1) & 1]);
                store_chunk<OutT, ACT, PLAIN>(a, et, acc[mh][0][rb], acc[mh][1][rb], cu_m0 + mh * 128 + wr * 64 + rb * 32,
                                              cu_n0 + wc * 64, scratch, lane, res[ch & 1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[mh][0][rb][r] = 0.f;
                    acc[mh][1][rb][r] = 0.f;
                }
            }
            cu_kt = 0;
            cu_q += nb;
            int tm, tn;
            if (tile_of(a, x, cu_q, tm, tn)) {
                cu_m0 = tm * 256;
                cu_n0 = tn * 256;
            }
            if (kOverlapEpi && wr == 1) raw_barrier();
            if (PROF) {  // own epilogue in its own slot
                unsigned long long tn_;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");
                psum[12] += tn_ - tprev;
                tprev = tn_;
            }
        }
    }
    if (wr == 0) raw_barrier();  // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dummy stagings past the end must land before the LDS is released
    if (PROF && a.prof && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) {
        unsigned long long* o = a.prof + (wave >> 2) * 32;
#pragma unroll
        for (int i = 0; i < 17; ++i) o[i] = psum[i];
        o[17] = (unsigned long long)total;
    }
#undef G256_STAMP
#undef G256_WAIT_LOADSEG
#undef G256_MFMA_SEG
#undef G256_NOP
}


// ---------------------------------------------------------------------------------------------------------------------
// Two phases per k-tile (round 3; the default).  The four-phase loop above pays two barrier rendezvous per 8 MFMAs of a wave.
// Here a phase is HALF a k-tile -- both column halves of one row half, 16 MFMAs per wave -- so the same rendezvous cost is
// paid half as often:
//   phase A: read W(col half 0), W(col half 1), A(row half 0)  [16 ds_read_b128];  quadrants (0,0) and (0,1)
//   phase B: read A(row half 1)                                 [ 8 ds_read_b128];  quadrants (1,1) and (1,0)
// Staging: the three half-tiles phase A reads are free after it and are re-staged (k-tile g + 2) inside phase B's MFMA
// segment (6 LDS-DMA pieces per wave, one after every second MFMA); A(row half 1) is read in phase B and re-staged inside
// the next phase A (2 pieces).  Every half-tile is issued >= 3 phases before its first read and retired by a counted
// vmcnt one phase before it (vmcnt(6) in phase A's load segment leaves phase B's six pieces in flight, vmcnt(2) in phase
// B's leaves phase A's two).  Waves 4-7 still run half a phase behind waves 0-3; the epilogue and the tile walk are
// the four-phase kernel's.
// Measured on the K/V projection (M 76800, N 6144, K 1152; tools/g256_var.py, same box, interleaved): 40.4 % of the bf16
// peak against 39.2 % for the four-phase loop.  Forms that measured behind it: all eight pieces in the load segments
// (4 + 4, clean 16-MFMA segments): 38.8 %; 4 + 4 inside the segments with the issue slots alternating between even and odd
// waves (a wave-uniform branch per slot): 35.7 %.  Ablations of this loop (wrong results, stores off): no LDS-DMA 49.8 %, no
// ds_read 55.7 %, MFMAs alone 70.7 %, the barrier / bookkeeping skeleton alone takes 348 us of the 1040; the same kernel on
// zero-filled operands runs 36 % faster (846 vs 1149 us): with random operands the matrix pipe is power-limited to ~1.75
// GHz, so ~72 % of the 2.4 GHz peak is all there is to reach.
template <typename OutT, int ACT, bool PLAIN>
__global__ __launch_bounds__(512) void gemm256p2_kernel(const Big a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;
    int ntiles = 0;
    {
        int tm, tn;
        for (int q = j; tile_of(a, x, q, tm, tn); q += nb) ++ntiles;
    }
    if (ntiles == 0) return;
    const int nkt = a.nkt;
    const long total = (long)ntiles * nkt;

    const int r0 = 16 * wave + (lane >> 3), r1 = r0 + 8;
    const int dc0 = ((lane & 7) ^ ((r0 >> 1) & 7)) * 8, dc1 = ((lane & 7) ^ ((r1 >> 1) & 7)) * 8;
    const int wcol0 = (r0 >> 5) * 64 + (r0 & 31), wcol1 = (r1 >> 5) * 64 + (r1 & 31);
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    // cursor X: A(row half 0) + both W halves, two k-tiles ahead; cursor Y: A(row half 1), one k-tile ahead
    struct CurX {
        int q, kt;
        unsigned a0, a1, b0, b1, b2, b3;
    };
    struct CurY {
        int q, kt;
        unsigned a0, a1;
    };
    auto bind_x = [&](CurX& c) {
        int tm, tn;
        if (!tile_of(a, x, c.q, tm, tn)) return false;
        const int mo = tm * 256, no = tn * 256;
        int ra = mo + r0, rb = mo + r1;
        ra = ra < a.M ? ra : a.M - 1;
        rb = rb < a.M ? rb : a.M - 1;
        int n0 = no + wcol0, n1 = no + wcol1, n2 = n0 + 32, n3 = n1 + 32;
        n0 = n0 < a.N ? n0 : a.N - 1;
        n1 = n1 < a.N ? n1 : a.N - 1;
        n2 = n2 < a.N ? n2 : a.N - 1;
        n3 = n3 < a.N ? n3 : a.N - 1;
        c.a0 = (unsigned)ra * (unsigned)a.lda + dc0;
        c.a1 = (unsigned)rb * (unsigned)a.lda + dc1;
        c.b0 = (unsigned)n0 * (unsigned)a.ldw + dc0;
        c.b1 = (unsigned)n1 * (unsigned)a.ldw + dc1;
        c.b2 = (unsigned)n2 * (unsigned)a.ldw + dc0;
        c.b3 = (unsigned)n3 * (unsigned)a.ldw + dc1;
        return true;
    };
    auto bind_y = [&](CurY& c) {
        int tm, tn;
        if (!tile_of(a, x, c.q, tm, tn)) return false;
        const int mo = tm * 256 + 128;
        int ra = mo + r0, rb = mo + r1;
        ra = ra < a.M ? ra : a.M - 1;
        rb = rb < a.M ? rb : a.M - 1;
        c.a0 = (unsigned)ra * (unsigned)a.lda + dc0;
        c.a1 = (unsigned)rb * (unsigned)a.lda + dc1;
        return true;
    };
    auto advance_x = [&](CurX& c) {
        if (++c.kt == nkt) {
            c.kt = 0;
            c.q += nb;
            if (!bind_x(c)) {
                const unsigned back = (unsigned)(nkt - 1) * 64;
                c.a0 -= back; c.a1 -= back; c.b0 -= back; c.b1 -= back; c.b2 -= back; c.b3 -= back;
            }
        } else {
            c.a0 += 64; c.a1 += 64; c.b0 += 64; c.b1 += 64; c.b2 += 64; c.b3 += 64;
        }
    };
    auto advance_y = [&](CurY& c) {
        if (++c.kt == nkt) {
            c.kt = 0;
            c.q += nb;
            if (!bind_y(c)) {
                const unsigned back = (unsigned)(nkt - 1) * 64;
                c.a0 -= back; c.a1 -= back;
            }
        } else {
            c.a0 += 64; c.a1 += 64;
        }
    };
    // one 1 KiB LDS-DMA piece: 8 rows x 128 B of A or W -> half-tile `slot` (0 A0h, 1 A1h, 2 B0h, 3 B1h) of buffer `buf`
    auto piece = [&](const bf16* base, unsigned off, int buf, int slot, int second) {
        unsigned char* dst = smem + buf * kBuf + slot * kHalf + wave * 2048 + second * 1024;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(base + off), (lds_void_t*)dst, 16, 0, 0);
    };

    const unsigned sw = (l31 >> 1) & 7;
    unsigned aaddr[4], baddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned ko = ((unsigned)(2 * ks + half) ^ sw) << 4;
        aaddr[ks] = lds0 + (wr * 64 + l31) * 128 + ko;
        baddr[ks] = lds0 + 2 * kHalf + (wc * 32 + l31) * 128 + ko;
    }
    f32x16_t acc[2][2][2];  // [row half][col half][row block]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) (&acc[0][0][0])[i][r] = 0.f;

    int cu_q = j, cu_kt = 0, cu_m0 = 0, cu_n0 = 0;
    {
        int tm, tn;
        tile_of(a, x, cu_q, tm, tn);
        cu_m0 = tm * 256;
        cu_n0 = tn * 256;
    }
    // ---- prologue, in the steady-state issue order: {A0h B0h B1h}(0), A1h(0), {A0h B0h B1h}(1)
    CurX cx{j, 0, 0u, 0u, 0u, 0u, 0u, 0u};
    CurY cy{j, 0, 0u, 0u};
    bind_x(cx);
    bind_y(cy);
    piece(a.A, cx.a0, 0, 0, 0); piece(a.A, cx.a1, 0, 0, 1);
    piece(a.W, cx.b0, 0, 2, 0); piece(a.W, cx.b1, 0, 2, 1);
    piece(a.W, cx.b2, 0, 3, 0); piece(a.W, cx.b3, 0, 3, 1);
    piece(a.A, cy.a0, 0, 1, 0); piece(a.A, cy.a1, 0, 1, 1);
    advance_x(cx);  // -> k-tile 1
    advance_y(cy);  // -> k-tile 1
    piece(a.A, cx.a0, 1, 0, 0); piece(a.A, cx.a1, 1, 0, 1);
    piece(a.W, cx.b0, 1, 2, 0); piece(a.W, cx.b1, 1, 2, 1);
    piece(a.W, cx.b2, 1, 3, 0); piece(a.W, cx.b3, 1, 3, 1);
    advance_x(cx);  // -> k-tile 2
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // {A0h B0h B1h}(0) landed
    raw_barrier();
    if (wr == 1) raw_barrier();  // waves 4-7 run half a phase behind

#define P2_MFMA(ACC, FB, KS, RB)                                                                                          \
    ACC[RB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FB[KS]), __builtin_bit_cast(bf16x8_t, fa[RB][KS]), \
                                                      ACC[RB], 0, 0, 0)
// 16 MFMAs: for every k-step the two row blocks of quadrant X, then of quadrant Y; HOOK(i) runs after MFMA pair i (0..7)
#define P2_SEG(ACCX, FBX, ACCY, FBY, HOOK)                   \
    do {                                                     \
        if (kPrio) __builtin_amdgcn_s_setprio(1);            \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {   \
            P2_MFMA(ACCX, FBX, ks, 0);                       \
            P2_MFMA(ACCX, FBX, ks, 1);                       \
            __builtin_amdgcn_sched_barrier(0);               \
            HOOK(2 * ks);                                    \
            __builtin_amdgcn_sched_barrier(0);               \
            P2_MFMA(ACCY, FBY, ks, 0);                       \
            P2_MFMA(ACCY, FBY, ks, 1);                       \
            __builtin_amdgcn_sched_barrier(0);               \
            HOOK(2 * ks + 1);                                \
            __builtin_amdgcn_sched_barrier(0);               \
        }                                                    \
        if (kPrio) __builtin_amdgcn_s_setprio(0);            \
    } while (0)
// phase A's segment re-stages A1h of k-tile g + 1 (cursor Y) into the OTHER buffer: pieces after MFMA pairs 1 and 5
#define P2_HOOK_A(i)                                          \
    do {                                                      \
        if ((i) == 1) piece(a.A, cy.a0, buf ^ 1, 1, 0);       \
        if ((i) == 5) piece(a.A, cy.a1, buf ^ 1, 1, 1);       \
    } while (0)
// phase B's segment re-stages {A0h B0h B1h} of k-tile g + 2 (cursor X) into THIS buffer: pieces after pairs 0..5
#define P2_HOOK_B(i)                                          \
    do {                                                      \
        if ((i) == 0) piece(a.A, cx.a0, buf, 0, 0);           \
        if ((i) == 1) piece(a.A, cx.a1, buf, 0, 1);           \
        if ((i) == 2) piece(a.W, cx.b0, buf, 2, 0);           \
        if ((i) == 3) piece(a.W, cx.b1, buf, 2, 1);           \
        if ((i) == 4) piece(a.W, cx.b2, buf, 3, 0);           \
        if ((i) == 5) piece(a.W, cx.b3, buf, 3, 1);           \
    } while (0)

    u32x4_t fa[2][4], fb0[4], fb1[4];
    unsigned bo = 0;
    for (long g = 0; g < total; ++g) {
        const int buf = (int)(bo != 0);
        // ================= phase A: quadrants (0,0), (0,1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ds_read128<0>(fb0[ks], baddr[ks] + bo);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            ds_read128<0>(fa[0][ks], aaddr[ks] + bo);
            ds_read128<4096>(fa[1][ks], aaddr[ks] + bo);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ds_read128<kHalf>(fb1[ks], baddr[ks] + bo);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // A1h of this k-tile has landed (read next phase)
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        P2_SEG(acc[0][0], fb0, acc[0][1], fb1, P2_HOOK_A);
        raw_barrier();
        advance_y(cy);
        // ================= phase B: quadrants (1,1), (1,0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            ds_read128<kHalf>(fa[0][ks], aaddr[ks] + bo);
            ds_read128<kHalf + 4096>(fa[1][ks], aaddr[ks] + bo);
        }
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // {A0h B0h B1h} of the next k-tile have landed
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        P2_SEG(acc[1][1], fb1, acc[1][0], fb0, P2_HOOK_B);
        raw_barrier();
        advance_x(cx);
        bo ^= (unsigned)kBuf;

        if (++cu_kt == nkt) {
            unsigned char* scratch = smem + kLds + wave * 4096;
            if (kOverlapEpi && wr == 0) raw_barrier();
            const EpiTile<OutT> et = epi_tile<OutT>(a, cu_n0 + wc * 64, lane);
            ResPiece res[2];
            load_residual<OutT, PLAIN>(a, et, cu_m0 + wr * 64, cu_n0 + wc * 64, lane, res[0]);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int mh = ch >> 1, rb = ch & 1;
                if (ch < 3)
                    load_residual<OutT, PLAIN>(a, et, cu_m0 + ((ch + 1) >> 1) * 128 + wr * 64 + ((ch + 1) & 1) * 32, cu_n0 + wc * 64, lane,
                                               res[(ch + 1) & 1]);
                store_chunk<OutT, ACT, PLAIN>(a, et, acc[mh][0][rb], acc[mh][1][rb], cu_m0 + mh * 128 + wr * 64 + rb * 32,
                                              cu_n0 + wc * 64, scratch, lane, res[ch & 1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[mh][0][rb][r] = 0.f;
                    acc[mh][1][rb][r] = 0.f;
                }
            }
            cu_kt = 0;
            cu_q += nb;
            int tm, tn;
            if (tile_of(a, x, cu_q, tm, tn)) {
                cu_m0 = tm * 256;
                cu_n0 = tn * 256;
            }
            if (kOverlapEpi && wr == 1) raw_barrier();
        }
    }
    if (wr == 0) raw_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef P2_MFMA
#undef P2_SEG
#undef P2_HOOK_A
#undef P2_HOOK_B
}

}  // namespace

// Eligible: bf16 operands, K % 64 == 0, row-contiguous destinations, enough rows to fill the chip.
bool gemm256_eligible(const GemmArgs& g) {
    static const bool off = getenv("DIMX_NO_G256") != nullptr;
    if (off || g.in_dtype != DIMX_BF16 || g.conv_T != 0 || g.out_slabs || g.force_simple) return false;
    const int kext = g.kloop ? g.kloop : g.ldw;
    if (g.K % 64 != 0 || g.K != kext || g.M < 4096 || g.N < 256 || g.N % 8 != 0) return false;
    // measured (tools/bench_prefill.py, round 3, two-phase loop + scalar epilogue): ahead of the 128 x 128 kernel on every
    // prefill shape with min(N, K) >= 384 and N K >= 384 x 1536 -- (N 1536, K 384) 225 -> 199 us, (N 384, K 1536) 156 -> 128 us,
    // (N 2304, K 384) 276 -> 174 us, (N 1152, K 384: the VQ stacks' fused q/k/v, row-contiguous since V went row-major) 131 -> 85 us,
    // level from K = 1152 up; the 384 x 384 projections stay behind (76 -> 106 us here)
    static const bool all = getenv("DIMX_G256_ALL") != nullptr;
    if (!all && (g.K < 384 || g.N < 384 || (long)g.N * g.K < 384L * 1152L)) return false;
    // ... and only with enough 256 x 256 tiles for the 256 CUs: at 4 800 rows (a training batch, a 16-clip prefill) N 768 K 1152
    // has 57 tiles and took 29.7 us against 18.9 on the 128 x 128 kernel, N 384 K 1536 (38 tiles) 33.1 against 13.0 on the
    // 64 x 64 one, 95 tiles (N 1152) 6 - 15 % behind; N 4608 K 1152 (342 tiles) 68.9 against 85.5 stays, and so do the 114 tiles
    // of N 1536 K 384 with the GELU / bf16 epilogue (a 16-clip prefill was 0.5 ms slower without them) -- tools/
    // bench_train_gemm.py, profiles/r03_train_gemm.txt
    if (!all && (long)ceil_div(g.M, 256) * ceil_div(g.N, 256) < 100) return false;
    if ((size_t)g.M * g.lda >= (1ull << 32) || (size_t)g.N * g.ldw >= (1ull << 32)) return false;  // 32-bit element offsets
    if (g.bias && ((uintptr_t)g.bias % 16)) return false;
    if (g.residual && (g.ldr % 4 || (uintptr_t)g.residual % 16)) return false;
    if (g.rowadd_mode && (g.ld_rowadd % 4 || (uintptr_t)g.rowadd % 16)) return false;
    if (g.nseg > 1 && g.seg_width % 64) return false;  // a wave's 64-column chunk lies inside one segment
    if (g.rowT > 1 && g.rowT < 32) return false;         // the epilogue steps 8 rows at a time with at most one clip wrap
    const int es = g.out_dtype == DIMX_BF16 ? 2 : 4;
    for (int i = 0; i < g.nseg; ++i) {
        const OutSeg& s = g.seg[i];
        if (s.sd != 1 || s.D % 4 || s.sb % 4 || s.st % 4 || s.sh % 4 || ((uintptr_t)s.ptr % (4 * es))) return false;
    }
    return true;
}

static int launch_big(const GemmArgs& g, const OutSeg* segs, int nseg, int seg_width, hipStream_t s) {
    Big a;
    memset(&a, 0, sizeof(a));
    a.A = (const bf16*)g.A;
    a.W = (const bf16*)g.W;
    a.lda = g.lda;
    a.ldw = g.ldw;
    a.M = g.M;
    a.N = g.N;
    a.K = g.K;
    a.bias = g.bias;
    a.act = g.act;
    a.residual = g.residual;
    a.ldr = g.ldr;
    a.rowadd = g.rowadd;
    a.rowadd_mode = g.rowadd_mode;
    a.rowadd_scale = g.rowadd_scale;
    a.rowadd_off = g.rowadd_off;
    a.ld_rowadd = g.ld_rowadd;
    a.rowadd_div = g.rowadd_div < 1 ? 1 : g.rowadd_div;
    a.rowT = g.rowT;
    DIMX_REQUIRE(nseg >= 1 && nseg <= 8, DIMX_ERR_ARG, "gemm256: %d output segments", nseg);
    for (int i = 0; i < nseg; ++i) a.seg[i] = segs[i];
    a.nseg = nseg;
    a.seg_width = seg_width;
    a.tiles_m = ceil_div(g.M, 256);
    a.tiles_n = ceil_div(g.N, 256);
    a.nkt = g.K / 64;
    // column tiles swept before the row tile advances: the group's share of W (n_group x 256 x K bf16) should stay in the
    // XCD's 4 MB L2 next to the A panels streaming through; groups are balanced
    int ng = (int)((size_t)(3840u << 10) / ((size_t)256 * g.K * 2));
    ng = ng < 1 ? 1 : (ng > a.tiles_n ? a.tiles_n : ng);
    const int ngroups = ceil_div(a.tiles_n, ng);
    a.n_group = ceil_div(a.tiles_n, ngroups);
    static const int ng_env = getenv("DIMX_G256_NGROUP") ? atoi(getenv("DIMX_G256_NGROUP")) : 0;  // tuning
    if (ng_env > 0) a.n_group = ng_env > a.tiles_n ? a.tiles_n : ng_env;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        DIMX_HIP(hipGetDevice(&dev));
        DIMX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int grid = cus / 8 * 8;
    DIMX_REQUIRE(grid >= 8, DIMX_ERR_ARG, "gemm256: device with %d CUs", cus);
    {
        // start-delay spans (see the kernel): one tile time ~ nkt x 2900 shader cycles (measured: 2.5 us per k-tile at
        // ~1.8 GHz with the epilogue bursts, ~1.6 us without).  Blocks with a spare slot: a full tile time.  When every
        // block has the same number of tiles nobody has slack: a quarter tile over all blocks still pays from ~8 tiles up.
        const int tt = a.tiles_m * a.tiles_n, nbx = grid / 8;
        int nt_max = 0, nt_min = 1 << 30;
        for (int x8 = 0; x8 < 8; ++x8) {
            const int cnt = (int)((long)tt * (x8 + 1) / 8) - (int)((long)tt * x8 / 8);
            const int hi = ceil_div(cnt, nbx), lo = cnt / nbx;
            nt_max = hi > nt_max ? hi : nt_max;
            nt_min = lo < nt_min ? lo : nt_min;
        }
        a.nt_max = nt_max;
        static const char* de = getenv("DIMX_G256_DESYNC");  // "slack_permille,full_permille"
        int ps = 0, pf = 0;  // measured (round 3): any start delay costs more than it returns -- the epilogue is bound by the CU's own store path, not by a chip-wide write burst
        if (de) {
            ps = atoi(de);
            const char* c = strchr(de, ',');
            pf = c ? atoi(c + 1) : 0;
        }
        const double tile_cycles = (double)a.nkt * 2900.0;
        a.desync_slack = (unsigned)(tile_cycles * ps / 1000.0);
        a.desync_full = (unsigned)(tile_cycles * pf / 1000.0);
        static const int abl_env = getenv("DIMX_G256_ABL") ? atoi(getenv("DIMX_G256_ABL")) : 0;
        a.abl = abl_env;
    }
    // tuning (tools/g256_var.py): DMA placement variant and the in-kernel interval profile, bf16 / no-activation instantiation only
    static const int var_env = getenv("DIMX_G256_VAR") ? atoi(getenv("DIMX_G256_VAR")) : kDefaultVar;
    static const bool prof_env = getenv("DIMX_G256_PROF") != nullptr;
    static unsigned long long* prof_buf = nullptr;
    const bool plain = !g.bias && !g.residual && !g.rowadd_mode;  // epilogue without bias / positional rows / residual
    const bool tunable = g.out_dtype == DIMX_BF16 && g.act == ACT_NONE && plain;
    const int var = tunable ? var_env : kDefaultVar;
    const bool prof = tunable && prof_env;
    if (prof && !prof_buf) DIMX_HIP(hipMalloc((void**)&prof_buf, 64 * sizeof(unsigned long long)));
    a.prof = prof ? prof_buf : nullptr;
#define G256_(OT, AC, PL, VR, PF)                                                                                           \
    do {                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<OT, AC, PL, VR, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal); \
        hipLaunchKernelGGL((gemm256_kernel<OT, AC, PL, VR, PF>), dim3(grid), dim3(512), kLdsTotal, s, a);                   \
    } while (0)
#define G256P2_(OT, AC, PL)                                                                                             \
    do {                                                                                                                \
        (void)hipFuncSetAttribute((const void*)gemm256p2_kernel<OT, AC, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal); \
        hipLaunchKernelGGL((gemm256p2_kernel<OT, AC, PL>), dim3(grid), dim3(512), kLdsTotal, s, a);                     \
    } while (0)
#define G256P2(OT, AC)                                                     \
    do {                                                                   \
        if (plain) G256P2_(OT, AC, true); else G256P2_(OT, AC, false);     \
    } while (0)
#define G256P2_ACT(OT)                                           \
    do {                                                         \
        switch (g.act) {                                         \
            case ACT_LEAKY: G256P2(OT, ACT_LEAKY); break;        \
            case ACT_GELU_TANH: G256P2(OT, ACT_GELU_TANH); break;\
            case ACT_GELU_ERF: G256P2(OT, ACT_GELU_ERF); break;  \
            default: G256P2(OT, ACT_NONE); break;                \
        }                                                        \
    } while (0)
    if (tunable && (var != kDefaultVar || prof)) {  // the four-phase loop's forms: A/B runs and the in-kernel profile
        switch (var * 2 + (prof ? 1 : 0)) {
            case 0: G256_(bf16, ACT_NONE, true, 0, false); break;
            case 1: G256_(bf16, ACT_NONE, true, 0, true); break;
            case 2: G256_(bf16, ACT_NONE, true, 1, false); break;
            case 3: G256_(bf16, ACT_NONE, true, 1, true); break;
            case 4: G256_(bf16, ACT_NONE, true, 2, false); break;
            case 5: G256_(bf16, ACT_NONE, true, 2, true); break;
            case 6: G256_(bf16, ACT_NONE, true, 3, false); break;
            case 7: G256_(bf16, ACT_NONE, true, 3, true); break;
            case 9: G256_(bf16, ACT_NONE, true, 3, true); break;  // DIMX_G256_PROF without a VAR: the four-phase loop's stamps
            default: DIMX_REQUIRE(false, DIMX_ERR_ARG, "gemm256: DIMX_G256_VAR=%d", var);
        }
    } else if (g.out_dtype == DIMX_BF16) {
        G256P2_ACT(bf16);
    } else {
        G256P2_ACT(float);
    }
#undef G256P2_ACT
#undef G256P2
#undef G256P2_
#undef G256_
    if (prof) {  // tuning only: synchronous read-back of block 0's interval sums (waves 0 and 4)
        unsigned long long hst[64];
        DIMX_HIP(hipStreamSynchronize(s));
        DIMX_HIP(hipMemcpy(hst, prof_buf, sizeof(hst), hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; ++w) {
            const double n = (double)hst[w * 32 + 17];
            fprintf(stderr, "g256prof var %d wave %d ktiles %.0f :", var, w * 4, n);
            for (int i = 0; i < 12; ++i) fprintf(stderr, " %.0f", (double)hst[w * 32 + i] / (n > 0 ? n : 1));
            fprintf(stderr, " | reads done");
            for (int i = 13; i < 17; ++i) fprintf(stderr, " %.0f", (double)hst[w * 32 + i] / (n > 0 ? n : 1));
            fprintf(stderr, " | epilogue per tile %.0f\n", (double)hst[w * 32 + 12] / (n > 0 ? n / a.nkt : 1));
        }
    }
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_gemm256(const GemmArgs& g, hipStream_t s) { return launch_big(g, g.seg, g.nseg, g.seg_width, s); }

// the same GEMM with up to 8 equal-width column segments (the decoder layers' cross-attention K | V caches in one launch)
int launch_gemm256_segs(const GemmArgs& g, const OutSeg* segs, int nseg, int seg_width, hipStream_t s) {
    DIMX_REQUIRE(seg_width % 64 == 0 && seg_width * nseg == g.N, DIMX_ERR_ARG, "gemm256_segs: bad segment width");
    for (int i = 0; i < nseg; ++i)
        DIMX_REQUIRE(segs[i].sd == 1 && segs[i].D % 8 == 0 && ((uintptr_t)segs[i].ptr % 16) == 0, DIMX_ERR_ARG,
                     "gemm256_segs: segment %d is not row-contiguous", i);
    return launch_big(g, segs, nseg, seg_width, s);
}

}  // namespace dimx
