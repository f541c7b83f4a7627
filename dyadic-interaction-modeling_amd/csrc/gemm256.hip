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

constexpr int kHalf = 16384;        // one half-tile: 128 rows x 128 B
constexpr int kBuf = 4 * kHalf;     // one k-tile: A0h A1h B0h B1h
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
    int abl;  // tuning ablations (DIMX_G256_ABL): 1 no DMA in the loop, 2 no ds_reads, 4 no MFMA, 8 no setprio
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

// XCD-local tile q of XCD x -> (tile_m, tile_n); false when past the XCD's last tile.  Order: column-tile groups of
// n_group (the last one may be narrower); inside a group row tile major, column tile minor.
__device__ __forceinline__ bool tile_of(const Big& a, int x, int q, int& tm, int& tn) {
    const int tml = (a.tiles_m - x + 7) >> 3;  // row tiles of this XCD: tm = x, x + 8, ...
    if (tml <= 0) return false;
    const int per_group = tml * a.n_group;
    const int full = a.tiles_n / a.n_group;
    int gi, r, w;
    if (q < full * per_group) {
        gi = q / per_group;
        r = q - gi * per_group;
        w = a.n_group;
    } else {
        gi = full;
        r = q - full * per_group;
        w = a.tiles_n - full * a.n_group;
        if (w <= 0) return false;
    }
    const int ti = r / w;
    if (ti >= tml) return false;
    tm = ti * 8 + x;
    tn = gi * a.n_group + (r - ti * w);
    return true;
}

// Epilogue of one 32-row x 64-column chunk of a wave (both column halves of row block rb of row half mh), through a
// private 4 KiB LDS scratch: the accumulators arrive "one row per lane, 4 consecutive columns per register group" and
// leave as 16-byte row-contiguous pieces, 8 lanes per 128-byte line (a direct store of the accumulator layout
// touches 32 lines per instruction and measured 24 us per tile).  bf16 output: one pass over 32 x 64; f32 output: two
// passes over 32 x 32.  Bias / activation run before the transpose, positional rows / residual after it.
template <typename OutT, int ACT>
__device__ __forceinline__ void store_chunk(const Big& a, const f32x16_t& acc0, const f32x16_t& acc1, int mrow0, int ncol0,
                                            unsigned char* scratch, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    constexpr bool BF = sizeof(OutT) == 2;
    constexpr int PASSES = BF ? 1 : 2;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        // ---- write: row l31 of the chunk
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            if (!BF && nh != ps) continue;
            const f32x16_t& acc = nh == 0 ? acc0 : acc1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = ncol0 + nh * 32 + 8 * q + 4 * half;
                float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                if (a.bias) {
                    const float4 bv = *(const float4*)(a.bias + (n < a.N ? n : 0));
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                v.x = act256<ACT>(v.x); v.y = act256<ACT>(v.y); v.z = act256<ACT>(v.z); v.w = act256<ACT>(v.w);
                if (BF) {
                    const int c16 = nh * 4 + q;  // 16-byte chunk of the 128-byte row; this lane's half of it
                    *(uint2*)(scratch + l31 * 128 + ((c16 ^ (l31 & 7)) << 4) + half * 8) =
                        make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                } else {
                    const int c16 = 2 * q + half;
                    *(float4*)(scratch + l31 * 128 + ((c16 ^ (l31 & 7)) << 4)) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave reads it back: no barrier needed
        // ---- read back row-contiguous: 8 lanes per row, 4 row groups
        const int c16 = lane & 7;
        const int n = BF ? ncol0 + c16 * 8 : ncol0 + ps * 32 + c16 * 4;
        int s = 0, nn = n < a.N ? n : 0;
        if (a.nseg > 1) {
            s = nn / a.seg_width;
            nn -= s * a.seg_width;
        }
        const OutSeg sg = a.seg[s];
        const int hh = nn / sg.D, dd = nn - hh * sg.D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (lane >> 3) + 8 * i;
            const uint4 raw = *(const uint4*)(scratch + row * 128 + ((c16 ^ (row & 7)) << 4));
            const int m = mrow0 + row;
            if (m >= a.M || n >= a.N) continue;
            int b = m, t = 0;
            if (a.rowT > 1) {
                b = m / a.rowT;
                t = m - b * a.rowT;
            }
            OutT* p = (OutT*)sg.ptr + (long)b * sg.sb + (long)t * sg.st + (long)hh * sg.sh + dd;
            if (BF) {
                uint4 o = raw;
                if (a.rowadd_mode || a.residual) {  // rare for bf16 destinations: unpack, add, repack
                    float f[8];
                    const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[2 * e] = __builtin_bit_cast(float, w4[e] << 16);
                        f[2 * e + 1] = __builtin_bit_cast(float, w4[e] & 0xffff0000u);
                    }
                    if (a.rowadd_mode) {
                        const int ri = a.rowadd_mode == 1 ? t : (a.rowadd_mode == 2 ? b / a.rowadd_div + a.rowadd_off : a.rowadd_off);
                        const float* pr = a.rowadd + (size_t)ri * a.ld_rowadd + n;
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += pr[e] * a.rowadd_scale;
                    }
                    if (a.residual) {
                        const float* pr = a.residual + (size_t)m * a.ldr + n;
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += pr[e];
                    }
                    o = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
                }
                *(uint4*)p = o;
            } else {
                float4 v = __builtin_bit_cast(float4, raw);
                if (a.rowadd_mode) {
                    const int ri = a.rowadd_mode == 1 ? t : (a.rowadd_mode == 2 ? b / a.rowadd_div + a.rowadd_off : a.rowadd_off);
                    const float4 q4 = *(const float4*)(a.rowadd + (size_t)ri * a.ld_rowadd + n);
                    v.x += q4.x * a.rowadd_scale; v.y += q4.y * a.rowadd_scale; v.z += q4.z * a.rowadd_scale; v.w += q4.w * a.rowadd_scale;
                }
                if (a.residual) {
                    const float4 q4 = *(const float4*)(a.residual + (size_t)m * a.ldr + n);
                    v.x += q4.x; v.y += q4.y; v.z += q4.z; v.w += q4.w;
                }
                *(float4*)p = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next pass / chunk overwrites
    }
}

template <typename OutT, int ACT>
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
    auto stage_a = [&](const Cur& c, int buf) {
        unsigned char* dst = smem + buf * kBuf + c.hA * kHalf + wave * 2048;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(a.A + c.a0), (lds_void_t*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(a.A + c.a1), (lds_void_t*)(dst + 1024), 16, 0, 0);
    };
    auto stage_b = [&](const Cur& c, int buf) {
        unsigned char* dst = smem + buf * kBuf + (2 + c.hB) * kHalf + wave * 2048;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(a.W + c.b0), (lds_void_t*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(a.W + c.b1), (lds_void_t*)(dst + 1024), 16, 0, 0);
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
    stage_a(c2, 0);
    stage_b(c2, 0);
    stage_b(c1, 0);
    stage_a(c1, 0);
    advance(c2);  // -> k-tile 1
    advance(c1);  // -> k-tile 1
    stage_a(c2, 1);
    stage_b(c2, 1);
    advance(c2);  // -> k-tile 2
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    raw_barrier();
    if (wr == 1) raw_barrier();  // waves 4-7 run half a phase behind

    u32x4_t fa[2][4], fb0[4], fb1[4];
    unsigned bo = 0;  // byte offset of the k-tile buffer being computed
    for (long g = 0; g < total; ++g) {
        const int buf = (int)(bo != 0);
        // ================= phase 1: quadrant (0,0) -- read A(row half 0) and W(col half 0); stage B1h of g + 1
        {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ds_read128<0>(fb0[ks], baddr[ks] + bo);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                ds_read128<0>(fa[0][ks], aaddr[ks] + bo);
                ds_read128<4096>(fa[1][ks], aaddr[ks] + bo);
            }
        }
        stage_b(c1, buf ^ 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                acc[0][0][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb0[ks]),
                                                                       __builtin_bit_cast(bf16x8_t, fa[rb][ks]), acc[0][0][rb], 0, 0, 0);
        if (kPrio) __builtin_amdgcn_s_setprio(0);
        raw_barrier();
        // ================= phase 2: quadrant (0,1) -- read W(col half 1); stage A1h of g + 1
        {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ds_read128<kHalf>(fb1[ks], baddr[ks] + bo);
        }
        stage_a(c1, buf ^ 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                acc[0][1][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb1[ks]),
                                                                       __builtin_bit_cast(bf16x8_t, fa[rb][ks]), acc[0][1][rb], 0, 0, 0);
        if (kPrio) __builtin_amdgcn_s_setprio(0);
        raw_barrier();
        advance(c1);
        // ================= phase 3: quadrant (1,1) -- read A(row half 1); stage A0h of g + 2
        {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                ds_read128<kHalf>(fa[0][ks], aaddr[ks] + bo);
                ds_read128<kHalf + 4096>(fa[1][ks], aaddr[ks] + bo);
            }
        }
        stage_a(c2, buf);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        raw_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                acc[1][1][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb1[ks]),
                                                                       __builtin_bit_cast(bf16x8_t, fa[rb][ks]), acc[1][1][rb], 0, 0, 0);
        if (kPrio) __builtin_amdgcn_s_setprio(0);
        raw_barrier();
        // ================= phase 4: quadrant (1,0) -- everything is in registers; stage B0h of g + 2
        stage_b(c2, buf);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        raw_barrier();
        if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                acc[1][0][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb0[ks]),
                                                                       __builtin_bit_cast(bf16x8_t, fa[rb][ks]), acc[1][0][rb], 0, 0, 0);
        if (kPrio) __builtin_amdgcn_s_setprio(0);
        raw_barrier();
        advance(c2);
        bo ^= (unsigned)kBuf;

        // ================= end of an output tile: store, clear, move on (no barriers in here)
        if (++cu_kt == nkt) {
            unsigned char* scratch = smem + kLds + wave * 4096;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    store_chunk<OutT, ACT>(a, acc[mh][0][rb], acc[mh][1][rb], cu_m0 + mh * 128 + wr * 64 + rb * 32,
                                           cu_n0 + wc * 64, scratch, lane);
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
        }
    }
    if (wr == 0) raw_barrier();  // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dummy stagings past the end must land before the LDS is released
}

}  // namespace

// Eligible: bf16 operands, K % 64 == 0, row-contiguous destinations, enough rows to fill the chip.
bool gemm256_eligible(const GemmArgs& g) {
    static const bool off = getenv("DIMX_NO_G256") != nullptr;
    if (off || g.in_dtype != DIMX_BF16 || g.conv_T != 0 || g.out_slabs || g.force_simple) return false;
    const int kext = g.kloop ? g.kloop : g.ldw;
    if (g.K % 64 != 0 || g.K != kext || g.M < 4096 || g.N < 256 || g.N % 8 != 0) return false;
    // measured (tools/bench_prefill.py): ahead of the 128 x 128 kernel from K = 768 and N = 1024 up; short K or narrow N
    // (one and a half column tiles at N = 384) leave it behind
    static const bool all = getenv("DIMX_G256_ALL") != nullptr;
    if (!all && (g.K < 768 || g.N < 1024)) return false;
    if ((size_t)g.M * g.lda >= (1ull << 32) || (size_t)g.N * g.ldw >= (1ull << 32)) return false;  // 32-bit element offsets
    if (g.bias && ((uintptr_t)g.bias % 16)) return false;
    if (g.residual && (g.ldr % 4 || (uintptr_t)g.residual % 16)) return false;
    if (g.rowadd_mode && (g.ld_rowadd % 4 || (uintptr_t)g.rowadd % 16)) return false;
    if (g.nseg > 1 && g.seg_width % 4) return false;
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
#define G256(OT, AC)                                                                                              \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<OT, AC>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal); \
        hipLaunchKernelGGL((gemm256_kernel<OT, AC>), dim3(grid), dim3(512), kLdsTotal, s, a);                     \
    } while (0)
#define G256_ACT(OT)                                                 \
    do {                                                             \
        switch (g.act) {                                             \
            case ACT_LEAKY: G256(OT, ACT_LEAKY); break;              \
            case ACT_GELU_TANH: G256(OT, ACT_GELU_TANH); break;      \
            case ACT_GELU_ERF: G256(OT, ACT_GELU_ERF); break;        \
            default: G256(OT, ACT_NONE); break;                      \
        }                                                            \
    } while (0)
    if (g.out_dtype == DIMX_BF16) G256_ACT(bf16); else G256_ACT(float);
#undef G256_ACT
#undef G256
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
