// gemm_x3.hip (round 6) -- the decode GEMMs of the f32 PARITY mode on the bf16 matrix cores, f32-equivalent.
//
// The parity mode computed every GEMM with v_mfma_f32_32x32x2_f32 -- exact f32 products at 1/16 of the bf16 MFMA rate: 25 launches per
// token, 149 of the mode's 391 ms per 256-clip batch (profiles/r05_parity_kernel_stats.csv).  gfx950 has no xf32.  But an f32 number is
// the EXACT sum of three bf16 numbers (8 + 8 + 8 significand bits, each plane the truncation of what the planes before it left), and
// a product of two bf16 numbers is exact in f32, so  a . w = sum over planes (a_i w_j)  with f32 accumulation is an f32 dot product
// whose only rounding is the accumulation's -- like any f32 GEMM.  The six leading cross products (i + j <= 2) are kept; the three
// dropped ones are below 2^-24 |a| |w| each, the size of ONE f32 rounding.  6/16 of the f32-MFMA time.
//   * W: three bf16 planes [3][N][ldw], split once when the weights are packed (split_x3_kernel below);
//   * A: f32 rows, fetched by LDS-DMA as they are and split in registers by the consumer waves (one wave per 32 x 32 output block,
//     fragment reads one k-tile ahead of the MFMAs);
//   * tile 128 x BN (BN = 36 / 64 / 72 / 96 columns = 2 or 3 MFMA column blocks, the last one partly valid), BK = 32 (a 128-byte
//     f32 row piece, the parity mode's k-tile), 8 or 12 consumer + 4 loader waves in the loop of gemm_ws72_kernel (4-slot LDS-DMA ring, counted
//     vmcnt per loader, one s_barrier per k-tile); more than half a CU's LDS, so a block owns its CU;
//   * a W plane's k-tile row is 64 bytes: two W rows share a 128-byte LDS row (row n -> LDS row n / 2, half n % 2), chunk c of an LDS
//     row sits at slot c ^ ((row >> 1) & 7) -- the lanes of a fragment read (rows 2r, 2r + 1 -> chunks c, c + 4) cover all 64 banks once;
//   * split-K slabs in slab order (deterministic); tile and split count depend on (N, K, slabs or not) only, never on M: a rank's shard
//     of a batch reproduces the rows of the whole batch bit for bit, like the f32 MFMA kernel it replaces.
// Reference arithmetic: the fp32 Linear layers of the x-transformers decoder, code/seq2seq_pretrain.py:413-418, one step of
// AutoregressiveWrapper.generate (:450).
#include "common.hpp"

namespace dimx {
namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int OFF> __device__ __forceinline__ void ds_read128(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ int lds_off(int row, int kc) { return row * 128 + (((kc ^ (row >> 1)) & 7) << 4); }

// x = p0 + p1 + p2 exactly (bf16 planes by truncation); 8 values -> three bf16x8 fragments
__device__ __forceinline__ void x3_split8(const u32x4_t& lo, const u32x4_t& hi, u32x4_t& p0, u32x4_t& p1, u32x4_t& p2) {
    const unsigned x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned h1[8], h2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned t0 = x[i] & 0xffff0000u;
        const float r1 = __builtin_bit_cast(float, x[i]) - __builtin_bit_cast(float, t0);   // exact
        const unsigned u1 = __builtin_bit_cast(unsigned, r1);
        const unsigned t1 = u1 & 0xffff0000u;
        const float r2 = r1 - __builtin_bit_cast(float, t1);                                 // exact, <= 8 significand bits left
        h1[i] = u1;
        h2[i] = __builtin_bit_cast(unsigned, r2);
    }
    // the high halves of two values in one register (v_perm_b32: {lo.b2, lo.b3, hi.b2, hi.b3}); the truncation IS the byte select
    unsigned q0[4], q1[4], q2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q0[i] = __builtin_amdgcn_perm(x[2 * i + 1], x[2 * i], 0x07060302u);
        q1[i] = __builtin_amdgcn_perm(h1[2 * i + 1], h1[2 * i], 0x07060302u);
        q2[i] = __builtin_amdgcn_perm(h2[2 * i + 1], h2[2 * i], 0x07060302u);
    }
    p0 = u32x4_t{q0[0], q0[1], q0[2], q0[3]};
    p1 = u32x4_t{q1[0], q1[1], q1[2], q1[3]};
    p2 = u32x4_t{q2[0], q2[1], q2[2], q2[3]};
}

// the parity mode's activations: exact libm forms (the bf16 mode's fast forms live in gemm.hip)
__device__ __forceinline__ float x3_act(int act, float x) {
    if (act == ACT_LEAKY) return x > 0.f ? x : 0.2f * x;
    if (act == ACT_GELU_TANH) return x * (0.5f * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * (x * x * x)))));
    if (act == ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
    return x;
}

template <int NCB, int BN, int STAGES = 4>
struct X3 {
    static constexpr int BM = 128, BK = 32, NCONS = 4 * NCB, THREADS = (NCONS + 4) * 64;
    static constexpr int A_BYTES = BM * 128;           // f32 rows of 32 k
    static constexpr int P_BYTES = NCB * 16 * 128;     // one W plane: NCB * 32 rows of 64 B, two per LDS row
    static constexpr int SLOT_BYTES = A_BYTES + 3 * P_BYTES;
    static constexpr int WP = (BN + 15) / 16;          // LDS-DMA pieces per plane (16 W rows = 8 LDS rows = 1 KiB)
    static constexpr int NWP = 3 * WP;
    static constexpr int RING_BYTES = STAGES * SLOT_BYTES;
    static constexpr int SMEM_BYTES = RING_BYTES > 84 * 1024 ? RING_BYTES : 84 * 1024;   // a block owns its CU
    static_assert(WP * 8 <= NCB * 16 && BN <= NCB * 32 && RING_BYTES <= 160 * 1024 - 1024, "tile");

    template <int LW>  // W pieces of this loader wave per k-tile
    static __device__ __forceinline__ void loader(const GemmArgs& a, int lw, int lane, int m0, int n0, int kt0, int nk, unsigned char* smem) {
        constexpr int LA = 4, LPT = LA + LW;
        static_assert((STAGES - 2) * LPT < 56, "ring depth");
        const float* __restrict__ A = (const float*)a.A;
        const bf16* __restrict__ W3 = (const bf16*)a.w3;
        const float* gA[LA];
        const bf16* gW[LW > 0 ? LW : 1];
        int oW[LW > 0 ? LW : 1];
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int row = (lw * LA + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            gA[j] = A + (size_t)m * a.lda + c * 4 + (size_t)kt0 * BK;
        }
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int q = lw + 4 * j, plane = q / WP, piece = q - plane * WP;
            const int lrow = piece * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);
            int n = n0 + 2 * lrow + (chunk >> 2);
            n = n < a.N ? n : a.N - 1;
            gW[j] = W3 + (size_t)plane * a.w3_plane + (size_t)n * a.ldw + (chunk & 3) * 8 + (size_t)kt0 * BK;
            oW[j] = A_BYTES + plane * P_BYTES + piece * 1024;
        }
        auto issue = [&](int kt, int slot) {
            unsigned char* base = smem + slot * SLOT_BYTES;
#pragma unroll
            for (int j = 0; j < LA; ++j)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gA[j] + (size_t)kt * BK), (lds_void_t*)(base + (lw * LA + j) * 1024), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < LW; ++j)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gW[j] + (size_t)kt * BK), (lds_void_t*)(base + oW[j]), 16, 0, 0);
        };
        int issued = 0;
        for (; issued < STAGES - 1 && issued < nk; ++issued) issue(issued, issued);
        int slot_next = issued % STAGES;
        for (int st = 0; st < nk; ++st) {
            const int y = issued - (st + 1);  // k-tiles issued behind slot st: 0 .. STAGES - 2
            if (y <= 0) wait_vmcnt<0>();
            else if (y == 1) wait_vmcnt<LPT>();
            else if (y == 2 || STAGES < 5) wait_vmcnt<2 * LPT>();
            else wait_vmcnt<3 * LPT>();
            __builtin_amdgcn_s_barrier();
            if (issued < nk && !(a.tile_map & 1)) {   // tile_map: ablation bits of tools (DIMX_X3_ABL; results are then wrong)
                issue(issued, slot_next);
                ++issued;
                slot_next = slot_next + 1 == STAGES ? 0 : slot_next + 1;
            }
        }
    }

    static __device__ __forceinline__ void body(const GemmArgs& a, int block_id, int nblocks, unsigned char* smem) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int tiles_n = a.N / BN, tiles_m = (a.M + BM - 1) / BM;
        const int ntiles = tiles_m * tiles_n;
        int bid = block_id;  // XCD-aware order as in gemm_ws72_kernel: an XCD's blocks are a contiguous range of n-major tiles
        {
            const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, i = bid >> 3;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        }
        const int split = bid / ntiles, tile = bid - split * ntiles;
        const int tile_n = tile / tiles_m, tile_m = tile - tile_n * tiles_m;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const int nk_all = a.ldw / BK;
        const int per = (nk_all + a.splitk - 1) / a.splitk;
        const int kt0 = split * per;
        int nk = nk_all - kt0;
        nk = nk > per ? per : nk;
        if (nk <= 0) return;
        if (wave >= NCONS) {
            const int lw = wave - NCONS;  // pieces q = lw, lw + 4, ... < NWP
            constexpr int L0 = (NWP + 3) / 4, L1 = (NWP + 2) / 4, L2 = (NWP + 1) / 4, L3 = NWP / 4;
            if (lw == 0) loader<L0>(a, lw, lane, m0, n0, kt0, nk, smem);
            else if (lw == 1) loader<L1>(a, lw, lane, m0, n0, kt0, nk, smem);
            else if (lw == 2) loader<L2>(a, lw, lane, m0, n0, kt0, nk, smem);
            else loader<L3>(a, lw, lane, m0, n0, kt0, nk, smem);
            return;
        }
        // ---------------- consumer wave w: row block w % 4 x column block w / 4 (NCB consumer waves per SIMD).
        // Forms measured on the decoder's shapes (profiles/r06_x3_gemm.txt): one wave per row block and all column blocks, without and
        // with the read-ahead below (ff1 32.6 / 33.8 us), this form without and with it (31.4 / 31.5 us): all within 7 % -- the loop is
        // bound by the consumers (no LDS-DMA in the loop: same time), and of their time the skeleton (barrier + fragment reads) is
        // 0.26 us per k-tile, split + MFMAs 0.5 us.
        const int rb = wave & 3, cb = wave >> 2;
        const int half = lane >> 5, l31 = lane & 31;
        const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
        unsigned aoff[2][2], woff[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            aoff[ks][0] = lds0 + lds_off(rb * 32 + l31, ks * 4 + half * 2);
            aoff[ks][1] = lds0 + lds_off(rb * 32 + l31, ks * 4 + half * 2 + 1);
            woff[ks] = lds0 + A_BYTES + cb * 2048 + lds_off(l31 >> 1, (l31 & 1) * 4 + ks * 2 + half);   // column block cb: 16 LDS rows in
        }
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // everything the epilogue needs is requested now (the consumers wait for the first tile anyway)
        const int n_lim = n0 + BN;   // N % BN == 0
        const int row_base = m0 + rb * 32 + 4 * half;
        const int mrem = a.M - row_base;
        const int ldc = (int)a.seg[0].st, ldr = a.ldr, act = a.act;
        const int ncol = n0 + cb * 32 + l31;
        const bool col_ok = ncol < n_lim;
        const int ncl = col_ok ? ncol : n_lim - 1;
        const float bias_v = (a.bias && split == 0) ? a.bias[ncl] : 0.f;
        float* cptr = (float*)a.seg[0].ptr + (a.out_slabs ? (size_t)split * a.slab_stride : 0) + (size_t)row_base * ldc + ncl;
        const float* rptr = a.residual ? a.residual + (size_t)row_base * ldr + ncl : nullptr;
        // The fragment reads of tile t + 1 are issued BEFORE tile t is computed (two register sets).  A consumer arrives at barrier
        // t + 1 with tile t's fragments in registers, so the loaders may refill tile t's slot right behind that barrier -- the
        // hand-off protocol of the loop is unchanged.
        struct Frag {
            u32x4_t a[2][2], w[2][3];
        };
        unsigned boff = 0;
        auto read_tile = [&](Frag& f) {   // A of both k-steps, then the W fragments (3 planes x 2 k-steps)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ds_read128<0>(f.a[ks][0], aoff[ks][0] + boff);
                ds_read128<0>(f.a[ks][1], aoff[ks][1] + boff);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < 3; ++p) ds_read128<0>(f.w[ks][p], woff[ks] + boff + (unsigned)(p * P_BYTES));
            boff = boff + SLOT_BYTES == STAGES * SLOT_BYTES ? 0u : boff + SLOT_BYTES;
        };
        auto landed = [&](Frag& f) {   // every fragment of the set passes through the wait: no use of them can move above it (a bare
                                       // s_waitcnt has no data dependence on what it waits for: hipcc hoisted the first uses above it)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[0][2]),
                           "+v"(f.w[1][0]), "+v"(f.w[1][1]), "+v"(f.w[1][2])
                         :
                         : "memory");
        };
        auto compute = [&](const Frag& f) {
            if (a.tile_map & 2) return;   // ablation: fragments read, nothing computed
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4_t p0, p1, p2;
                if (a.tile_map & 8) {   // ablation: no operand split (the raw fragment registers stand in for the planes; wrong results)
                    p0 = f.a[ks][0];
                    p1 = f.a[ks][1];
                    p2 = f.a[ks][0];
                } else {
                    x3_split8(f.a[ks][0], f.a[ks][1], p0, p1, p2);
                }
                const bf16x8_t A0 = __builtin_bit_cast(bf16x8_t, p0), A1 = __builtin_bit_cast(bf16x8_t, p1), A2 = __builtin_bit_cast(bf16x8_t, p2);
                // fixed order, smallest terms first: a2 w0, a1 w1, a0 w2 (2^-16), a1 w0, a0 w1 (2^-8), a0 w0
#define DIMX_X3_MMA(AP, WPL) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AP, __builtin_bit_cast(bf16x8_t, f.w[ks][WPL]), acc, 0, 0, 0)
                if (!(a.tile_map & 4)) {   // DIMX_X3_ABL=4 (measurement only, NOT a parity mode): the three 2^-16 terms dropped =
                                           // two bf16 planes per operand, 16 significand bits -- what does "half the MFMAs" cost in tokens?
                    DIMX_X3_MMA(A2, 0);
                    DIMX_X3_MMA(A1, 1);
                    DIMX_X3_MMA(A0, 2);
                }
                DIMX_X3_MMA(A1, 0);
                DIMX_X3_MMA(A0, 1);
                DIMX_X3_MMA(A0, 0);
#undef DIMX_X3_MMA
            }
        };
        Frag f0, f1;
        __builtin_amdgcn_s_barrier();   // tile 0
        read_tile(f0);
        int st = 0;
        for (; st + 1 < nk; st += 2) {
            landed(f0);
            __builtin_amdgcn_s_barrier();   // tile st + 1 (and: this wave holds tile st in registers)
            read_tile(f1);
            compute(f0);
            landed(f1);
            if (st + 2 < nk) {
                __builtin_amdgcn_s_barrier();   // tile st + 2
                read_tile(f0);
            }
            compute(f1);
        }
        if (st < nk) {
            landed(f0);
            compute(f0);
        }
        // ---------------- epilogue: bias, activation (exact), residual, plain row-major f32 destination or split-K slab
        if (col_ok) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = x3_act(act, acc[r] + bias_v);
            if (rptr) {
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int off = (r & 3) + 8 * (r >> 2);
                    off = off < mrem ? off : (mrem > 0 ? mrem - 1 : 0);
                    rv[r] = rptr[(size_t)off * ldr];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += rv[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = (r & 3) + 8 * (r >> 2);
                if (off < mrem) cptr[(size_t)off * ldc] = v[r];
            }
        }
    }
};

template <int NCB, int BN, int STAGES>
__global__ __launch_bounds__((X3<NCB, BN, STAGES>::THREADS)) void gemm_x3_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[X3<NCB, BN, STAGES>::SMEM_BYTES];
    X3<NCB, BN, STAGES>::body(a, blockIdx.x, gridDim.x, smem);
}

// W f32 -> three bf16 planes (truncation splits: p0 + p1 + p2 == w exactly)
__global__ __launch_bounds__(256) void split_x3_kernel(const float* __restrict__ w, bf16* __restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned x = __builtin_bit_cast(unsigned, w[i]);
    const unsigned t0 = x & 0xffff0000u;
    const float r1 = __builtin_bit_cast(float, x) - __builtin_bit_cast(float, t0);
    const unsigned u1 = __builtin_bit_cast(unsigned, r1), t1 = u1 & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, t1);
    p[i].x = (unsigned short)(x >> 16);
    p[n + i].x = (unsigned short)(u1 >> 16);
    p[2 * n + i].x = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
}

}  // namespace

// Column tile of the kernel for an N-wide projection and its K-split count at the design point M = 256 (2 row tiles): both depend on
// (N, K, slabs or not) only.  Cost = k-tiles per block x MFMA column blocks (+ half a k-tile per slab for its write and the
// consumer's read; partly valid column blocks lose a tie), blocks <= 256, at most 8 slabs (the consumers' limit).
bool gemm_x3_plan(const GemmArgs& a, int* bn_out, int* sp_out) {
    static const int cand[4][2] = {{96, 3}, {72, 3}, {64, 2}, {36, 2}};
    const int nk = a.ldw / 32;
    float best = 1e30f;
    int bbn = 0, bsp = 1;
    for (int c = 0; c < 4; ++c) {
        const int bn = cand[c][0], ncb = cand[c][1];
        if (a.N % bn != 0) continue;
        const int tiles = 2 * (a.N / bn);
        if (tiles > 256) continue;
        int lo = 1, hi = a.out_slabs ? 8 : 1;
        if (a.out_slabs && a.force_splitk > 0) lo = hi = a.force_splitk;
        for (int sp = lo; sp <= hi; ++sp) {
            if ((sp > lo && tiles * sp > 256) || sp > nk) break;
            const float cost = (float)(((nk + sp - 1) / sp) * ncb) + 0.5f * (float)sp + ((bn == 36 || bn == 72) ? 0.25f : 0.f);
            if (cost < best) {
                best = cost;
                bbn = bn;
                bsp = sp;
            }
        }
    }
    if (!bbn) return false;
    *bn_out = bbn;
    *sp_out = bsp;
    return true;
}

// the kernel takes: the decode step's GEMMs (whatever their M) with f32 operands, the three W planes at hand and a plain row-major f32 destination
bool gemm_use_x3(const GemmArgs& a) {
    static const bool off = getenv("DIMX_NO_X3") != nullptr;
    if (off || !a.w3 || !a.x3_decode || a.in_dtype != DIMX_F32 || a.out_dtype != DIMX_F32) return false;   // any M: see GemmArgs.x3_decode
    if (a.conv_T != 0 || a.kloop != 0 || a.K != a.ldw || a.K % 32 != 0 || a.force_simple || a.w_tiled || a.ln_stats || a.cfg != 0) return false;
    if (a.nseg != 1 || a.seg[0].sd != 1 || a.seg[0].sh != 0 || a.seg[0].sb != a.seg[0].st * (long)a.rowT || a.rowadd_mode != 0) return false;
    if (((uintptr_t)a.w3 % 16) != 0 || (a.w3_plane * 2) % 16 != 0) return false;
    int bn, sp;
    return gemm_x3_plan(a, &bn, &sp);
}

int launch_gemm_x3(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    int bn = 0, sp = 1;
    DIMX_REQUIRE(gemm_x3_plan(a, &bn, &sp), DIMX_ERR_ARG, "gemm(x3): no column tile for N=%d", a.N);
    a.splitk = sp;
    if (a.out_slabs) a.residual = nullptr;
    const int blocks = ceil_div(a.M, 128) * (a.N / bn) * sp;
    static const int abl = getenv("DIMX_X3_ABL") ? atoi(getenv("DIMX_X3_ABL")) : 0;       // tuning: 1 no DMA in the loop, 2 no compute, 4 three products only, 8 no operand split
    static const int stages = getenv("DIMX_X3_STAGES") ? atoi(getenv("DIMX_X3_STAGES")) : 4; // tuning: ring depth of the two-column-block tiles (5 measured no faster)
    a.tile_map = abl;
    switch (bn) {
        case 96: hipLaunchKernelGGL((gemm_x3_kernel<3, 96, 4>), dim3(blocks), dim3(1024), 0, s, a); break;
        case 72: hipLaunchKernelGGL((gemm_x3_kernel<3, 72, 4>), dim3(blocks), dim3(1024), 0, s, a); break;
        case 64:
            if (stages == 5) hipLaunchKernelGGL((gemm_x3_kernel<2, 64, 5>), dim3(blocks), dim3(768), 0, s, a);
            else hipLaunchKernelGGL((gemm_x3_kernel<2, 64, 4>), dim3(blocks), dim3(768), 0, s, a);
            break;
        default:
            if (stages == 5) hipLaunchKernelGGL((gemm_x3_kernel<2, 36, 5>), dim3(blocks), dim3(768), 0, s, a);
            else hipLaunchKernelGGL((gemm_x3_kernel<2, 36, 4>), dim3(blocks), dim3(768), 0, s, a);
            break;
    }
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

int launch_split_x3(const float* w, void* planes, size_t n, hipStream_t s) {
    DIMX_REQUIRE(w && planes && n > 0, DIMX_ERR_ARG, "split_x3: null operand");
    hipLaunchKernelGGL(split_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, (bf16*)planes, n);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
