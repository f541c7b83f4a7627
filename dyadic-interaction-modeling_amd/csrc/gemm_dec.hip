// gemm_dec.hip -- the decode step's chip-wide GEMMs (M = clips <= 256): C = A[M,K] . W[N,K]^T, bf16 operands, f32 accumulate.
//
// These are the three projections of a decoder layer that are too large for the XCD-local chain kernels: the fused q/k/v
// projection, and the two feed-forward projections (x-transformers Decoder, constructed at reference
// code/seq2seq_pretrain.py:413-419; one step of AutoregressiveWrapper.generate, :450): 12 launches per generated token.
//
// What round 5 measured on the kernel this one replaces (gemm_ws_kernel / gemm_ws72_kernel, in-kernel stamps,
// profiles/r05_gemm_ws72.txt): start 0.5 us, first k-tile 1.1 us later, then 0.31 us per k-tile x 18, epilogue 1.4-1.8 us.
// The k-tile time is a chain, not a rate: loader waves issue 4-5 LDS-DMA pieces (60-100 cycles each), wait for them, meet
// the consumers at a barrier; the consumers read 8 fragments from LDS, run 4 DEPENDENT MFMAs, and come back.  17 KiB per
// k-tile and CU is 270 cycles of the CU's vector-memory path; the loop took 650.
//
// This kernel takes the chain apart:
//   * W never touches LDS.  The weights are static, so they are packed once (pack_w_frag_kernel) into MFMA fragment order:
//     a consumer wave's B operands of one k-tile are 2 x 1 KiB contiguous, loaded straight into registers with a ring of R
//     k-tiles in flight per wave -- no barrier, no LDS-DMA issue slot, no LDS read for half of the operand bytes, and the
//     HBM latency of the weight stream is covered by the ring, not by the block's first barrier.
//   * Only A (the clips' rows, written by the previous kernel) goes through LDS: 8 pieces per k-tile, 2 per loader wave,
//     a ring of STAGES slots (one block per CU owns the LDS: 11 k-tiles ahead).
//   * Six consumer waves = 3 column blocks x 2 k-halves: a wave runs 4 MFMAs per k-tile on TWO independent accumulators
//     (rows 0-31 and 32-63) over k-steps {0,1} or {2,3} of the tile; the two k-halves meet once, through LDS, after the loop
//     (fixed order), and each then finishes one row half: bias / deferred-LayerNorm correction / GELU / store.
//   * 64 x 72 tiles: 4 x {64, 16, 32} tiles x {1, 4, 2} K splits = exactly 256 blocks for N = 4608 / 1152 / 2304, one per
//     CU (the LDS request makes a second block per CU impossible: see gemm_ws72_kernel).
//   * a lean argument block and epilogue: every scalar the epilogue needs is in registers before the main loop.
// Summation order: per output element, the k-steps {0,1} of all k-tiles in order, plus the k-steps {2,3} of all k-tiles in
// order -- independent of M and of the block's position, so a rank's shard of a batch reproduces the whole batch's rows.
#include "common.hpp"

namespace dimx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lds_off(int row, int kc) { return row * 128 + (((kc ^ (row >> 1)) & 7) << 4); }
__device__ __forceinline__ void ds_read128(u32x4_t& v, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); }
// W fragment loads are issued and waited for BY HAND: behind hipcc's own bookkeeping every use of a ring register waited vmcnt(0)
// (the ring was one k-tile deep in effect); the asm loads are invisible to that pass, the counted waits below are exact
template <int OFF> __device__ __forceinline__ void gload128(u32x4_t& v, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "i"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }

// Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7), v_exp_f32 / v_rcp_f32: the same erf-GELU as gemm.hip's FAST epilogue
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.7071067811865476f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float er = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + (x < 0.f ? -er : er));
}

struct DecArgs {
    const bf16* A;
    const unsigned char* Wf;  // fragment-packed weights (pack_w_frag_kernel)
    void* C;
    const float* bias;
    const float* ln_stats;    // deferred LayerNorm of the A rows (GemmArgs.ln_stats), or null
    const float* ln_colsum;
    unsigned* ln_err;
    unsigned long long* prof;
    long slab_stride;         // elements between split-K slabs (0: one slab)
    int lda, ldc;
    int M, N, nkt_all, splitk;
    int act;                  // ACT_NONE or ACT_GELU_ERF
    int ln_C;
    int dbg;                  // DIMX_DEC_ABL (tuning only, wrong results): 1 no W refills, 2 no A refills, 4 no LDS reads / MFMAs
};

constexpr int BM = 64, BN = 72, BK = 64;
constexpr int NCONS = 8, NLOAD = 4, NTHREADS = (NCONS + NLOAD) * 64;
constexpr int A_TILE = BM * 128;  // bytes of one k-tile of the A panel in LDS

// wait until at most 2 * n of this wave's vector-memory operations are outstanding (immediate operands only)
__device__ __forceinline__ void wait_vm_pairs(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<2>(); break;
        case 2: wait_vmcnt<4>(); break;
        case 3: wait_vmcnt<6>(); break;
        case 4: wait_vmcnt<8>(); break;
        case 5: wait_vmcnt<10>(); break;
        case 6: wait_vmcnt<12>(); break;
        case 7: wait_vmcnt<14>(); break;
        case 8: wait_vmcnt<16>(); break;
        case 9: wait_vmcnt<18>(); break;
        case 10: wait_vmcnt<20>(); break;
        case 11: wait_vmcnt<22>(); break;
        default: wait_vmcnt<24>(); break;
    }
}

// S ring slots of KT k-tiles each (the A panel in LDS; one barrier per slot), R k-tiles of W fragments in registers per wave.
template <typename OutT, int S, int KT, int R, bool PROF>
__global__ __launch_bounds__(NTHREADS) void gemm_dec_kernel(const DecArgs a) {
    static_assert(S >= 3 && KT >= 1 && (S - 1) * KT <= 12 && R >= 2 && R <= 8 && R % KT == 0, "ring depths");
    constexpr int SLOT = KT * A_TILE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * SLOT];
    __shared__ __attribute__((aligned(16))) float ln_sm[128];  // deferred LayerNorm: mean[64], rstd[64] of the tile's rows
    static_assert(S * SLOT >= 84 * 1024, "two blocks must not fit a CU (see gemm_ws72_kernel)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = a.N / BN, tiles_m = (a.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int bid = blockIdx.x;  // XCD-aware order: an XCD's blocks are a contiguous range of n-major tiles (the row tiles of a column
                           // tile share its weight fragments through that XCD's L2)
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int split = bid / ntiles, tile = bid - split * ntiles;
    const int tile_n = tile / tiles_m, tile_m = tile - tile_n * tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int per = (a.nkt_all + a.splitk - 1) / a.splitk;
    const int kt0 = split * per;
    int nk = a.nkt_all - kt0;
    nk = nk > per ? per : nk;
    if (nk <= 0) return;
    if (a.dbg & 32) return;          // ablation: the empty launch
    if (a.dbg & 8) nk = 1;           // ablation: one k-tile (prologue + epilogue only)
    const int nst = (nk + KT - 1) / KT;  // slots' worth of k-tiles ("super-tiles"); the last one may be partial
    auto stamp = [&](int i) {
        if (PROF && lane == 0 && (wave == 0 || wave == NCONS) && i < 32)
            a.prof[(size_t)blockIdx.x * 64 + (wave == NCONS ? 32 : 0) + i] = wall_clock64();
    };
    stamp(0);

    if (wave >= NCONS) {
        // ---------------- loader waves: the A panel, S - 1 slots ahead of the consumers; 2 pieces per k-tile and wave
        const int lw = wave - NCONS;
        const bf16* gA[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (lw * 2 + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            gA[j] = a.A + (size_t)m * a.lda + c * 8 + (size_t)kt0 * BK;
        }
        auto issue = [&](int st, int slot) {
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int kt = st * KT + j;
                if (kt < nk) {
                    unsigned char* base = smem + slot * SLOT + j * A_TILE + lw * 2048;
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(gA[0] + (size_t)kt * BK), (lds_void_t*)base, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(gA[1] + (size_t)kt * BK), (lds_void_t*)(base + 1024), 16, 0, 0);
                }
            }
        };
        // deferred LayerNorm: loader lw reduces the 32 partial sums of rows lw * 16 + (lane & 15) (gemm_ws_kernel's scheme: the
        // loads are this wave's oldest memory operations, the reduction runs when it has nothing left to issue)
        float2 lp[8];
        if (a.ln_stats) {
            int m = m0 + lw * 16 + (lane & 15);
            m = m < a.M ? m : a.M - 1;
            const float2* sp = (const float2*)a.ln_stats + ((size_t)(m >> 5) * 32 + (lane >> 4) * 8) * 32 + (m & 31);
#pragma unroll
            for (int i = 0; i < 8; ++i) lp[i] = sp[(size_t)i * 32];
        }
        int issued = 0;  // slots issued so far
        for (; issued < S - 1 && issued < nst; ++issued) issue(issued, issued);
        stamp(1);
        int slot_next = issued % S;
        const int st_ln = nst >= 2 ? nst - 2 : 0;
        for (int st = 0; st < nst; ++st) {
            if (st < 12) stamp(2 + 2 * st);
            {   // k-tiles issued behind slot st: each is two of this wave's operations
                int done = (st + 1) * KT, all = issued * KT;
                done = done < nk ? done : nk;
                all = all < nk ? all : nk;
                if (a.dbg & 2) wait_vmcnt<0>(); else wait_vm_pairs(all - done);
            }
            if (st < 12) stamp(3 + 2 * st);
            __builtin_amdgcn_s_barrier();
            if (issued < nst) {
                if (!(a.dbg & 2)) issue(issued, slot_next);
                ++issued;
                slot_next = slot_next + 1 == S ? 0 : slot_next + 1;
            }
            if (a.ln_stats && st == st_ln) {
                // {sum x, M2} of 32 column slices -> mean and variance by the parallel-variance formula, two passes
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s1 += lp[i].x;
                s1 += xor_lane_f32<16>(s1);
                s1 += xor_lane_f32<32>(s1);
                const float inv_c = 1.0f / (float)a.ln_C;
                const float mean = s1 * inv_c;
                const float ncs = (float)a.ln_C * (1.0f / 32.0f), inv_n = 32.0f * inv_c;
                float s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = lp[i].x * inv_n - mean;
                    s2 += lp[i].y + ncs * d * d;
                }
                s2 += xor_lane_f32<16>(s2);
                s2 += xor_lane_f32<32>(s2);
                const float var = s2 * inv_c;
                const float rstd = rsqrtf(var + 1e-5f);
                if (a.ln_err && lane < 16 && m0 + lw * 16 + lane < a.M && mean * mean > 64.0f * var)
                    atomicOr(a.ln_err, 4u);  // bf16(x) un-normalised is too coarse for this row (see chain.hip)
                if (lane < 16) {
                    // written with inline asm: behind a C++ LDS store hipcc would drain vmcnt(0) (LDS-DMA in flight)
                    const unsigned am = (unsigned)(size_t)(lds_void_t*)(ln_sm + lw * 16 + lane);
                    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:256" ::"v"(am), "v"(mean), "v"(rstd) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in LDS before this wave arrives at the next barrier
            }
        }
        stamp(28);
        __builtin_amdgcn_s_barrier();  // the consumers' exchange of the two k-halves (below)
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ---------------- consumer waves: (column block wn of the first two, k-step ks of every k-tile).  A wave's MFMAs per k-tile:
    // rows 0-31 and 32-63 against its own block's fragment, and rows 32 wn .. against the THIRD block's fragment of the same
    // k-step (the block with 8 valid columns is shared out: 3 MFMAs per wave and k-tile on three independent accumulators, two
    // consumer waves per SIMD -- six MFMAs per SIMD and k-tile, the balanced optimum for a 64 x 96 MFMA tile)
    const int wn = wave & 1, ks = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    unsigned aoff[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) aoff[mi] = lds0 + lds_off(mi * 32 + l31, 2 * ks + half);
    // everything the epilogue needs, now (kernel arguments, the lane's bias / column sums): nothing left to fetch later.
    // After the loop wave (wn, q = ks) finishes: block wn, row block q >> 1, register groups 2 (q & 1), 2 (q & 1) + 1 (8 values);
    // block 2, row block wn, register group q (4 values, lanes of the 8 valid columns only)
    const int q = ks;
    const int n_own = n0 + wn * 32 + l31;                 // always inside the tile
    const bool b2_ok = l31 < BN - 64;
    const int n_b2 = n0 + 64 + (b2_ok ? l31 : 0);
    const float bias_own = (a.bias && split == 0) ? a.bias[n_own] : 0.f;
    const float bias_b2 = (a.bias && split == 0) ? a.bias[n_b2] : 0.f;
    const float cs_own = a.ln_stats ? a.ln_colsum[n_own] : 0.f;
    const float cs_b2 = a.ln_stats ? a.ln_colsum[n_b2] : 0.f;
    const int act = a.act;
    const int ldc = a.ldc;
    const int row_own = m0 + (q >> 1) * 32 + (q & 1) * 16 + 4 * half;  // rows row_own + {0..3, 8..11}
    const int row_b2 = m0 + wn * 32 + 8 * q + 4 * half;                // rows row_b2 + {0..3}
    OutT* cbase = (OutT*)a.C + (size_t)split * a.slab_stride;
    OutT* c_own = cbase + (size_t)row_own * ldc + n_own;
    OutT* c_b2 = cbase + (size_t)row_b2 * ldc + n_b2;
    const int M = a.M;
    const bool has_ln = a.ln_stats != nullptr;
    // weight fragments: (column tile, column block, k-tile) = 4 KiB; k-step ks of it at + ks KiB; lane l at + 16 l
    const unsigned char* wbase = a.Wf + ((size_t)(tile_n * 3) * a.nkt_all + kt0) * 4096 + ks * 1024;
    const unsigned char* wp_own = wbase + (size_t)wn * a.nkt_all * 4096 + lane * 16;
    // (lanes of the third block beyond the tile's 72 columns re-read a valid lane's fragment: every lane always loads, the wave's
    // vmcnt arithmetic never depends on a lane mask, and those accumulator columns are never stored)
    const unsigned char* wp_b2 = wbase + (size_t)2 * a.nkt_all * 4096 + (b2_ok ? lane : (lane & 0x27)) * 16;
    u32x4_t wr[R][2];  // [ring slot][own block, third block]
    auto wload = [&](int r, int it) {
        gload128<0>(wr[r][0], wp_own + (size_t)it * 4096);
        gload128<0>(wr[r][1], wp_b2 + (size_t)it * 4096);
    };
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r < nk) wload(r, r);
    f32x16_t acc[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = acc[2][r] = 0.f;
    unsigned soff = 0;  // LDS offset of the current slot
    u32x4_t fa[2][2];   // [buffer][row block]
    auto aread = [&](int buf, unsigned off) {
        ds_read128(fa[buf][0], aoff[0] + off);
        ds_read128(fa[buf][1], aoff[1] + off);
    };
    const bool skip = (a.dbg & 4) || ((a.dbg & 64) && wave >= 4);
    for (int it0 = 0; it0 < nk; it0 += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int it = it0 + r;
            const int j = r % KT;  // position inside the slot (R % KT == 0: static)
            if (it < nk) {
                if (j == 0) {
                    if (it / KT < 12) stamp(2 + 2 * (it / KT));
                    __builtin_amdgcn_s_barrier();  // the slot's k-tiles have landed (and the slot S - 1 back is free again)
                    if (it / KT < 12) stamp(3 + 2 * (it / KT));
                    if (!skip) aread(0, soff);
                }
                if (!skip) {
                    const bool more = j + 1 < KT && it + 1 < nk;  // the next k-tile of this slot: its fragments are requested before
                    if (more) aread((j + 1) & 1, soff + (j + 1) * A_TILE);  // this one's MFMAs start
                    // the wave's outstanding vector-memory operations are its W fragments only (two per k-tile, in order): tile `it`
                    // has landed once at most 2 * min(nk - 1 - it, R - 1) younger ones are in flight
                    if (a.dbg & 1) {
                        wait_vmcnt<0>();
                    } else {
                        const int rem = nk - 1 - it;
                        if (rem >= R - 1) wait_vmcnt<2 * (R - 1)>(); else wait_vm_pairs(rem);
                    }
                    if (more) wait_lgkm<2>(); else wait_lgkm<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    const int b = j & 1;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[b][0]), __builtin_bit_cast(bf16x8_t, wr[r][0]), acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[b][1]), __builtin_bit_cast(bf16x8_t, wr[r][0]), acc[1], 0, 0, 0);
                    if (wn == 0)
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[b][0]), __builtin_bit_cast(bf16x8_t, wr[r][1]), acc[2], 0, 0, 0);
                    else
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[b][1]), __builtin_bit_cast(bf16x8_t, wr[r][1]), acc[2], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);  // the refill below overwrites the registers these MFMAs read
                    if (it + R < nk && !(a.dbg & 1)) wload(r, it + R);
                }
                if (j == KT - 1) {
                    wait_lgkm<0>();  // every LDS read of the slot is complete before the next barrier frees it
                    soff = soff + SLOT == S * SLOT ? 0u : soff + SLOT;
                }
            }
        }
    }
    stamp(28);
    // ---- the four k-steps meet: every wave parks its 12 register groups (acc[0], acc[1], acc[2]: 4 groups of 4 registers each)
    // in LDS as [wave][group][lane] x 16 B, then picks up the groups it finishes from all four k-steps, summed in k-step order
    __builtin_amdgcn_s_barrier();  // every wave is done with the A ring the exchange area aliases
    {
        float* xo = (float*)smem + wave * 3072 + lane * 4;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(xo + (t * 4 + g) * 256) = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
    __builtin_amdgcn_s_barrier();
    float v[8], v2[4];
    {
        // own block: accumulator q >> 1, groups 2 (q & 1) and 2 (q & 1) + 1, from the waves (wn, k-step 0..3) = wave ids wn + 2 k
        const float* xi = (const float*)smem + wn * 3072 + lane * 4;
        const int g0 = (q >> 1) * 4 + 2 * (q & 1);
        float4 p0 = *(const float4*)(xi + (g0)*256), p1 = *(const float4*)(xi + (g0 + 1) * 256);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 o0 = *(const float4*)(xi + k * 6144 + g0 * 256), o1 = *(const float4*)(xi + k * 6144 + (g0 + 1) * 256);
            p0.x += o0.x; p0.y += o0.y; p0.z += o0.z; p0.w += o0.w;
            p1.x += o1.x; p1.y += o1.y; p1.z += o1.z; p1.w += o1.w;
        }
        v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w;
        v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
        // third block: accumulator 2 of the waves (wn, k-step 0..3), group q
        float4 r0 = *(const float4*)(xi + (8 + q) * 256);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 o = *(const float4*)(xi + k * 6144 + (8 + q) * 256);
            r0.x += o.x; r0.y += o.y; r0.z += o.z; r0.w += o.w;
        }
        v2[0] = r0.x; v2[1] = r0.y; v2[2] = r0.z; v2[3] = r0.w;
    }
    if (PROF && wave == 0 && lane == 0) a.prof[(size_t)blockIdx.x * 64 + 24] = wall_clock64();
    if (has_ln) {  // acc <- rstd[m] * (acc - mean[m] * colsum[n]); a register group holds 4 consecutive rows
        const float* so = ln_sm + (q >> 1) * 32 + (q & 1) * 16 + 4 * half;  // rows of v[0..3]; v[4..7] are 8 rows further
        const float* sb = ln_sm + wn * 32 + 8 * q + 4 * half;
        const float4 m0q = *(const float4*)so, r0q = *(const float4*)(so + 64);
        const float4 m1q = *(const float4*)(so + 8), r1q = *(const float4*)(so + 72);
        const float4 mbq = *(const float4*)sb, rbq = *(const float4*)(sb + 64);
        v[0] = r0q.x * (v[0] - m0q.x * cs_own); v[1] = r0q.y * (v[1] - m0q.y * cs_own);
        v[2] = r0q.z * (v[2] - m0q.z * cs_own); v[3] = r0q.w * (v[3] - m0q.w * cs_own);
        v[4] = r1q.x * (v[4] - m1q.x * cs_own); v[5] = r1q.y * (v[5] - m1q.y * cs_own);
        v[6] = r1q.z * (v[6] - m1q.z * cs_own); v[7] = r1q.w * (v[7] - m1q.w * cs_own);
        v2[0] = rbq.x * (v2[0] - mbq.x * cs_b2); v2[1] = rbq.y * (v2[1] - mbq.y * cs_b2);
        v2[2] = rbq.z * (v2[2] - mbq.z * cs_b2); v2[3] = rbq.w * (v2[3] - mbq.w * cs_b2);
    }
    if (act == ACT_GELU_ERF) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_erf_fast(v[r] + bias_own);
#pragma unroll
        for (int r = 0; r < 4; ++r) v2[r] = gelu_erf_fast(v2[r] + bias_b2);
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bias_own;
#pragma unroll
        for (int r = 0; r < 4; ++r) v2[r] += bias_b2;
    }
    if (PROF && wave == 0 && lane == 0) a.prof[(size_t)blockIdx.x * 64 + 25] = wall_clock64();
    if (!(a.dbg & 16)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int off = (r & 3) + 8 * (r >> 2);
            if (row_own + off < M) store_from_f32<OutT>(c_own + (size_t)off * ldc, v[r]);
        }
        if (b2_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row_b2 + r < M) store_from_f32<OutT>(c_b2 + (size_t)r * ldc, v2[r]);
        }
    }
    if (PROF) {
        if (wave == 0 && lane == 0) a.prof[(size_t)blockIdx.x * 64 + 26] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(29);
    }
}

// W[N][ldw] (row-major bf16, K = 64 nkt columns used) -> fragment order: block (column tile tn of 72, column block wn of 32,
// k-tile kt) = 4 KiB = 4 k-steps x 64 lanes x 16 B; lane l = 32 h + c of k-step ks holds W[72 tn + 32 wn + c][64 kt + 8 (2 ks + h) ..+8]
// (zeros where 32 wn + c >= 72: those lanes are never loaded)
__global__ void pack_w_frag_kernel(const bf16* __restrict__ W, int ldw, int N, int nkt, uint4* __restrict__ out, size_t nchunks) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunks) return;
    const int l = (int)(i & 63), ks = (int)((i >> 6) & 3);
    const size_t blk = i >> 8;
    const int kt = (int)(blk % nkt);
    const size_t tw = blk / nkt;
    const int wn = (int)(tw % 3), tn = (int)(tw / 3);
    const int c = l & 31, h = l >> 5;
    const int col = wn * 32 + c, n = tn * BN + col;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (col < BN && n < N) v = *(const uint4*)(W + (size_t)n * ldw + kt * BK + (2 * ks + h) * 8);
    out[i] = v;
}

}  // namespace

size_t gemm_dec_frag_bytes(int N, int K) { return (N % BN == 0 && K % BK == 0) ? (size_t)(N / BN) * 3 * (K / BK) * 4096 : 0; }

int launch_pack_w_frag(const void* W, int ldw, int N, int K, void* out, hipStream_t s) {
    DIMX_REQUIRE(N % BN == 0 && K % BK == 0 && ldw % 8 == 0 && ((uintptr_t)W % 16) == 0, DIMX_ERR_ARG, "pack_w_frag: N %% 72, K %% 64 (N=%d K=%d)", N, K);
    const size_t nchunks = gemm_dec_frag_bytes(N, K) / 16;
    hipLaunchKernelGGL(pack_w_frag_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, (const bf16*)W, ldw, N, K / BK, (uint4*)out,
                       nchunks);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

// plain row-major destination (one segment, unit column stride, rows b * sb with rowT == 1 or b * rowT * st): what
// gemm_set_plain_out builds
static bool plain_out(const GemmArgs& a) {
    return a.nseg == 1 && a.seg[0].sd == 1 && a.seg[0].sh == 0 && a.seg[0].sb == a.seg[0].st * (long)a.rowT && a.rowadd_mode == 0;
}

bool gemm_dec_eligible(const GemmArgs& a) {
    static const bool off = getenv("DIMX_NO_GEMM_DEC") != nullptr;
    if (off || !a.w_frag || a.in_dtype != DIMX_BF16) return false;
    const int kext = a.kloop ? a.kloop : a.ldw;
    if (a.conv_T != 0 || a.K % BK != 0 || a.K != kext || a.force_simple || a.w_tiled) return false;
    if (a.M > 256 || a.N % BN != 0 || ceil_div(a.M, BM) * (a.N / BN) > 256) return false;
    if (!plain_out(a) || a.residual || (a.act != ACT_NONE && a.act != ACT_GELU_ERF)) return false;
    if (a.ln_stats && !(a.ln_colsum && a.ln_C > 0 && a.K >= 2 * BK)) return false;
    return true;
}

int launch_gemm_dec(const GemmArgs& g, hipStream_t s) {
    DIMX_REQUIRE(gemm_dec_eligible(g), DIMX_ERR_ARG, "gemm_dec: not eligible (M=%d N=%d K=%d)", g.M, g.N, g.K);
    DIMX_REQUIRE(g.splitk >= 1 && (!g.ln_stats || g.splitk == 1) && (g.splitk == 1 || g.out_slabs), DIMX_ERR_ARG,
                 "gemm_dec: split-K needs slab output, the deferred LayerNorm needs one split (splitk %d)", g.splitk);
    DIMX_REQUIRE(((uintptr_t)g.A % 16) == 0 && g.lda % 8 == 0, DIMX_ERR_ARG, "gemm_dec: A rows must be 16-byte aligned");
    DecArgs a;
    memset(&a, 0, sizeof(a));
    a.A = (const bf16*)g.A;
    a.Wf = (const unsigned char*)g.w_frag;
    a.C = g.seg[0].ptr;
    a.bias = g.bias;
    a.ln_stats = g.ln_stats;
    a.ln_colsum = g.ln_colsum;
    a.ln_err = g.ln_err;
    a.prof = g.prof;
    a.slab_stride = g.out_slabs ? g.slab_stride : 0;
    a.lda = g.lda;
    a.ldc = (int)g.seg[0].st;
    a.M = g.M;
    a.N = g.N;
    a.nkt_all = g.K / BK;
    a.splitk = g.splitk;
    a.act = g.act;
    a.ln_C = g.ln_C;
    static const int dbg = getenv("DIMX_DEC_ABL") ? atoi(getenv("DIMX_DEC_ABL")) : 0;
    a.dbg = dbg;
    const int blocks = ceil_div(g.M, BM) * (g.N / BN) * g.splitk;
    // ring geometry (DIMX_DEC_KT, tuning): k-tiles per barrier -- 2 (6 slots of 16 KiB) or 4 (3 slots of 32 KiB); 1 = 12 slots of 8 KiB
    static const int kt_env = getenv("DIMX_DEC_KT") ? atoi(getenv("DIMX_DEC_KT")) : 2;
#define GD(OT, SS, KK, RR, PR) hipLaunchKernelGGL((gemm_dec_kernel<OT, SS, KK, RR, PR>), dim3(blocks), dim3(NTHREADS), 0, s, a)
#define GD_T(SS, KK, RR)                                             \
    do {                                                             \
        if (g.out_dtype == DIMX_BF16) {                              \
            if (g.prof) GD(bf16, SS, KK, RR, true); else GD(bf16, SS, KK, RR, false);   \
        } else {                                                     \
            if (g.prof) GD(float, SS, KK, RR, true); else GD(float, SS, KK, RR, false); \
        }                                                            \
    } while (0)
    if (kt_env == 4) GD_T(3, 4, 8);
    else if (kt_env == 1) GD_T(12, 1, 6);
    else GD_T(6, 2, 6);
#undef GD_T
#undef GD
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
