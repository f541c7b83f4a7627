// chain.hip -- XCD-local chain kernels of the autoregressive decoder step (bf16 perf mode, 8 <= B <= 256).
//
// Between two attention kernels a decoder layer runs  projection -> residual add -> LayerNorm -> projection
// (x-transformers Decoder, pre-norm, constructed at reference code/seq2seq_pretrain.py:413-419; one step of
// AutoregressiveWrapper.generate, :450).  As separate launches those are 3 dependent kernels of 5-8 us each whose
// cost is fixed overhead, not work (profiles/r01c_*: 43 kernels per step).  Clips are independent, so the batch is
// split into 8 groups of 32 clips and group g lives entirely on XCD g (block b runs on XCD b % 8: measured,
// profiles/r02_xcd_probe.txt; verified in the kernel).  Inside a group:
//   * GEMM phases: M = 32 rows (one 32x32x16 MFMA row block), the N columns are split over the XCD's 32 CUs; a CU
//     pulls its whole weight slice (and the 32 activation rows) into LDS by LDS-DMA in one burst, the 8 waves split
//     K and combine through LDS in a fixed order (deterministic).
//   * row phase: CU i owns clip i of the group: residual add (+ split-K slabs of a preceding chip-wide GEMM), two
//     exact LayerNorm passes, bf16 row for the next GEMM.
//   * hand-offs stay inside the XCD's coherent L2: plain stores, s_waitcnt vmcnt(0), an arrival on a per-XCD
//     counter (0.9 us per barrier measured vs 4-7 us for a chip-wide one), readers use sc1 loads (L1 bypass).
// The weights of a phase are read once per XCD (8x the chip-wide GEMM's fabric traffic, measured 7 TB/s), which is why
// only the small projections (attention out / cross-q / logits: 0.6-1.8 MB each) take this path and the feed-forward
// and fused-qkv GEMMs (5-11 MB) stay chip-wide launches.
#include <algorithm>

#include "common.hpp"
#include "decode_attn_body.hpp"

namespace dimx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kGroupCUs = 32;   // CUs (= blocks) per XCD
constexpr int kGroupRows = 32;  // clips per group = one MFMA row block
constexpr int AUX_PLAIN = 0, AUX_SC1 = 16;
constexpr size_t kMaxDynLds = 160 * 1024 - 512;  // the kernel also has static LDS (row statistics, < 512 B)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned ld_sc1_u32(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1_f32(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int lds_off(int row, int kc) { return row * 128 + (((kc ^ (row >> 1)) & 7) << 4); }

// Barrier among the 32 blocks of one XCD.  Every thread drains its own stores (they are then in the XCD's L2), the
// block meets, lane 0 arrives on the group's monotonic counter and polls it with L1-bypassing loads.  Bounded: a
// placement that is not one-block-per-CU would otherwise hang the GPU.
__device__ __forceinline__ void xcd_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(ld_sc1_u32(ctr) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if (spins > (1u << 20) || ((spins & 1023u) == 0 && (ld_sc1_u32(err) & 2u))) {  // bounded; dead once one timed out
                atomicOr(err, 2u);
                break;
            }
        }
    }
    __syncthreads();
}

// rows [r_begin, r_begin + rows_pad) (clamped to r_last) x K of a row-major bf16 matrix -> LDS as K/64 tiles of
// rows_pad x 128 B; chunk c of a row sits at slot c ^ ((row >> 1) & 7).  Pieces of 8 rows x 128 B (1 KiB, one wave-wide
// global_load_lds) are dealt round-robin to the 8 waves.
template <int AUX, int NW>
__device__ __forceinline__ void issue_panel_nw(const bf16* base, int ld, int r_begin, int r_last, int rows_pad, int nkt,
                                               unsigned char* dst, int wave, int lane) {
    const int groups = rows_pad >> 3;
    const int pieces = nkt * groups;
    for (int p = wave; p < pieces; p += NW) {
        const int kt = p / groups, gr = p - kt * groups;
        const int row = gr * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int r = r_begin + row;
        r = r <= r_last ? r : r_last;
        const bf16* src = base + (size_t)r * ld + kt * 64 + c * 8;
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst + (size_t)p * 1024), 16, 0, AUX);
    }
}
template <int AUX>
__device__ __forceinline__ void issue_panel(const bf16* base, int ld, int r_begin, int r_last, int rows_pad, int nkt,
                                            unsigned char* dst, int wave, int lane) {
    issue_panel_nw<AUX, kWaves>(base, ld, r_begin, r_last, rows_pad, nkt, dst, wave, lane);
}
template <int AUX>
__device__ __forceinline__ void issue_panel4(const bf16* base, int ld, int r_begin, int r_last, int rows_pad, int nkt,
                                             unsigned char* dst, int wave, int lane) {
    issue_panel_nw<AUX, 4>(base, ld, r_begin, r_last, rows_pad, nkt, dst, wave, lane);
}

// acc[j] += A(32 x K) . W_j(32 x K)^T over the k-tiles kt == wave (mod 8)
template <int NCB>
__device__ __forceinline__ void mfma_panel(const unsigned char* Apan, const unsigned char* Wpan, int nkt, int wrows_pad,
                                           f32x16_t (&acc)[NCB], int wave, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    for (int kt = wave; kt < nkt; kt += kWaves) {
        const unsigned char* At = Apan + (size_t)kt * (kGroupRows * 128);
        const unsigned char* Wt = Wpan + (size_t)kt * wrows_pad * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kc = 2 * ks + half;
            const uint4 fa = *(const uint4*)(At + lds_off(l31, kc));
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                const uint4 fw = *(const uint4*)(Wt + lds_off(j * 32 + l31, kc));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa),
                                                                 __builtin_bit_cast(bf16x8_t, fw), acc[j], 0, 0, 0);
            }
        }
    }
}

// the 8 waves' partial 32 x (NCB*32) tiles -> LDS -> summed in wave order -> out[row0 + m][n0 + col] (valid part)
template <int NCB>
__device__ __forceinline__ void reduce_store(const f32x16_t (&acc)[NCB], float* red, float* out, long ld_out, int row0,
                                             int nrows, int n0, int ncols, int wave, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    __syncthreads();  // every wave is done with the panels the scratch aliases
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[((wave * NCB + j) * 32 + m) * 32 + l31] = acc[j][r];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < NCB * 1024; e += kThreads) {
        const int j = e >> 10, m = (e >> 5) & 31, n = e & 31;
        float v = red[((0 * NCB + j) * 32 + m) * 32 + n];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[((w * NCB + j) * 32 + m) * 32 + n];
        const int col = j * 32 + n;
        if (col < ncols && m < nrows) out[(size_t)(row0 + m) * ld_out + n0 + col] = v;
    }
}

// Deferred-LayerNorm form of reduce_store (ChainArgs.defer): the summed projection gets the residual of the CU's own
// column slice (xs, loaded at kernel start with the same element mapping), x and bf16(x) are written, and the partial
// row sums of the slice go to stats[row][2].  Thread t holds rows (t >> 5) and (t >> 5) + 16, one column per 32-lane half.
template <int NCB>
__device__ __forceinline__ void reduce_store_defer(const f32x16_t (&acc)[NCB], float* red, const float (&xs)[2 * NCB], float* x,
                                                   bf16* y, int C, float* stats, int row0, int nrows, int n0, int ncols,
                                                   int wave, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    __syncthreads();  // every wave is done with the panels the scratch aliases
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[((wave * NCB + j) * 32 + m) * 32 + l31] = acc[j][r];
        }
    __syncthreads();
    // Partial statistics of the slice, in a form that survives |mean| >> std (sum x^2 - (sum x)^2 / n in f32
    // cancels catastrophically there): {sum x, M2 = sum (x - mean_slice)^2}, two passes over the values the thread holds; the
    // consumer combines the 32 slices with the parallel-variance formula, again in two passes.
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, xv[2 * NCB];
    bool ok[2 * NCB];
#pragma unroll
    for (int i = 0; i < 2 * NCB; ++i) {
        const int e = threadIdx.x + i * kThreads;
        const int j = e >> 10, m = (e >> 5) & 31, n = e & 31;
        float v = red[((0 * NCB + j) * 32 + m) * 32 + n];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[((w * NCB + j) * 32 + m) * 32 + n];
        const int col = j * 32 + n;
        ok[i] = col < ncols && m < nrows;
        xv[i] = 0.f;
        if (ok[i]) {
            const float xp = xs[i] + v;
            const size_t o = (size_t)(row0 + m) * C + n0 + col;
            x[o] = xp;
            y[o].x = f32_to_bf16(xp);
            xv[i] = xp;
            s1[i & 1] += xp;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // the 32 lanes of a row: four in-row steps on DPP (== xor 1, 2, 4, 8), one lane swap
        s1[k] = row16_sum(s1[k]);
        s1[k] += xor_lane_f32<16>(s1[k]);
    }
    const float inv_n = 1.0f / (float)ncols;
#pragma unroll
    for (int i = 0; i < 2 * NCB; ++i) {
        const float d = xv[i] - s1[i & 1] * inv_n;
        if (ok[i]) s2[i & 1] += d * d;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s2[k] = row16_sum(s2[k]);
        s2[k] += xor_lane_f32<16>(s2[k]);
    }
    if (l31 == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = (threadIdx.x >> 5) + 16 * k;
            stats[2 * m] = s1[k];
            stats[2 * m + 1] = s2[k];
        }
    }
}

// second projection of the deferred form: out = rstd[m] * (sum - mean[m] * colsum[col])
template <int NCB>
__device__ __forceinline__ void reduce_store_ln(const f32x16_t (&acc)[NCB], float* red, float* out, long ld_out, int row0,
                                                int nrows, int n0, int ncols, const float* mr, const float* colsum, int wave,
                                                int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[((wave * NCB + j) * 32 + m) * 32 + l31] = acc[j][r];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < NCB * 1024; e += kThreads) {
        const int j = e >> 10, m = (e >> 5) & 31, n = e & 31;
        float v = red[((0 * NCB + j) * 32 + m) * 32 + n];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[((w * NCB + j) * 32 + m) * 32 + n];
        const int col = j * 32 + n;
        if (col < ncols && m < nrows) out[(size_t)(row0 + m) * ld_out + n0 + col] = mr[2 * m + 1] * (v - mr[2 * m] * colsum[n0 + col]);
    }
}

#define CHAIN_STAMP(i)                                                            \
    do {                                                                          \
        if (a.prof && threadIdx.x == 0) a.prof[blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)

// block barrier that orders LDS traffic only: it must NOT drain vmcnt (the loader waves have weight DMAs in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace

// HAS_G1: projection of the attention output first; HAS_G2: projection of the normalised row last.
// NCB1 / NCB2: 32-column blocks of a CU's slice of the two projections.
// Wave roles after the first group barrier: waves 0-3 normalise the CU's row, waves 4-7 stream the second
// projection's weight slice into LDS (vmcnt is per wave, so the row's loads do not queue behind that burst).
// DEFER: deferred LayerNorm (ChainArgs.defer, needs HAS_G1): no row phase, one group barrier less.
template <bool HAS_G1, bool HAS_G2, int NCB1, int NCB2, bool DEFER = false>
__global__ __launch_bounds__(kThreads) void xcd_chain_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ float sm_s[4], sm_q[4];
    __shared__ float sm_mr[2 * kGroupRows];  // DEFER: {mean, rstd} of the group's rows
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // The group IS the XCD the block really runs on: consecutive block ids go to consecutive XCDs, but the first block of
    // a launch does not always land on XCD 0 (the rotation continues from the previous dispatch: observed after kernels
    // whose grid is not a multiple of 8).  Under any rotation the blocks of one XCD have distinct blockIdx >> 3, which
    // makes (xcc, blockIdx >> 3) a bijection onto (group, CU slot); that is checked, not assumed (seen[] below).
    const int g = (int)(xcc_id() & (a.fault ? 6u : 7u)), li = blockIdx.x >> 3;
    const unsigned stamp = (unsigned)(*a.step) + 1u;
    unsigned seen_old = 0;
    if (tid == 0) seen_old = atomicExch(a.seen + g * kGroupCUs + li, stamp);
    const int row0 = g * kGroupRows;
    const int nrows = a.B - row0 < kGroupRows ? a.B - row0 : kGroupRows;
    if (nrows <= 0) {  // the whole group leaves (all of its blocks take this branch)
        if (tid == 0 && seen_old == stamp) atomicOr(a.err, 1u);
        return;
    }
    const int r_last = row0 + nrows - 1;
    unsigned* ctr = a.counters + 16 * g;
    unsigned target = (unsigned)(*a.step) * (unsigned)(a.nbar * kGroupCUs);

    CHAIN_STAMP(0);
    // ---- row waves: everything of the CU's own row that is already there goes into registers now (x, gamma, the
    // split-K slabs of a preceding chip-wide GEMM, all loads in flight together); only xr has to wait for barrier 1
    const bool own = !DEFER && li < nrows && wave < 4;
    const int row = row0 + (li < nrows ? li : 0), C = a.C;
    float* xrow = a.x + (size_t)row * C;
    float v[6], gm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] = 0.f;
        gm[i] = 0.f;
    }
    if (own) {
        float t[8][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + i * 256;
            if (c < C) {
                v[i] = xrow[c];
                gm[i] = a.gamma[c];
            }
        }
        if (a.nslab > 0) {
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const float* sp = a.slabs + (size_t)(sl < a.nslab ? sl : 0) * a.slab_stride + (size_t)row * C;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int c = tid + i * 256;
                    t[sl][i] = sp[c < C ? c : 0];  // unconditional loads (all in flight together), masked below
                }
            }
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)  // slab order = the order the one-kernel-per-op step adds them in
                if (sl < a.nslab) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) v[i] += (tid + i * 256 < C) ? t[sl][i] : 0.f;
                }
        }
    }

    // ---- burst 1 + first projection: partial[32 rows][this CU's columns] -> xr
    if (HAS_G1) {
        issue_panel<AUX_PLAIN>((const bf16*)a.A1, a.lda1, row0, r_last, kGroupRows, a.g1.nkt, lds + a.offA1, wave, lane);
        const int n0 = li * a.g1.cols;
        issue_panel<AUX_PLAIN>((const bf16*)a.g1.W, a.g1.ldw, n0, n0 + a.g1.cols - 1, a.g1.rows_pad, a.g1.nkt,
                               lds + a.offW1, wave, lane);
        float xs[2 * NCB1];  // DEFER: the residual of this CU's column slice, element mapping of reduce_store_defer
        if (DEFER) {
#pragma unroll
            for (int i = 0; i < 2 * NCB1; ++i) {
                const int e = tid + i * kThreads;
                const int j = e >> 10, m = (e >> 5) & 31, col = j * 32 + (e & 31);
                const int mm = m < nrows ? m : nrows - 1, cc = col < a.g1.cols ? col : a.g1.cols - 1;
                xs[i] = a.x[(size_t)(row0 + mm) * C + n0 + cc];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CHAIN_STAMP(1);
        f32x16_t acc[NCB1];
#pragma unroll
        for (int j = 0; j < NCB1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        mfma_panel<NCB1>(lds + a.offA1, lds + a.offW1, a.g1.nkt, a.g1.rows_pad, acc, wave, lane);
        CHAIN_STAMP(2);
        if (DEFER) {
            reduce_store_defer<NCB1>(acc, (float*)(lds + a.offRed1), xs, a.x, (bf16*)a.y, C,
                                     a.stats + ((size_t)(g * kGroupCUs + li) * kGroupRows) * 2, row0, nrows, n0, a.g1.cols, wave,
                                     lane);
            CHAIN_STAMP(3);
            if (tid == 0 && seen_old == stamp) atomicOr(a.err, 1u);
            if (!HAS_G2) return;  // x, y and the partial sums are complete at the end of the launch: no group barrier
        } else {
            reduce_store<NCB1>(acc, (float*)(lds + a.offRed1), a.xr, a.C, row0, nrows, n0, a.g1.cols, wave, lane);
            CHAIN_STAMP(3);
        }
        target += kGroupCUs;
        if (DEFER) {
            // the group barrier, with the W2 slice requested between the arrival and the wait (there is no row phase to hide
            // that burst behind any more): stores drained by every thread, block meets, waves 4-7 issue the DMAs, thread 0
            // arrives and polls, the block meets again WITHOUT draining vmcnt
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave >= 4) {
                const int n2 = li * a.g2.cols;
                issue_panel4<AUX_PLAIN>((const bf16*)a.g2.W, a.g2.ldw, n2, n2 + a.g2.cols - 1, a.g2.rows_pad, a.g2.nkt,
                                        lds + a.offW2, wave - 4, lane);
            }
            if (tid == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while ((int)(ld_sc1_u32(ctr) - target) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if (spins > (1u << 20) || ((spins & 1023u) == 0 && (ld_sc1_u32(a.err) & 2u))) {
                        atomicOr(a.err, 2u);
                        break;
                    }
                }
            }
            lds_barrier();
        } else {
            xcd_barrier(ctr, target, a.err);
        }
        CHAIN_STAMP(4);
    }
    if (DEFER) {
        // ---- deferred form, after the only group barrier: the un-normalised rows of the group, the 32 x 32 partial sums ->
        // {mean, rstd} per row; second projection with the LayerNorm folded into its epilogue
        issue_panel<AUX_SC1>((const bf16*)a.y, a.C, row0, r_last, kGroupRows, a.g2.nkt, lds + a.offA2, wave, lane);
        {
            const int m = tid >> 4, part = tid & 15;
            const float* sp = a.stats + ((size_t)(g * kGroupCUs + 2 * part) * kGroupRows + m) * 2;
            const float a1 = ld_sc1_f32(sp), b1 = ld_sc1_f32(sp + 2 * kGroupRows);          // slice sums of two CUs
            const float a2 = ld_sc1_f32(sp + 1), b2 = ld_sc1_f32(sp + 2 * kGroupRows + 1);  // their centred squares
            const float t1 = row16_sum(a1 + b1);  // == the xor 1, 2, 4, 8 butterfly over the row's 16 parts
            const float mean = t1 * (1.0f / C);
            const float ncs = (float)a.g1.cols, inv_n = 1.0f / ncs;
            const float da = a1 * inv_n - mean, db = b1 * inv_n - mean;
            const float t2 = row16_sum(a2 + b2 + ncs * (da * da + db * db));
            if (part == 0) {
                const float var = t2 * (1.0f / C);
                sm_mr[2 * m] = mean;
                sm_mr[2 * m + 1] = rsqrtf(var + 1e-5f);
                // the operand of the next projection is bf16(x), NOT bf16(x - mean): with |mean| > 8 std its rounding noise
                // is no longer small against the row's spread -- flag it, the host repeats the batch with the row-phase form
                if (m < nrows && mean * mean > 64.0f * var) atomicOr(a.err, 4u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CHAIN_STAMP(7);
        f32x16_t acc[NCB2];
#pragma unroll
        for (int j = 0; j < NCB2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        mfma_panel<NCB2>(lds + a.offA2, lds + a.offW2, a.g2.nkt, a.g2.rows_pad, acc, wave, lane);
        CHAIN_STAMP(8);
        reduce_store_ln<NCB2>(acc, (float*)(lds + a.offRed2), a.out2, a.ld_out2, row0, nrows, li * a.g2.cols, a.g2.cols, sm_mr,
                              a.colsum2, wave, lane);
        CHAIN_STAMP(9);
        return;
    }

    // ---- loader waves: the second projection's weight slice (independent of everything computed here)
    if (HAS_G2 && wave >= 4) {
        const int n0 = li * a.g2.cols;
        issue_panel4<AUX_PLAIN>((const bf16*)a.g2.W, a.g2.ldw, n0, n0 + a.g2.cols - 1, a.g2.rows_pad, a.g2.nkt,
                                lds + a.offW2, wave - 4, lane);
    }
    // ---- row waves: residual add, two exact LayerNorm passes
    float s = 0.f;
    if (own) {
        if (HAS_G1) {
            float xr[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int c = tid + i * 256;
                xr[i] = c < C ? ld_sc1_f32(a.xr + (size_t)row * C + c) : 0.f;  // written by the other CUs of the XCD
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] += xr[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) s += v[i];
        s = wave_sum_fast(s);
        if (lane == 0) sm_s[wave] = s;
    }
    lds_barrier();
    float mean = 0.f, q = 0.f;
    if (own) {
        mean = (sm_s[0] + sm_s[1] + sm_s[2] + sm_s[3]) * (1.0f / C);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + i * 256;
            const float d = c < C ? v[i] - mean : 0.f;
            q += d * d;
        }
        q = wave_sum_fast(q);
        if (lane == 0) sm_q[wave] = q;
    }
    lds_barrier();
    if (own) {
        const float rstd = rsqrtf((sm_q[0] + sm_q[1] + sm_q[2] + sm_q[3]) * (1.0f / C) + 1e-5f);
        bf16* yrow = (bf16*)a.y + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + i * 256;
            if (c < C) {
                if (HAS_G1 || a.nslab > 0) xrow[c] = v[i];
                yrow[c].x = f32_to_bf16((v[i] - mean) * rstd * gm[i]);
            }
        }
    }
    CHAIN_STAMP(5);
    if (tid == 0 && seen_old == stamp) atomicOr(a.err, 1u);  // two blocks claimed the same (XCD, slot): not a bijection
    if (!HAS_G2) return;
    // group barrier 2: only the row waves have stores to drain; the loader waves keep their DMAs in flight
    target += kGroupCUs;
    if (wave < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(ld_sc1_u32(ctr) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if (spins > (1u << 20) || ((spins & 1023u) == 0 && (ld_sc1_u32(a.err) & 2u))) {
                atomicOr(a.err, 2u);
                break;
            }
        }
    }
    lds_barrier();
    CHAIN_STAMP(6);

    // ---- second projection on the normalised rows of the whole group (written by 32 CUs of this XCD: sc1)
    {
        issue_panel<AUX_SC1>((const bf16*)a.y, a.C, row0, r_last, kGroupRows, a.g2.nkt, lds + a.offA2, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows, and on waves 4-7 the weight slice
        __syncthreads();
        CHAIN_STAMP(7);
        f32x16_t acc[NCB2];
#pragma unroll
        for (int j = 0; j < NCB2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        mfma_panel<NCB2>(lds + a.offA2, lds + a.offW2, a.g2.nkt, a.g2.rows_pad, acc, wave, lane);
        const int n0 = li * a.g2.cols;
        CHAIN_STAMP(8);
        reduce_store<NCB2>(acc, (float*)(lds + a.offRed2), a.out2, a.ld_out2, row0, nrows, n0, a.g2.cols, wave, lane);
        CHAIN_STAMP(9);
    }
}

// ------------------------------------------------------------------------------------------------
// xcd_layer_kernel (round 5): the attention half of a decoder layer as ONE launch.
//
// Before: self attention | chain A {out-projection + residual, cross-q} | cross attention | chain B {out-projection + residual}
// = four launches per layer, each paying the launch floor (2.1 us for an empty 256-block kernel, profiles/r05_gemm_dec_experiment.txt),
// and the two chain launches a 3 us weight burst before their first MFMA.  Clip i of group g is CU slot i of XCD g in the chain
// kernels already; give that CU the clip's 12 (clip, head) attention waves as well and every hand-off of the four stages stays
// inside the XCD: four group barriers instead of three kernel boundaries, and the projections' weight slices are requested by
// a thirteenth wave while the attention waves stream their K/V caches (HBM-bound, 23-38 us: the bursts disappear behind them).
//   waves 0-11: one (clip, head) pair each during the attentions (decode_attn_body, unchanged: the kernel stays a pure stream);
//               waves 0-7 run the MFMA phases (8-way split K, fixed-order LDS reduce, as in xcd_chain_kernel)
//   wave 12:    issues the weight-slice LDS-DMAs (its own vmcnt: nothing the attention waves wait for)
// Deferred LayerNorm throughout (the form xcd_chain_kernel<.., DEFER> runs): x, bf16(x) and the partial row sums are written by
// the CU that owns the column slice; cross-q corrects its result with {mean, rstd}; the last stage leaves the statistics for the
// feed-forward GEMM's epilogue (GemmArgs.ln_stats).  The cross-attention query is read with L1-bypassing loads (QSC1): it was
// written by the XCD's other CUs during this launch.
constexpr int kLayerWaves = 13, kLayerThreads = kLayerWaves * 64, kAttnWaves = 12;

// waves 0-7: partial tiles -> LDS -> summed in wave order; + residual slice; x, y = bf16(x), partial statistics of the slice
// (reduce_store_defer for a block of 13 waves: the block barriers are met by every wave, the work is done by the first 8)
template <int NCB>
__device__ __forceinline__ void layer_reduce_defer(const f32x16_t (&acc)[NCB], float* red, const float (&xs)[2 * NCB], float* x, bf16* y,
                                                   int C, float* stats, int row0, int nrows, int n0, int ncols, int wave, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    const bool mf = wave < kWaves;
    __syncthreads();  // every wave is done with the panels the scratch aliases
    if (mf) {
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                red[((wave * NCB + j) * 32 + m) * 32 + l31] = acc[j][r];
            }
    }
    __syncthreads();
    if (!mf) return;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, xv[2 * NCB];
    bool ok[2 * NCB];
#pragma unroll
    for (int i = 0; i < 2 * NCB; ++i) {
        const int e = threadIdx.x + i * kThreads;
        const int j = e >> 10, m = (e >> 5) & 31, n = e & 31;
        float v = red[((0 * NCB + j) * 32 + m) * 32 + n];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[((w * NCB + j) * 32 + m) * 32 + n];
        const int col = j * 32 + n;
        ok[i] = col < ncols && m < nrows;
        xv[i] = 0.f;
        if (ok[i]) {
            const float xp = xs[i] + v;
            const size_t o = (size_t)(row0 + m) * C + n0 + col;
            x[o] = xp;
            y[o].x = f32_to_bf16(xp);
            xv[i] = xp;
            s1[i & 1] += xp;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s1[k] = row16_sum(s1[k]);
        s1[k] += xor_lane_f32<16>(s1[k]);
    }
    const float inv_n = 1.0f / (float)ncols;
#pragma unroll
    for (int i = 0; i < 2 * NCB; ++i) {
        const float d = xv[i] - s1[i & 1] * inv_n;
        if (ok[i]) s2[i & 1] += d * d;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s2[k] = row16_sum(s2[k]);
        s2[k] += xor_lane_f32<16>(s2[k]);
    }
    if (l31 == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = (threadIdx.x >> 5) + 16 * k;
            stats[2 * m] = s1[k];
            stats[2 * m + 1] = s2[k];
        }
    }
}

template <int NCB>
__device__ __forceinline__ void layer_reduce_ln(const f32x16_t (&acc)[NCB], float* red, float* out, long ld_out, int row0, int nrows,
                                                int n0, int ncols, const float* mr, const float* colsum, int wave, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    const bool mf = wave < kWaves;
    __syncthreads();
    if (mf) {
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                red[((wave * NCB + j) * 32 + m) * 32 + l31] = acc[j][r];
            }
    }
    __syncthreads();
    if (!mf) return;
    for (int e = threadIdx.x; e < NCB * 1024; e += kThreads) {
        const int j = e >> 10, m = (e >> 5) & 31, n = e & 31;
        float v = red[((0 * NCB + j) * 32 + m) * 32 + n];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[((w * NCB + j) * 32 + m) * 32 + n];
        const int col = j * 32 + n;
        if (col < ncols && m < nrows) out[(size_t)(row0 + m) * ld_out + n0 + col] = mr[2 * m + 1] * (v - mr[2 * m] * colsum[n0 + col]);
    }
}

// group barrier for the 13-wave block: every thread drains its own stores (optionally a wave issues a weight burst between the
// arrival and the wait), thread 0 arrives and polls; the block meets WITHOUT draining vmcnt when a burst is in flight
template <typename F>
__device__ __forceinline__ void layer_barrier(unsigned* ctr, unsigned target, unsigned* err, F&& between) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    between();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(ld_sc1_u32(ctr) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if (spins > (1u << 20) || ((spins & 1023u) == 0 && (ld_sc1_u32(err) & 2u))) {
                atomicOr(err, 2u);
                break;
            }
        }
    }
    lds_barrier();
}

#define LAYER_STAMP(i)                                                              \
    do {                                                                            \
        if (a.prof && threadIdx.x == 0) a.prof[blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)

template <int NCB_SO, int NCB_CQ, int NCB_CO>
__global__ __launch_bounds__(kLayerThreads) void xcd_layer_kernel(const LayerChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ float sm_mr[2 * kGroupRows];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = (int)(xcc_id() & (a.fault ? 6u : 7u)), li = blockIdx.x >> 3;
    const unsigned epoch = (unsigned)((a.step ? *a.step : 0) + a.epoch_add);
    const unsigned stamp = epoch + 1u;
    unsigned seen_old = 0;
    if (tid == 0) seen_old = atomicExch(a.seen + g * kGroupCUs + li, stamp);
    const int row0 = g * kGroupRows;
    const int nrows = a.B - row0 < kGroupRows ? a.B - row0 : kGroupRows;
    if (nrows <= 0) {  // the whole group leaves (all of its blocks take this branch)
        if (tid == 0 && seen_old == stamp) atomicOr(a.err, 1u);
        return;
    }
    if (tid == 0 && seen_old == stamp) atomicOr(a.err, 1u);  // two blocks claimed the same (XCD, slot): not a bijection
    const int r_last = row0 + nrows - 1, C = a.C;
    unsigned* ctr = a.counters + 16 * g;
    unsigned target = epoch * (unsigned)(4 * kGroupCUs);
    const bool has_clip = li < nrows;
    const int clip = row0 + (has_clip ? li : 0);
    float* sc = (float*)lds;
    unsigned char* base = lds + a.off_base;
    LAYER_STAMP(0);

    // ---- stage 1: self attention of this CU's clip; wave 12 requests the out-projection's weight slice meanwhile
    if (wave == kAttnWaves) {
        const int n0 = li * a.g_so.cols;
        if (!(a.abl & 1))
        issue_panel_nw<AUX_PLAIN, 1>((const bf16*)a.g_so.W, a.g_so.ldw, n0, n0 + a.g_so.cols - 1, a.g_so.rows_pad, a.g_so.nkt, base + a.off_W1, 0,
                                     lane);
    } else if (has_clip) {
        const int aclip = (a.perm && nrows == kGroupRows) ? row0 + ((li + wave) & (kGroupRows - 1)) : clip;
        decode_attn_body<bf16, true, true, 1, kAttnWaves>(a.sa, aclip, sc, a.sc_stride, nullptr, nullptr, nullptr);
    }
    LAYER_STAMP(1);
    target += kGroupCUs;
    layer_barrier(ctr, target, a.err, [] {});
    LAYER_STAMP(2);

    // ---- stage 2: x += o . Wso^T (this CU's column slice of the group's 32 rows), y = bf16(x), partial row sums
    {
        const int n0 = li * a.g_so.cols;
        if (wave < kWaves)
            issue_panel<AUX_SC1>((const bf16*)a.o, a.ld_o, row0, r_last, kGroupRows, a.g_so.nkt, base + a.off_A1, wave, lane);
        float xs[2 * NCB_SO];
        if (wave < kWaves) {
#pragma unroll
            for (int i = 0; i < 2 * NCB_SO; ++i) {
                const int e = tid + i * kThreads;
                const int j = e >> 10, m = (e >> 5) & 31, col = j * 32 + (e & 31);
                const int mm = m < nrows ? m : nrows - 1, cc = col < a.g_so.cols ? col : a.g_so.cols - 1;
                xs[i] = a.x[(size_t)(row0 + mm) * C + n0 + cc];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows, wave 12: the weight slice
        __syncthreads();
        LAYER_STAMP(3);
        f32x16_t acc[NCB_SO];
#pragma unroll
        for (int j = 0; j < NCB_SO; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (wave < kWaves) mfma_panel<NCB_SO>(base + a.off_A1, base + a.off_W1, a.g_so.nkt, a.g_so.rows_pad, acc, wave, lane);
        layer_reduce_defer<NCB_SO>(acc, (float*)(base + a.off_A1), xs, a.x, (bf16*)a.y, C,
                                   a.stats + ((size_t)(g * kGroupCUs + li) * kGroupRows) * 2, row0, nrows, n0, a.g_so.cols, wave, lane);
    }
    LAYER_STAMP(4);
    target += kGroupCUs;
    layer_barrier(ctr, target, a.err, [&] {  // the cross-q weight slice goes out between the arrival and the wait, from EVERY wave:
        // one wave issuing its 54 pieces kept the whole block at the closing block barrier for ~2.5 us (profiles/r05_layer_kernel.txt)
        const int n2 = li * a.g_cq.cols;
        if (!(a.abl & 1))
        issue_panel_nw<AUX_PLAIN, kLayerWaves>((const bf16*)a.g_cq.W, a.g_cq.ldw, n2, n2 + a.g_cq.cols - 1, a.g_cq.rows_pad, a.g_cq.nkt,
                                               base + a.off_W2, wave, lane);
    });
    LAYER_STAMP(5);

    // ---- stage 3: qc = rstd * (x . Wq'^T - mean * colsum) on the un-normalised rows of the group
    {
        if (wave < kWaves) issue_panel<AUX_SC1>((const bf16*)a.y, C, row0, r_last, kGroupRows, a.g_cq.nkt, base + a.off_A2, wave, lane);
        if (tid < 512) {
            const int m = tid >> 4, part = tid & 15;
            const float* sp = a.stats + ((size_t)(g * kGroupCUs + 2 * part) * kGroupRows + m) * 2;
            const float a1 = ld_sc1_f32(sp), b1 = ld_sc1_f32(sp + 2 * kGroupRows);
            const float a2 = ld_sc1_f32(sp + 1), b2 = ld_sc1_f32(sp + 2 * kGroupRows + 1);
            const float t1 = row16_sum(a1 + b1);
            const float mean = t1 * (1.0f / C);
            const float ncs = (float)a.g_so.cols, inv_n = 1.0f / ncs;
            const float da = a1 * inv_n - mean, db = b1 * inv_n - mean;
            const float t2 = row16_sum(a2 + b2 + ncs * (da * da + db * db));
            if (part == 0) {
                const float var = t2 * (1.0f / C);
                sm_mr[2 * m] = mean;
                sm_mr[2 * m + 1] = rsqrtf(var + 1e-5f);
                if (m < nrows && mean * mean > 64.0f * var) atomicOr(a.err, 4u);  // bf16(x) too coarse for this row (see above)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        LAYER_STAMP(6);
        f32x16_t acc[NCB_CQ];
#pragma unroll
        for (int j = 0; j < NCB_CQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (wave < kWaves) mfma_panel<NCB_CQ>(base + a.off_A2, base + a.off_W2, a.g_cq.nkt, a.g_cq.rows_pad, acc, wave, lane);
        layer_reduce_ln<NCB_CQ>(acc, (float*)(base + a.off_A2), a.qc, a.ld_qc, row0, nrows, li * a.g_cq.cols, a.g_cq.cols, sm_mr, a.colsum_cq,
                                wave, lane);
    }
    LAYER_STAMP(7);
    target += kGroupCUs;
    layer_barrier(ctr, target, a.err, [] {});
    LAYER_STAMP(8);

    // ---- stage 4: cross attention of this CU's clip (the query rows were written by the XCD's other CUs: L1-bypassing loads);
    // wave 12 requests the second out-projection's weight slice meanwhile
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), visibly to hipcc: no LDS-DMA of this wave is pending when the attention's
                                         // LDS traffic starts (behind a possibly pending one it would drain vmcnt in the key loop)
    if (wave == kAttnWaves) {
        const int n0 = li * a.g_co.cols;
        if (!(a.abl & 1))
        issue_panel_nw<AUX_PLAIN, 1>((const bf16*)a.g_co.W, a.g_co.ldw, n0, n0 + a.g_co.cols - 1, a.g_co.rows_pad, a.g_co.nkt, base + a.off_W3, 0,
                                     lane);
    } else if (has_clip) {
        const int aclip = (a.perm && nrows == kGroupRows) ? row0 + ((li + wave) & (kGroupRows - 1)) : clip;
        decode_attn_body<bf16, false, true, 1, kAttnWaves, true>(a.ca, aclip, sc, a.sc_stride, nullptr, nullptr, nullptr);
    }
    LAYER_STAMP(9);
    target += kGroupCUs;
    layer_barrier(ctr, target, a.err, [] {});
    LAYER_STAMP(10);

    // ---- stage 5: x += o . Wco^T, y = bf16(x), partial row sums for the feed-forward GEMM's epilogue
    {
        const int n0 = li * a.g_co.cols;
        if (wave < kWaves)
            issue_panel<AUX_SC1>((const bf16*)a.o, a.ld_o, row0, r_last, kGroupRows, a.g_co.nkt, base + a.off_A3, wave, lane);
        float xs[2 * NCB_CO];
        if (wave < kWaves) {
#pragma unroll
            for (int i = 0; i < 2 * NCB_CO; ++i) {
                const int e = tid + i * kThreads;
                const int j = e >> 10, m = (e >> 5) & 31, col = j * 32 + (e & 31);
                const int mm = m < nrows ? m : nrows - 1, cc = col < a.g_co.cols ? col : a.g_co.cols - 1;
                xs[i] = a.x[(size_t)(row0 + mm) * C + n0 + cc];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        LAYER_STAMP(11);
        f32x16_t acc[NCB_CO];
#pragma unroll
        for (int j = 0; j < NCB_CO; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (wave < kWaves) mfma_panel<NCB_CO>(base + a.off_A3, base + a.off_W3, a.g_co.nkt, a.g_co.rows_pad, acc, wave, lane);
        layer_reduce_defer<NCB_CO>(acc, (float*)(base + a.off_W3), xs, a.x, (bf16*)a.y, C,
                                   a.stats + ((size_t)(g * kGroupCUs + li) * kGroupRows) * 2, row0, nrows, n0, a.g_co.cols, wave, lane);
    }
    LAYER_STAMP(12);
}

// ------------------------------------------------------------------------------------------------ host side
static void desc_fill(ChainGemmDesc& d) {
    d.cols = d.N / kGroupCUs;
    d.rows_pad = (d.cols + 7) / 8 * 8;
    d.ncb = (d.cols + 31) / 32;
    d.nkt = d.K / 64;
}

bool chain_supported(const ChainArgs& a0, int cu_count) {
    ChainArgs a = a0;
    if (cu_count != 8 * kGroupCUs) return false;  // one block per CU, 32 CUs per XCD
    if (a.B < 1 || a.B > 8 * kGroupRows || a.C > 3 * kThreads || a.C % 64 != 0) return false;
    for (ChainGemmDesc* d : {&a.g1, &a.g2}) {
        if (!d->W) continue;
        if (d->N % kGroupCUs != 0 || d->K % 64 != 0 || d->ldw % 8 != 0) return false;
        desc_fill(*d);
        if (d->ncb < 1 || d->ncb > 2) return false;
    }
    if (a.g1.W && (a.g1.N != a.C || a.lda1 % 8 != 0)) return false;
    if (a.g2.W && a.g2.K != a.C) return false;
    return chain_plan(a) <= kMaxDynLds;
}

// LDS plan (bytes); returns the dynamic LDS size.  Burst 1 = {A1, W1}; its reduction scratch aliases them once the
// MFMAs are done; A2 (the normalised rows) reuses the front, W2 sits behind A2.
size_t chain_plan(ChainArgs& a) {
    size_t front = 0, total = 0;
    if (a.g1.W) {
        desc_fill(a.g1);
        a.offA1 = 0;
        a.offW1 = a.g1.nkt * kGroupRows * 128;
        const size_t w1 = (size_t)a.g1.nkt * a.g1.rows_pad * 128 + 4096;  // + read slack of the padded column block
        a.offRed1 = 0;
        const size_t red1 = (size_t)kWaves * a.g1.ncb * 4096;
        front = a.offW1 + w1;
        front = front > red1 ? front : red1;
        total = front;
    }
    if (a.g2.W) {
        desc_fill(a.g2);
        const size_t a2 = (size_t)a.g2.nkt * kGroupRows * 128;
        const size_t red2 = (size_t)kWaves * a.g2.ncb * 4096;
        // W2 is issued after the first group barrier (burst 1 and its scratch are dead by then) and must only stay
        // clear of A2, which is loaded in front of it later; scratch 2 aliases A2 / W2 once the MFMAs are done
        const size_t w2_at = a2;
        a.offA2 = 0;
        a.offRed2 = 0;
        a.offW2 = (int)w2_at;
        size_t end2 = w2_at + (size_t)a.g2.nkt * a.g2.rows_pad * 128 + 4096;
        end2 = end2 > red2 ? end2 : red2;
        total = total > end2 ? total : end2;
    }
    return total;
}

int launch_chain(const ChainArgs& a0, hipStream_t s) {
    ChainArgs a = a0;
    DIMX_REQUIRE(a.x && a.y && a.gamma && a.counters && a.step && a.err, DIMX_ERR_ARG, "chain: null operand");
    DIMX_REQUIRE(!a.g1.W || (a.A1 && a.xr), DIMX_ERR_ARG, "chain: first projection needs A1 and xr");
    DIMX_REQUIRE(!a.g2.W || a.out2, DIMX_ERR_ARG, "chain: second projection needs out2");
    DIMX_REQUIRE(a.nslab == 0 || a.slabs, DIMX_ERR_ARG, "chain: slabs missing");
    const size_t lds = chain_plan(a);
    DIMX_REQUIRE(lds <= kMaxDynLds, DIMX_ERR_ARG, "chain: LDS plan %zu bytes", lds);
    a.nbar = (a.g1.W ? 1 : 0) + (a.g2.W ? 1 : 0);
    if (a.defer) {
        DIMX_REQUIRE(a.g1.W && a.stats && (!a.g2.W || a.colsum2), DIMX_ERR_ARG, "chain: deferred LayerNorm needs g1, stats (and colsum2)");
        a.nbar = a.g2.W ? 1 : 0;
    }
    dim3 grid(8 * kGroupCUs), block(kThreads);
#define CH(G1, G2, N1, N2)                                                                                      \
    do {                                                                                                        \
        (void)hipFuncSetAttribute((const void*)xcd_chain_kernel<G1, G2, N1, N2>,                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);                      \
        hipLaunchKernelGGL((xcd_chain_kernel<G1, G2, N1, N2>), grid, block, lds, s, a);                         \
    } while (0)
#define CHD(G2, N1, N2)                                                                                         \
    do {                                                                                                        \
        (void)hipFuncSetAttribute((const void*)xcd_chain_kernel<true, G2, N1, N2, true>,                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);                 \
        hipLaunchKernelGGL((xcd_chain_kernel<true, G2, N1, N2, true>), grid, block, lds, s, a);                 \
    } while (0)
    const int n1 = a.g1.W ? a.g1.ncb : 1, n2 = a.g2.W ? a.g2.ncb : 1;
    if (a.defer) {
        if (a.g2.W) {
            if (n1 == 1 && n2 == 1) CHD(true, 1, 1);
            else if (n1 == 2 && n2 == 1) CHD(true, 2, 1);
            else if (n1 == 1 && n2 == 2) CHD(true, 1, 2);
            else CHD(true, 2, 2);
        } else {
            if (n1 == 1) CHD(false, 1, 1);
            else CHD(false, 2, 1);
        }
    } else if (a.g1.W && a.g2.W) {
        if (n1 == 1 && n2 == 1) CH(true, true, 1, 1);
        else if (n1 == 2 && n2 == 1) CH(true, true, 2, 1);
        else if (n1 == 1 && n2 == 2) CH(true, true, 1, 2);
        else CH(true, true, 2, 2);
    } else if (a.g1.W) {
        if (n1 == 1) CH(true, false, 1, 1);
        else CH(true, false, 2, 1);
    } else if (a.g2.W) {
        if (n2 == 1) CH(false, true, 1, 1);
        else CH(false, true, 1, 2);
    } else {
        CH(false, false, 1, 1);
    }
#undef CH
#undef CHD
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}


// LDS plan of xcd_layer_kernel (bytes from the dynamic LDS base): scores [12][sc_stride] f32 | stage area.  Stage 2: A1 | W1 (its
// reduction scratch aliases A1 / W1 once the MFMAs are done); stage 3: A2 | W2 (W2 is requested after stage 2's reduction: it
// lies behind that scratch and may overlap the dead W1); stage 5: W3 | A3 (W3 is requested after stage 3; scratch aliases W3).
static size_t layer_plan(LayerChainArgs& a) {
    desc_fill(a.g_so);
    desc_fill(a.g_cq);
    desc_fill(a.g_co);
    const size_t sc_bytes = ((size_t)kAttnWaves * a.sc_stride * 4 + 1023) / 1024 * 1024;
    a.off_base = (int)sc_bytes;
    auto wbytes = [](const ChainGemmDesc& d) { return (size_t)d.nkt * d.rows_pad * 128 + 4096; };
    auto abytes = [](const ChainGemmDesc& d) { return (size_t)d.nkt * kGroupRows * 128; };
    auto rbytes = [](const ChainGemmDesc& d) { return (size_t)kWaves * d.ncb * 4096; };
    a.off_A1 = 0;
    a.off_W1 = (int)abytes(a.g_so);
    const size_t end1 = std::max(a.off_W1 + wbytes(a.g_so), rbytes(a.g_so));
    a.off_A2 = 0;
    a.off_W2 = (int)std::max(abytes(a.g_cq), rbytes(a.g_so));  // clear of stage 2's scratch (read while W2 lands) and of A2
    const size_t end2 = std::max(a.off_W2 + wbytes(a.g_cq), rbytes(a.g_cq));
    a.off_W3 = 0;
    a.off_A3 = (int)std::max(wbytes(a.g_co), rbytes(a.g_co));
    const size_t end3 = a.off_A3 + abytes(a.g_co);
    return sc_bytes + std::max(end1, std::max(end2, end3));
}

bool layer_chain_supported(const LayerChainArgs& a0, int cu_count) {
    LayerChainArgs a = a0;
    if (cu_count != 8 * kGroupCUs) return false;
    if (a.B < 1 || a.B > 8 * kGroupRows || a.C > 3 * kThreads || a.C % 64 != 0) return false;
    if (a.sa.H != kAttnWaves || a.ca.H != kAttnWaves || a.sa.dtype != DIMX_BF16 || a.ca.dtype != DIMX_BF16) return false;
    if (a.B * kAttnWaves * 2 <= 3072) return false;  // smaller batches split a (clip, head) pair over several waves (decode_attn.hip)
    if (a.ca.rows_per_clip > 1) return false;
    for (ChainGemmDesc* d : {&a.g_so, &a.g_cq, &a.g_co}) {
        if (!d->W || d->N % kGroupCUs != 0 || d->K % 64 != 0 || d->ldw % 8 != 0) return false;
        desc_fill(*d);
        if (d->ncb < 1 || d->ncb > 2) return false;
    }
    if (a.g_so.N != a.C || a.g_co.N != a.C || a.g_cq.K != a.C || a.g_so.ncb != 2 || a.g_co.ncb != 2 || a.g_cq.ncb != 1) return false;
    if (a.ld_o % 8 != 0) return false;
    return layer_plan(a) <= kMaxDynLds;
}

int launch_layer_chain(const LayerChainArgs& a0, hipStream_t s) {
    LayerChainArgs a = a0;
    DIMX_REQUIRE(a.x && a.y && a.o && a.qc && a.stats && a.colsum_cq && a.counters && a.seen && a.err, DIMX_ERR_ARG,
                 "layer_chain: null operand");
    const size_t lds = layer_plan(a);
    DIMX_REQUIRE(lds <= kMaxDynLds, DIMX_ERR_ARG, "layer_chain: LDS plan %zu bytes", lds);
    (void)hipFuncSetAttribute((const void*)xcd_layer_kernel<2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    hipLaunchKernelGGL((xcd_layer_kernel<2, 1, 2>), dim3(8 * kGroupCUs), dim3(kLayerThreads), lds, s, a);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
