// gemm.hip -- C = epilogue(A[M,K] . W[N,K]^T) on the gfx950 matrix cores.
//
// One kernel template serves every dense op of the path (reference: nn.Linear / nn.Conv1d calls in
// code/models/stage1_BIWI.py:258-393, code/models/lib/base_models.py:43-146 and the x-transformers
// Attention/FeedForward projections constructed at code/seq2seq_pretrain.py:388-418):
//   * T = bf16  -> v_mfma_f32_32x32x16_bf16 (perf mode), T = float -> v_mfma_f32_32x32x2_f32
//     (parity mode: exact f32 products, f32 accumulate).
//   * A and W tiles are staged global -> registers -> LDS as 16-byte chunks; an LDS row holds 128 bytes
//     of K (64 bf16 / 32 f32) and chunk c of row r is stored at slot c ^ ((r>>1)&7) so that the
//     ds_read_b128 fragment reads (32 different rows, same chunk) are bank-conflict free.
//   * a lane's 16-byte fragment feeds 1 bf16 MFMA (K=16) or 4 f32 MFMAs (K=2 each; the two lane halves
//     hold k and k+4 -- a k-permutation that leaves the sum unchanged).
//   * double-buffered LDS, next tile's global loads are issued before the MFMAs of the current tile.
//   * CONV: the A operand is gathered on the fly for the k=5 replicate-padded temporal convolution
//     (K = 5*C, W repacked tap-major), rows clamp inside their own clip [0, len_b).
//   * epilogue: bias, LeakyReLU / GELU, positional row add, f32 residual, and a strided scatter that
//     can write row-major activations, head-major K/V caches or the transposed V^T the attention
//     kernel consumes.
#include "common.hpp"
#include "decode_attn_body.hpp"

namespace dimx {

static int gemm_cfg_small();
static bool gemm_use_ws72(const GemmArgs& a);

void gemm_args_init(GemmArgs& a) {
    memset(&a, 0, sizeof(a));
    a.in_dtype = DIMX_F32;
    a.out_dtype = DIMX_F32;
    a.rowadd_scale = 1.f;
    a.nseg = 1;
    a.rowT = 1;
    a.splitk = 1;
    a.rowadd_div = 1;
}

void gemm_set_plain_out(GemmArgs& a, void* C, int ldc) {
    a.nseg = 1;
    a.seg_width = a.N;
    a.seg[0].ptr = C;
    a.seg[0].sb = (long)ldc * a.rowT;
    a.seg[0].st = ldc;
    a.seg[0].sh = 0;
    a.seg[0].sd = 1;
    a.seg[0].D = a.N > 0 ? a.N : 1;
}

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        const float4 fa = __builtin_bit_cast(float4, a), fb = __builtin_bit_cast(float4, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int lds_off(int row, int kc) { return row * 128 + (((kc ^ (row >> 1)) & 7) << 4); }


// ------------------------------------------------------------------------------------------------
// Epilogue shared by both kernels.  PMC on the prefill shapes showed ~2200 VALU instructions per wave against
// 48 MFMAs -- the epilogue, not the main loop, was the bottleneck -- so it is specialised:
//   * ACT is a compile-time parameter (the caller switches once, outside the element loops);
//   * FAST (bf16 perf mode): GELU through v_exp_f32 / v_rcp_f32 approximations (error far below bf16
//     rounding; the bare v_rcp_f32 -- __frcp_rn expands to a 12-instruction IEEE division, and the decode ff1
//     epilogue spent 1.5 us in 16 GELUs per lane, tools/gemm_phases.py); the f32 parity mode keeps tanhf / erff;
//   * a plain row-major destination gets a dedicated path (one 64-bit base per column, 32-bit row offsets);
//     the generic path handles head-major K/V caches, transposed V^T and per-row positional adds.
// ------------------------------------------------------------------------------------------------
template <int ACT, bool FAST> __device__ __forceinline__ float act_fn(float x) {
    if (ACT == ACT_LEAKY) return x > 0.f ? x : 0.2f * x;
    if (ACT == ACT_GELU_TANH) {
        const float u = 0.7978845608028654f * (x + 0.044715f * (x * x * x));
        if (FAST) {  // tanh(u) = 1 - 2 / (1 + e^{2u})
            const float e = __expf(2.0f * u);
            const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
            return x * (0.5f * (1.0f + t));
        }
        return x * (0.5f * (1.0f + tanhf(u)));
    }
    if (ACT == ACT_GELU_ERF) {
        if (FAST) {  // Abramowitz-Stegun 7.1.26, |error| < 1.5e-7
            const float z = fabsf(x) * 0.7071067811865476f;
            const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
            const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
            const float er = 1.0f - poly * __expf(-z * z);
            return 0.5f * x * (1.0f + (x < 0.f ? -er : er));
        }
        return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
    }
    return x;
}

template <typename OutT, int MI, int NI, int ACT, bool FAST>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& a, const f32x16_t (&acc)[MI][NI], int m0w, int n0w,
                                              int half, int l31, int split, unsigned long long* st = nullptr,
                                              int n_lim = 0x7fffffff, const float* bias_pre = nullptr) {
    // n_lim: first column the block does NOT own (gemm_ws72_kernel: a 96-wide MFMA tile holds 72 valid columns);
    // bias_pre[j]: the lane's bias values requested before the main loop (a dependent load here cost 0.5 us per launch)
    const bool first = split == 0;
    const bool plain = a.nseg == 1 && a.seg[0].sd == 1 && a.seg[0].sh == 0 && a.seg[0].sb == a.seg[0].st * (long)a.rowT;
    if (plain && a.rowadd_mode == 0) {
        const int ldc = (int)a.seg[0].st;
        OutT* C = (OutT*)a.seg[0].ptr + (a.out_slabs ? (size_t)split * a.slab_stride : 0);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0w + j * 32 + l31;
            if (n >= a.N || n >= n_lim) continue;
            float bias_v = bias_pre ? bias_pre[j] : ((a.bias && first) ? a.bias[n] : 0.f);
            if (st) {  // tools/gemm_phases.py: kernel arguments + bias in registers
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(bias_v)::"memory");
                st[24] = wall_clock64();
            }
            OutT* col = C + n;
            const float* rcol = a.residual ? a.residual + n : nullptr;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int mb = m0w + i * 32 + 4 * half;
                float vals[16];
                if (rcol) {  // all 16 residual loads in flight together, rows past M read row M - 1 (never stored)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int m = mb + (r & 3) + 8 * (r >> 2);
                        m = m < a.M ? m : a.M - 1;
                        vals[r] = rcol[(size_t)m * a.ldr];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) vals[r] += act_fn<ACT, FAST>(acc[i][j][r] + bias_v);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) vals[r] = act_fn<ACT, FAST>(acc[i][j][r] + bias_v);
                }
                if (st) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(vals[r]));
                    st[25] = wall_clock64();
                }
                if (mb + 27 < a.M) {  // the lane's last row is mb + 27: no per-row checks for interior tiles
#pragma unroll
                    for (int r = 0; r < 16; ++r) store_from_f32<OutT>(col + (size_t)(mb + (r & 3) + 8 * (r >> 2)) * ldc, vals[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        if (m < a.M) store_from_f32<OutT>(col + (size_t)m * ldc, vals[r]);
                    }
                }
                if (st) st[26] = wall_clock64();
            }
        }
        return;
    }
    const int rowT = a.rowT;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0w + j * 32 + l31;
        if (n >= a.N || n >= n_lim) continue;
        const float bias_v = bias_pre ? bias_pre[j] : ((a.bias && first) ? a.bias[n] : 0.f);
        int s = 0, nn = n;
        if (a.nseg > 1) {
            s = n / a.seg_width;
            nn = n - s * a.seg_width;
        }
        const OutSeg sg = a.seg[s];
        const int hh = nn / sg.D, dd = nn - hh * sg.D;
        OutT* obase = (OutT*)sg.ptr + (long)hh * sg.sh + (long)dd * sg.sd;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mb = m0w + i * 32 + 4 * half;
            if (rowT == 1 && a.rowadd_mode == 0 && !a.residual) {
                // decode step (one row per clip, e.g. the fused q/k/v projection writing q and the K/V cache rows): the
                // destination of row m is obase + m * sb; interior tiles store without per-row checks
                OutT* p = obase + (long)mb * sg.sb;
                if (mb + 27 < a.M) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        store_from_f32<OutT>(p + (long)((r & 3) + 8 * (r >> 2)) * sg.sb, act_fn<ACT, FAST>(acc[i][j][r] + bias_v));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = (r & 3) + 8 * (r >> 2);
                        if (mb + off < a.M) store_from_f32<OutT>(p + (long)off * sg.sb, act_fn<ACT, FAST>(acc[i][j][r] + bias_v));
                    }
                }
                continue;
            }
            int b0, t0;
            if (rowT == 1) {
                b0 = mb;
                t0 = 0;
            } else {
                b0 = mb / rowT;
                t0 = mb - b0 * rowT;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = (r & 3) + 8 * (r >> 2);
                const int m = mb + off;
                if (m >= a.M) continue;
                int b = b0, t = t0;
                if (rowT == 1) {
                    b = m;
                } else {
                    t += off;
                    while (t >= rowT) {
                        t -= rowT;
                        ++b;
                    }
                }
                float v = act_fn<ACT, FAST>(acc[i][j][r] + bias_v);
                if (a.rowadd_mode) {
                    const int ri = a.rowadd_mode == 1 ? t : (a.rowadd_mode == 2 ? b / a.rowadd_div + a.rowadd_off : a.rowadd_off);
                    v += a.rowadd[(size_t)ri * a.ld_rowadd + n] * a.rowadd_scale;
                }
                if (a.residual) v += a.residual[(size_t)m * a.ldr + n];
                store_from_f32<OutT>(obase + (long)b * sg.sb + (long)t * sg.st, v);
            }
        }
    }
}

template <typename T, typename OutT, int MI, int NI>
__device__ __forceinline__ void epilogue(const GemmArgs& a, const f32x16_t (&acc)[MI][NI], int m0w, int n0w, int half,
                                         int l31, int split, unsigned long long* st = nullptr, int n_lim = 0x7fffffff,
                                         const float* bias_pre = nullptr) {
    constexpr bool FAST = sizeof(T) == 2;
    switch (a.act) {
        case ACT_LEAKY: epilogue_tile<OutT, MI, NI, ACT_LEAKY, FAST>(a, acc, m0w, n0w, half, l31, split, st, n_lim, bias_pre); break;
        case ACT_GELU_TANH: epilogue_tile<OutT, MI, NI, ACT_GELU_TANH, FAST>(a, acc, m0w, n0w, half, l31, split, st, n_lim, bias_pre); break;
        case ACT_GELU_ERF: epilogue_tile<OutT, MI, NI, ACT_GELU_ERF, FAST>(a, acc, m0w, n0w, half, l31, split, st, n_lim, bias_pre); break;
        default: epilogue_tile<OutT, MI, NI, ACT_NONE, FAST>(a, acc, m0w, n0w, half, l31, split, st, n_lim, bias_pre); break;
    }
}

// (staged epilogue only) a.seg[s] with a run-time s, field by field so that nothing forces the by-value kernel
// argument into scratch
__device__ __forceinline__ OutSeg pick_seg_stg(const GemmArgs& a, int s) {
    OutSeg g;
#define DIMX_PICK(f) g.f = s == 0 ? a.seg[0].f : (s == 1 ? a.seg[1].f : a.seg[2].f)
    DIMX_PICK(ptr);
    DIMX_PICK(sb);
    DIMX_PICK(sh);
    DIMX_PICK(st);
    DIMX_PICK(sd);
    DIMX_PICK(D);
#undef DIMX_PICK
    return g;
}

// ------------------------------------------------------------------------------------------------
// Staged epilogue (STG instantiations only: large-M prefill GEMMs; the decode kernels do not contain this code).
// Each wave dumps a 32-row slice of its accumulators raw (f32) into a private LDS region (the main loop's ring,
// free after a block barrier) and writes it out as 8/16-byte pieces contiguous along the output row; bias /
// activation / positional row / residual run on the row-contiguous data with epilogue_tile's arithmetic.
// ------------------------------------------------------------------------------------------------
// One 32-row slice: raw f32 accumulators of the wave sit in its private LDS region as [32][NI*32]; this writes
// them out.  Row-major segments: a lane owns 4 consecutive columns of a row (bias / activation / positional row /
// residual are applied here on row-contiguous data, then one 8-byte (bf16) or 16-byte (f32) store; 16 or 32 lanes
// cover a full row of the wave tile).  Transposed segments (V^T: contiguous along time): a lane owns 4
// consecutive rows of one column and stores them packed when the launcher proved that legal (vt_pack4).
template <typename OutT, int NI, int ACT, bool FAST>
__device__ __forceinline__ void stage_copy_out(const GemmArgs& a, const unsigned char* stage, int mb, int n0w, int lane) {
    constexpr int ES = sizeof(OutT);
    constexpr int ROWF = NI * 32;      // floats per staged row
    constexpr int PPR = NI * 8;        // 4-column pieces per row
    constexpr int RPI = 64 / PPR;      // rows per wave-wide step
    constexpr int NIT = 32 / RPI;
    const int rowT = a.rowT;
    int b0, t0;
    if (rowT == 1) {
        b0 = mb;
        t0 = 0;
    } else {
        b0 = mb / rowT;
        t0 = mb - b0 * rowT;
    }
    // ---- row-major segments
    {
        const int cc = lane % PPR, crow = lane / PPR;
        const int n = n0w + cc * 4;
        const int nc = n < a.N ? n : 0;
        int s = 0, nn = nc;
        if (a.nseg > 1) {
            s = nc / a.seg_width;
            nn = nc - s * a.seg_width;
        }
        const OutSeg sg = pick_seg_stg(a, s);
        if (n < a.N && sg.sd == 1) {
            const int hh = nn / sg.D, dd = nn - hh * sg.D;
            OutT* const cbase = (OutT*)sg.ptr + (long)hh * sg.sh + dd;
            const float4 bias4 = a.bias ? *(const float4*)(a.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
            int b = b0, t = rowT == 1 ? 0 : t0 + crow;
            if (rowT != 1 && t >= rowT) {
                t -= rowT;
                ++b;
            }
            // the residual pieces of all NIT rows in flight together (one dependent global load per row otherwise: the
            // f32 residual GEMMs of the encoders spent most of their epilogue waiting for them); rows past M read row M - 1
            float4 res[NIT];
            if (a.residual) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    int m = mb + it * RPI + crow;
                    m = m < a.M ? m : a.M - 1;
                    res[it] = *(const float4*)(a.residual + (size_t)m * a.ldr + nc);
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + crow;
                const int m = mb + row;
                if (m < a.M) {
                    float4 v = *(const float4*)(stage + (row * ROWF + cc * 4) * 4);
                    v.x = act_fn<ACT, FAST>(v.x + bias4.x);
                    v.y = act_fn<ACT, FAST>(v.y + bias4.y);
                    v.z = act_fn<ACT, FAST>(v.z + bias4.z);
                    v.w = act_fn<ACT, FAST>(v.w + bias4.w);
                    const int bb = rowT == 1 ? m : b;
                    if (a.rowadd_mode) {
                        const int ri = a.rowadd_mode == 1 ? t : (a.rowadd_mode == 2 ? bb / a.rowadd_div + a.rowadd_off : a.rowadd_off);
                        const float4 q = *(const float4*)(a.rowadd + (size_t)ri * a.ld_rowadd + nc);
                        v.x += q.x * a.rowadd_scale;
                        v.y += q.y * a.rowadd_scale;
                        v.z += q.z * a.rowadd_scale;
                        v.w += q.w * a.rowadd_scale;
                    }
                    if (a.residual) {
                        const float4 q = res[it];
                        v.x += q.x;
                        v.y += q.y;
                        v.z += q.z;
                        v.w += q.w;
                    }
                    OutT* p = cbase + (long)bb * sg.sb + (long)t * sg.st;
                    if (ES == 2) {
                        *(uint2*)p = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                    } else {
                        *(float4*)p = v;
                    }
                }
                if (rowT != 1) {
                    t += RPI;
                    if (t >= rowT) {
                        t -= rowT;
                        ++b;
                    }
                }
            }
        }
    }
    // ---- transposed segments (whole 32-column blocks belong to one segment: seg_width % 32 == 0)
    if (a.nseg > 1) {
        const int col = lane & 31, qh = lane >> 5;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nb = n0w + j * 32;
            if (nb >= a.N) continue;
            const int s = nb / a.seg_width;
            const OutSeg sg = pick_seg_stg(a, s);
            if (sg.sd == 1) continue;  // wave-uniform
            const int n = nb + col;
            const int nc = n < a.N ? n : a.N - 1;
            const int nn = nc - s * a.seg_width;
            const int hh = nn / sg.D, dd = nn - hh * sg.D;
            OutT* const obase = (OutT*)sg.ptr + (long)hh * sg.sh + (long)dd * sg.sd;
            const float bias_v = a.bias ? a.bias[nc] : 0.f;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r0 = (it * 2 + qh) * 4;  // first of 4 consecutive rows
                const int m0r = mb + r0;
                int b = b0, t = t0 + r0;
                if (rowT == 1) {
                    b = m0r;
                    t = 0;
                } else if (t >= rowT) {
                    t -= rowT;
                    ++b;
                }
                float v4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = act_fn<ACT, FAST>(*(const float*)(stage + ((r0 + q) * ROWF + j * 32 + col) * 4) + bias_v);
                    const int m = m0r + q;
                    const int mc = m < a.M ? m : a.M - 1;
                    if (a.rowadd_mode) {
                        int bq = b, tq = t + q;
                        if (rowT != 1 && tq >= rowT) {
                            tq -= rowT;
                            ++bq;
                        }
                        const int ri = a.rowadd_mode == 1 ? tq : (a.rowadd_mode == 2 ? (rowT == 1 ? mc : bq) / a.rowadd_div + a.rowadd_off : a.rowadd_off);
                        v += a.rowadd[(size_t)ri * a.ld_rowadd + nc] * a.rowadd_scale;
                    }
                    if (a.residual) v += a.residual[(size_t)mc * a.ldr + nc];
                    v4[q] = v;
                }
                if (n >= a.N || m0r >= a.M) continue;
                if (a.vt_pack4) {  // 4 consecutive time steps of one clip (rowT % 4 == 0, M % 4 == 0, st == 1)
                    OutT* p = obase + (long)b * sg.sb + (long)t * sg.st;
                    if (ES == 2) {
                        *(uint2*)p = make_uint2(pack_bf16x2(v4[0], v4[1]), pack_bf16x2(v4[2], v4[3]));
                    } else {
                        *(float4*)p = make_float4(v4[0], v4[1], v4[2], v4[3]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (m0r + q >= a.M) continue;
                        int bq = b, tq = t + q;
                        if (rowT == 1) {
                            bq = m0r + q;
                            tq = 0;
                        } else if (tq >= rowT) {
                            tq -= rowT;
                            ++bq;
                        }
                        store_from_f32<OutT>(obase + (long)bq * sg.sb + (long)tq * sg.st, v4[q]);
                    }
                }
            }
        }
    }
}

// slices I .. MI-1 of the wave tile: raw dump (compile-time accumulator indices, immediate LDS offsets), copy-out
template <typename OutT, int I, int MI, int NI, int ACT, bool FAST>
__device__ __forceinline__ void stage_slices(const GemmArgs& a, const f32x16_t (&acc)[MI][NI], int m0w, int n0w,
                                             int half, int l31, unsigned char* stage) {
    if constexpr (I < MI) {
        const int mb = m0w + I * 32;
        if (mb < a.M) {
            float* base = (float*)stage + (4 * half) * (NI * 32) + l31;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * (NI * 32) + j * 32] = acc[I][j][r];
            // the slice is re-read with a different lane<->element mapping (and vector type): keep the compiler from
            // moving those loads above the stores; the LDS itself executes one wave's operations in order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stage_copy_out<OutT, NI, ACT, FAST>(a, stage, mb, n0w, half * 32 + l31);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next slice overwrites
        }
        stage_slices<OutT, I + 1, MI, NI, ACT, FAST>(a, acc, m0w, n0w, half, l31, stage);
    }
}

template <typename OutT, int MI, int NI, int ACT, bool FAST>
__device__ __forceinline__ void epilogue_staged(const GemmArgs& a, const f32x16_t (&acc)[MI][NI], int m0w, int n0w,
                                                int half, int l31, unsigned char* stage) {
    stage_slices<OutT, 0, MI, NI, ACT, FAST>(a, acc, m0w, n0w, half, l31, stage);
}

template <typename T, typename OutT, int MI, int NI>
__device__ __forceinline__ void epilogue_stg(const GemmArgs& a, const f32x16_t (&acc)[MI][NI], int m0w, int n0w, int half,
                                             int l31, unsigned char* stage) {
    constexpr bool FAST = sizeof(T) == 2;
    switch (a.act) {
        case ACT_LEAKY: epilogue_staged<OutT, MI, NI, ACT_LEAKY, FAST>(a, acc, m0w, n0w, half, l31, stage); break;
        case ACT_GELU_TANH: epilogue_staged<OutT, MI, NI, ACT_GELU_TANH, FAST>(a, acc, m0w, n0w, half, l31, stage); break;
        case ACT_GELU_ERF: epilogue_staged<OutT, MI, NI, ACT_GELU_ERF, FAST>(a, acc, m0w, n0w, half, l31, stage); break;
        default: epilogue_staged<OutT, MI, NI, ACT_NONE, FAST>(a, acc, m0w, n0w, half, l31, stage); break;
    }
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, bool CONV>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmArgs a) {
    constexpr int NT = WM * WN * 64;
    constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int BK = 8 * EPC;
    constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + BN) * 128];

    const int tid = threadIdx.x;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ A = (const T*)a.A;
    const T* __restrict__ W = (const T*)a.W;

    // ---- per-thread chunk descriptors
    int a_lds[CA], w_lds[CB];
    size_t a_base[CA], w_base[CB];
    bool a_ok[CA], w_ok[CB];
    int a_kc[CA];
    int a_b[CA], a_t[CA], a_len[CA];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int c = tid + i * NT, row = c >> 3, kc = c & 7;
        const int m = m0 + row;
        a_ok[i] = m < a.M;
        const int mc = a_ok[i] ? m : a.M - 1;
        a_kc[i] = kc * EPC;
        a_lds[i] = lds_off(row, kc);
        if (CONV) {
            a_b[i] = mc / a.conv_T;
            a_t[i] = mc - a_b[i] * a.conv_T;
            int len = a.conv_lens ? a.conv_lens[a_b[i]] : a.conv_T;
            a_len[i] = len < 1 ? 1 : (len > a.conv_T ? a.conv_T : len);
            a_base[i] = 0;
        } else {
            a_base[i] = (size_t)mc * a.lda + kc * EPC;
        }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int c = tid + i * NT, row = c >> 3, kc = c & 7;
        const int n = n0 + row;
        w_ok[i] = n < a.N;
        w_base[i] = (size_t)(w_ok[i] ? n : a.N - 1) * a.ldw + kc * EPC;
        w_lds[i] = BM * 128 + lds_off(row, kc);
    }

    uint4 ra[CA], rw[CB];
    auto gload = [&](int k0) {
        int tap = 0, cbase = k0;
        if (CONV) {
            tap = k0 / a.conv_C;
            cbase = k0 - tap * a.conv_C;
        }
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const bool ok = a_ok[i] && (k0 + a_kc[i] < a.K);
            const T* p;
            if (CONV) {
                int tt = a_t[i] + tap - 2;
                tt = tt < 0 ? 0 : (tt > a_len[i] - 1 ? a_len[i] - 1 : tt);
                p = A + ((size_t)a_b[i] * a.conv_T + tt) * a.lda + cbase + a_kc[i];
            } else {
                p = A + a_base[i] + k0;
            }
            ra[i] = ok ? *(const uint4*)p : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            rw[i] = w_ok[i] ? *(const uint4*)(W + w_base[i] + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CA; ++i) *(uint4*)(&smem[buf][a_lds[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < CB; ++i) *(uint4*)(&smem[buf][w_lds[i]]) = rw[i];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (a.kloop ? a.kloop : a.ldw) / BK;  // W is padded to a multiple of BK
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const unsigned char* sA = &smem[cur][0];
        const unsigned char* sW = &smem[cur][BM * 128];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kc = 2 * ks + half;
            uint4 fa[MI], fw[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *(const uint4*)(sA + lds_off(wm * MI * 32 + i * 32 + l31, kc));
#pragma unroll
            for (int j = 0; j < NI; ++j) fw[j] = *(const uint4*)(sW + lds_off(wn * NI * 32 + j * 32 + l31, kc));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(acc[i][j], fa[i], fw[j]);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    epilogue<T, OutT, MI, NI>(a, acc, m0 + wm * MI * 32, n0 + wn * NI * 32, half, l31, 0);
}


// ------------------------------------------------------------------------------------------------
// Pipelined variant (K % BK == 0, no gather): tiles go global -> LDS by LDS-DMA (global_load_lds,
// 16 B per lane, 1 KiB per wave instruction) into a STAGES-deep ring, STAGES-1 tiles in flight.  The
// wave-uniform destination is linear, so the XOR swizzle is applied on the per-lane SOURCE address.
// Fragment reads are inline-asm ds_read_b128 (the compiler would otherwise drain vmcnt(0) before every
// LDS read while a DMA is pending); tile arrival is ordered by a counted s_waitcnt vmcnt + one raw
// s_barrier per k-tile.  Optional split-K: partial sums are added with global_atomic_add_f32 onto the
// f32 residual stream in place (bf16 perf mode only; the parity mode keeps a fixed summation order).
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int OFF> __device__ __forceinline__ void ds_read128(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }

template <typename T, int MI, int NI> struct FragMma;
template <int MI, int NI> struct FragMma<bf16, MI, NI> {
    static __device__ __forceinline__ void run(f32x16_t (&acc)[MI][NI], const u32x4_t (&fa)[MI], const u32x4_t (&fw)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i]),
                                                                   __builtin_bit_cast(bf16x8_t, fw[j]), acc[i][j], 0, 0, 0);
    }
};
template <int MI, int NI> struct FragMma<float, MI, NI> {
    static __device__ __forceinline__ void run(f32x16_t (&acc)[MI][NI], const u32x4_t (&fa)[MI], const u32x4_t (&fw)[NI]) {
        // NB: cast the whole 128-bit fragment first, then index (per-element bit_cast of an asm output
        // vector miscompiles to element 0 on hipcc 7.2)
        f32x4_t a4[MI], w4[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a4[i] = __builtin_bit_cast(f32x4_t, fa[i]);
#pragma unroll
        for (int j = 0; j < NI; ++j) w4[j] = __builtin_bit_cast(f32x4_t, fw[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][0], w4[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][1], w4[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][2], w4[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][3], w4[j][3], acc[i][j], 0, 0, 0);
            }
    }
};

// ABL (ablation, tuning only): 0 normal, 1 = no MFMA/ds_read in the loop (DMA only), 2 = no DMA in the loop
// KPI: k-tiles per loop iteration (one vmcnt wait + one barrier per KPI tiles; the ring holds STAGES groups of KPI tiles).
// The decode GEMMs (M = batch) do 4 MFMAs per wave and k-tile, so the per-iteration wait / barrier / DMA-issue overhead
// (~1000 cycles) is what they cost: KPI = 2 halves the number of iterations.
template <typename T, typename OutT, int BM, int BN, int WM, int WN, int STAGES, int ABL = 0, bool STG = false, int KPI = 1>
__global__ __launch_bounds__(WM* WN * 64) void gemm_glds_kernel(const GemmArgs a) {
    constexpr int NW = WM * WN;
    constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int BK = 8 * EPC;
    constexpr int LA = BM / 8 / NW, LW = BN / 8 / NW;  // wave-wide DMA instructions per tile per wave
    constexpr int LPT = LA + LW;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row DMA pieces per wave");
    static_assert(STAGES >= 1 && (STAGES - 1) * LPT * KPI < 64, "vmcnt range");
    static_assert(KPI == 1 || (STAGES >= 2 && ABL == 0 && !STG), "KPI > 1: plain ring only");
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * KPI * TILE_BYTES];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    // XCD-aware order: consecutive block ids land on different XCDs (8 private L2s), so give every XCD a
    // contiguous run of tiles -- tiles that share an A row panel then hit the same L2 (speed only).
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int split = bid / ntiles, tile = bid - split * ntiles;
    // few row tiles (decode: M = batch): n-major order, so that the blocks sharing a W column tile are neighbours
    // and land on the same XCD -- its L2 then fetches every weight byte once instead of once per row tile (the A
    // panel is tiny and stays resident everywhere).  Many row tiles (prefill): m-major, A panels are the big operand.
    int tile_m, tile_n;
    if (a.tile_map == 1) {
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int tq = (tiles_m + 3) >> 2, tnh = tiles_n >> 1;
        const int im = i / tnh;
        tile_m = (x >> 1) * tq + im;
        tile_n = (x & 1) * tnh + (i - im * tnh);
        if (tile_m >= tiles_m) return;
    } else if (tiles_m <= 8) {
        tile_n = tile / tiles_m;
        tile_m = tile - tile_n * tiles_m;
    } else {
        tile_m = tile / tiles_n;
        tile_n = tile - tile_m * tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ A = (const T*)a.A;
    const T* __restrict__ W = (const T*)a.W;

    // k-tiles of this split
    const int nk_all = (a.kloop ? a.kloop : a.ldw) / BK;
    const int per = (nk_all + a.splitk - 1) / a.splitk;
    const int kt0 = split * per;
    int nk = nk_all - kt0;
    nk = nk > per ? per : nk;
    if (nk <= 0) return;

    // DMA piece j of this wave covers tile rows (wave*L + j)*8 .. +7; lane l -> row + (l>>3), slot l&7
    const T* gA[LA];
    const T* gW[LW];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int row = (wave * LA + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        gA[j] = A + (size_t)m * a.lda + c * EPC + (size_t)kt0 * BK;
    }
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int row = (wave * LW + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int n = n0 + row;
        n = n < a.N ? n : a.N - 1;
        gW[j] = W + (size_t)n * a.ldw + c * EPC + (size_t)kt0 * BK;
    }
    auto issue = [&](int kt, int buf) {
        unsigned char* base = smem + buf * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < LA; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gA[j] + (size_t)kt * BK),
                                             (lds_void_t*)(base + (wave * LA + j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < LW; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gW[j] + (size_t)kt * BK),
                                             (lds_void_t*)(base + BM * 128 + (wave * LW + j) * 1024), 16, 0, 0);
    };

    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    unsigned aoff[4], woff[4];
    {
        const int ra = wm * MI * 32 + l31, rw = wn * NI * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            aoff[ks] = lds0 + lds_off(ra, 2 * ks + half);
            woff[ks] = lds0 + BM * 128 + lds_off(rw, 2 * ks + half);
        }
    }

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one k-tile from ring slot `slot`: all fragment reads of a group of k-steps are issued back to back; the MFMAs of
    // k-step s start as soon as the reads of steps <= s have returned (LDS returns in order: counted lgkmcnt)
    auto compute_tile = [&](int slot) {
        const unsigned boff = (unsigned)(slot * TILE_BYTES);
        constexpr int RPK = MI + NI;                      // ds_read_b128 per k-step
        constexpr int GROUP = (4 * RPK <= 12) ? 4 : (2 * RPK <= 12 ? 2 : 1);  // k-steps per group (lgkmcnt: 4 bits)
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += GROUP) {
            u32x4_t fa[GROUP][MI], fw[GROUP][NI];
#pragma unroll
            for (int q = 0; q < GROUP; ++q) {
                const unsigned pa = aoff[g0 + q] + boff, pw = woff[g0 + q] + boff;
                ds_read128<0>(fa[q][0], pa);
                if (MI > 1) ds_read128<4096>(fa[q][MI > 1 ? 1 : 0], pa);
                if (MI > 2) ds_read128<8192>(fa[q][MI > 2 ? 2 : 0], pa);
                if (MI > 3) ds_read128<12288>(fa[q][MI > 3 ? 3 : 0], pa);
                ds_read128<0>(fw[q][0], pw);
                if (NI > 1) ds_read128<4096>(fw[q][NI > 1 ? 1 : 0], pw);
                if (NI > 2) ds_read128<8192>(fw[q][NI > 2 ? 2 : 0], pw);
                if (NI > 3) ds_read128<12288>(fw[q][NI > 3 ? 3 : 0], pw);
            }
#pragma unroll
            for (int q = 0; q < GROUP; ++q) {
                if (q == 0) wait_lgkm<(GROUP - 1) * RPK>();
                if (q == 1) wait_lgkm<(GROUP - 2) * RPK>();
                if (q == 2) wait_lgkm<(GROUP > 2 ? GROUP - 3 : 0) * RPK>();
                if (q == 3) wait_lgkm<0>();
                __builtin_amdgcn_sched_barrier(0);
                FragMma<T, MI, NI>::run(acc, fa[q], fw[q]);
            }
        }
    };

    if constexpr (KPI > 1) {
        // groups of KPI k-tiles: group gi = tiles [gi*KPI, gi*KPI + KPI) in ring slots (gi % STAGES)*KPI + j
        const int ng = (nk + KPI - 1) / KPI;
        auto issue_group = [&](int gi) {
#pragma unroll
            for (int jj = 0; jj < KPI; ++jj) {
                const int kt = gi * KPI + jj;
                issue(kt < nk ? kt : nk - 1, (gi % STAGES) * KPI + jj);  // a short last group re-reads the last tile
            }
        };
#pragma unroll
        for (int s2 = 0; s2 < STAGES - 1; ++s2)
            if (s2 < ng) issue_group(s2);
        for (int gi = 0; gi < ng; ++gi) {
            if (ng - 1 - gi < STAGES - 2)
                wait_vmcnt<0>();
            else
                wait_vmcnt<(STAGES - 2) * LPT * KPI>();
            __builtin_amdgcn_s_barrier();
            if (gi + STAGES - 1 < ng) issue_group(gi + STAGES - 1);
#pragma unroll
            for (int jj = 0; jj < KPI; ++jj)
                if (gi * KPI + jj < nk) compute_tile((gi % STAGES) * KPI + jj);
        }
    } else {
    auto stamp = [&](int i) {
        if (ABL == 3 && tid == 0 && i < 32) a.prof[(size_t)blockIdx.x * 32 + i] = wall_clock64();
    };
    stamp(0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk && (ABL != 2 || s == 0)) issue(s, s);
    stamp(1);

    for (int it = 0; it < nk; ++it) {
        if (ABL == 3 && it < 24) stamp(2 + it);
        if (STAGES == 1) {
            // single LDS buffer: latency is hidden by the other resident blocks (up to 5 per CU), not by a ring
            if (it > 0) __builtin_amdgcn_s_barrier();  // everyone is done reading the previous tile
            issue(it, 0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        } else {
            // tile `it` has landed once at most min(STAGES-2, nk-1-it) younger tiles are still in flight
            if (ABL == 2 || nk - 1 - it < STAGES - 2)
                wait_vmcnt<0>();
            else
                wait_vmcnt<(STAGES >= 2 ? STAGES - 2 : 0) * LPT>();
            if (ABL == 3 && it == 8) stamp(20);
            __builtin_amdgcn_s_barrier();
            if (ABL == 3 && it == 8) stamp(21);
            // Measured on this sequence (tools/gemm_phases.py, tools/gemm_ab.py, same box): wave 0 spends 0.07 us in the
            // vmcnt wait, 0.04 in the barrier, 0.14 issuing its 4 DMA pieces, 0.15 on 8 ds_reads + 4 MFMAs.  Issuing the
            // fragment reads before the DMAs: qkv 10.3 -> 11.2 us; non-temporal DMA loads of W: ff1 13.5 -> 15.9 us (every
            // M tile re-reads the W tile from L2); a wave-uniform branch per DMA piece: +1 us per kernel.
            if (ABL != 2 && it + STAGES - 1 < nk) issue(it + STAGES - 1, (it + STAGES - 1) % STAGES);
            if (ABL == 3 && it == 8) stamp(22);
        }
        if (ABL == 1) continue;
        compute_tile(it % STAGES);
        if (ABL == 3 && it == 8) stamp(23);
    }
    stamp(28);
    }

    // ---- epilogue (split-K: this split's partial sums go to its own f32 slab)
    if constexpr (STG) {
        constexpr int SLICE = 32 * NI * 32 * 4;  // raw f32 slice per wave
        static_assert(SLICE * NW <= STAGES * TILE_BYTES, "staging slices must fit in the ring");
        __syncthreads();  // every wave is done reading the ring (all DMA landed before the last k-tile)
        epilogue_stg<T, OutT, MI, NI>(a, acc, m0 + wm * MI * 32, n0 + wn * NI * 32, half, l31, smem + wave * SLICE);
    } else {
        epilogue<T, OutT, MI, NI>(a, acc, m0 + wm * MI * 32, n0 + wn * NI * 32, half, l31, split);
    }
    if (ABL == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) a.prof[(size_t)blockIdx.x * 32 + 29] = wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------------
// Decode GEMM (M = batch, a few 64-row tiles): the same 64 x 64 tile, ring and swizzle as gemm_glds_kernel, but the
// loop's two halves run on different waves.  In the 4-wave kernel a wave does wait -> barrier -> 4 DMA issues ->
// 8 ds_reads -> 4 MFMAs in sequence, and with ONE wave per SIMD nothing overlaps them: 0.385 us per 16 KiB k-tile,
// of which 0.14 is DMA issue and 0.15 fragment reads + MFMA.  Here waves 0-3 only consume (barrier -> reads -> MFMA)
// and waves 4-7 only load (wait for their pieces of tile t -> barrier -> issue tile t + STAGES - 1); a SIMD holds
// one of each, so its loader's issue time hides behind its consumer's matrix time.  One s_barrier per k-tile does both
// hand-offs: tile t is complete (every loader waited for its own pieces before arriving) and slot (t - 1) % STAGES is
// free (every consumer finished its reads of tile t - 1 before arriving).
// ------------------------------------------------------------------------------------------------
//
// (Replacing the barrier by two LDS counters -- loaders add to `landed`, consumers to `consumed`, each side polls the
// other's -- so that loaders never wait for each other measured slower: qkv 8.6 -> 10.5 us, the polls sit on both
// critical paths.)
// ABLW (tuning only, tools/gemm_ab.py): 1 = consumers skip reads + MFMA, 2 = loaders issue no DMA after the prologue,
// 3 = the W pieces are replaced by a second copy of the A pieces (every DMA an L2 hit).
// The body as a device function (the block's index and the block count come as arguments, the LDS from the caller) so that
// the attention / GEMM co-residency probe below can run it next to decode-attention blocks in one launch.
template <typename T, typename OutT, int STAGES, bool PROF, int ABLW>
__device__ __forceinline__ void gemm_ws_body(const GemmArgs& a, int block_id, int nblocks, unsigned char* smem, float* ln_sm) {
    constexpr int BM = 64, BN = 64, NL = 4;
    constexpr int EPC = Elem<T>::kPerChunk;
    constexpr int BK = 8 * EPC;
    constexpr int LA = BM / 8 / NL, LW = BN / 8 / NL;  // DMA pieces per tile per loader wave
    constexpr int LPT = LA + LW;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    static_assert(STAGES >= 3 && (STAGES - 1) * LPT < 64, "ring depth");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int bid = block_id;  // XCD-aware order, n-major for few row tiles: see gemm_glds_kernel
    {
        const int nblk = nblocks, q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int split = bid / ntiles, tile = bid - split * ntiles;
    int tile_m, tile_n;
    if (tiles_m <= 8) {
        tile_n = tile / tiles_m;
        tile_m = tile - tile_n * tiles_m;
    } else {
        tile_m = tile / tiles_n;
        tile_n = tile - tile_m * tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk_all = (a.kloop ? a.kloop : a.ldw) / BK;
    const int per = (nk_all + a.splitk - 1) / a.splitk;
    const int kt0 = split * per;
    int nk = nk_all - kt0;
    nk = nk > per ? per : nk;
    if (nk <= 0) return;
    // PROF (tools/gemm_phases.py): 64 wall-clock stamps per block, 0-31 by consumer wave 0, 32-63 by loader wave 4
    auto stamp = [&](int i) {
        if (PROF && lane == 0 && (wave == 0 || wave == 4) && i < 32)
            a.prof[(size_t)block_id * 64 + (wave == 4 ? 32 : 0) + i] = wall_clock64();
    };
    stamp(0);

    if (wave >= 4) {
        // ---------------- loader waves
        const int lw = wave - 4;
        const T* __restrict__ A = (const T*)a.A;
        const T* __restrict__ W = (const T*)a.W;
        const T* gA[LA];
        const T* gW[LW];
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int row = (lw * LA + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            gA[j] = A + (size_t)m * a.lda + c * EPC + (size_t)kt0 * BK;
        }
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int row = (lw * LW + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int n = n0 + row;
            n = n < a.N ? n : a.N - 1;
            gW[j] = W + (size_t)n * a.ldw + c * EPC + (size_t)kt0 * BK;
            if (a.w_tiled)  // block (n / 8, k-tile) = 8 rows x 128 B, contiguous; the k-tiles of a row group follow each other
                gW[j] = W + ((size_t)(n >> 3) * (a.ldw / BK) + kt0) * (8 * BK) + (n & 7) * BK + c * EPC;
            if (ABLW == 3) gW[j] = gA[j < LA ? j : 0];
        }
        const int wstep = a.w_tiled ? 8 * BK : BK;  // elements from one k-tile of a piece to the next
        auto issue = [&](int kt, int buf) {
            unsigned char* base = smem + buf * TILE_BYTES;
#pragma unroll
            for (int j = 0; j < LA; ++j)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gA[j] + (size_t)kt * BK),
                                                 (lds_void_t*)(base + (lw * LA + j) * 1024), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < LW; ++j)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gW[j] + (size_t)kt * wstep),
                                                 (lds_void_t*)(base + BM * 128 + (lw * LW + j) * 1024), 16, 0, 0);
        };
        // Deferred LayerNorm of the A rows (GemmArgs.ln_stats): the LOADER waves reduce the 32 per-CU partial sums of the tile's
        // 64 rows -- loader lw takes rows lw * 16 + (lane & 15), lane >> 4 picks 8 of the 32 CUs.  The loads are this wave's
        // oldest memory operations (in-order return: every counted vmcnt wait below covers them), the reduction runs after the
        // wave's last barrier but one, when it has nothing left to issue; the consumers find {mean, rstd} in LDS after the loop.
        float2 lp[8];
        if (a.ln_stats) {
            int m = m0 + lw * 16 + (lane & 15);
            m = m < a.M ? m : a.M - 1;
            const float2* sp = (const float2*)a.ln_stats + ((size_t)(m >> 5) * 32 + (lane >> 4) * 8) * 32 + (m & 31);
#pragma unroll
            for (int i = 0; i < 8; ++i) lp[i] = sp[(size_t)i * 32];
        }
#pragma unroll
        for (int st = 0; st < STAGES - 1; ++st)
            if (st < nk) issue(st, st);
        stamp(1);
        for (int it = 0; it < nk; ++it) {
            if (it < 12) stamp(2 + 2 * it);
            // this wave's pieces of tile `it` have landed once at most min(STAGES-2, nk-1-it) younger tiles are in flight
            if (nk - 1 - it < STAGES - 2)
                wait_vmcnt<0>();
            else
                wait_vmcnt<(STAGES - 2) * LPT>();
            if (it < 12) stamp(3 + 2 * it);
            __builtin_amdgcn_s_barrier();
            if (ABLW != 2 && it + STAGES - 1 < nk) issue(it + STAGES - 1, (it + STAGES - 1) % STAGES);
            if (a.ln_stats && it == nk - 2) {  // vmcnt is 0 here (the wait of this iteration), nothing left to issue
                // {sum x, M2} of 32 column slices -> mean and variance by the parallel-variance formula, two passes (no
                // sum x^2 - mean^2 cancellation)
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
                    ln_sm[lw * 16 + lane] = mean;
                    ln_sm[64 + lw * 16 + lane] = rstd;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in LDS before this wave arrives at the last barrier
            }
        }
        stamp(28);
        return;
    }

    // ---------------- consumer waves: 2 x 2, one 32 x 32 accumulator each
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        aoff[ks] = lds0 + lds_off(wm * 32 + l31, 2 * ks + half);
        woff[ks] = lds0 + BM * 128 + lds_off(wn * 32 + l31, 2 * ks + half);
    }
    f32x16_t acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    // deferred LayerNorm of the A rows: {mean, rstd} come from the loader waves through LDS (above); the lane's column sum of
    // the gamma-scaled weights is requested once the prologue tiles have landed
    float ln_cs = 0.f;
    for (int it = 0; it < nk; ++it) {
        const unsigned boff = (unsigned)((it % STAGES) * TILE_BYTES);
        if (it < 12) stamp(2 + 2 * it);
        __builtin_amdgcn_s_barrier();
        if (ABLW == 1) continue;
        if (it == 0 && a.ln_stats) {
            const int n = n0 + wn * 32 + l31;
            ln_cs = a.ln_colsum[n < a.N ? n : a.N - 1];
        }
        if (it < 12) stamp(3 + 2 * it);
        u32x4_t fa[4][1], fw[4][1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ds_read128<0>(fa[q][0], aoff[q] + boff);
            ds_read128<0>(fw[q][0], woff[q] + boff);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 0) wait_lgkm<6>();
            if (q == 1) wait_lgkm<4>();
            if (q == 2) wait_lgkm<2>();
            if (q == 3) wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            FragMma<T, 1, 1>::run(acc, fa[q], fw[q]);
        }
    }
    stamp(28);
    if (a.ln_stats) {
        // an accumulator register group holds rows 8q + 4 half + (0..3) of the wave's 32: 8 ds_read_b128
        const float* strip = ln_sm + wm * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 mq = *(const float4*)(strip + 8 * q + 4 * half);
            const float4 rq = *(const float4*)(strip + 64 + 8 * q + 4 * half);
            acc[0][0][4 * q + 0] = rq.x * (acc[0][0][4 * q + 0] - mq.x * ln_cs);
            acc[0][0][4 * q + 1] = rq.y * (acc[0][0][4 * q + 1] - mq.y * ln_cs);
            acc[0][0][4 * q + 2] = rq.z * (acc[0][0][4 * q + 2] - mq.z * ln_cs);
            acc[0][0][4 * q + 3] = rq.w * (acc[0][0][4 * q + 3] - mq.w * ln_cs);
        }
    }
    epilogue<T, OutT, 1, 1>(a, acc, m0 + wm * 32, n0 + wn * 32, half, l31, split,
                            (PROF && wave == 0 && lane == 0) ? a.prof + (size_t)block_id * 64 : nullptr);
    if (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(29);
    }
}

template <typename T, typename OutT, int STAGES, bool PROF = false, int ABLW = 0>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const GemmArgs a) {
    constexpr int TILE_BYTES = (64 + 64) * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float ln_sm[128];  // deferred LayerNorm: mean[64], rstd[64] of the tile's rows
    gemm_ws_body<T, OutT, STAGES, PROF, ABLW>(a, blockIdx.x, gridDim.x, smem, ln_sm);
}

// ------------------------------------------------------------------------------------------------
// gemm_ws72_kernel (round 5): the loader/consumer decode GEMM on 64 x 72 tiles -- ONE block per CU, exactly.
// Measured (tools/attic/r05_gemm_blocks.py, profiles/r05_gemm_blocks.txt): the three chip-wide decode GEMMs launch 288 blocks of
// 64 x 64 tiles on 256 CUs; a CU that hosts two blocks pulls twice the operand bytes through its LDS-DMA path (55-67 GB/s
// per CU whatever the block count) and the launch lasts as long as those 32 CUs: the same kernel at a column count that gives
// 256 blocks runs 7.7 instead of 11.1 us (K = 4608, 4 slabs), 9.2 instead of 10.8 (ff1), 6.3 instead of 6.6 (qkv).  No
// 64-column tiling of N = 4608 / 1152 / 2304 at M = 256 gives 256 equal blocks (9 | N / 128); 72 = N / 64, N / 16, N / 32 does:
// 4 row tiles x {64, 16, 32} column tiles x {1, 4, 2} K splits = 256 blocks with the SAME k-tile counts (18 / 18 / 9) and the
// same slab counts as before.  A 72-column tile is three 32-column MFMA blocks of which the third holds 8 valid columns: six
// consumer waves (2 x 3, one 32 x 32 accumulator each, the loop body of gemm_ws_kernel unchanged), four loader waves; a k-tile
// is 8 A pieces + 9 W pieces of 1 KiB (17 KiB against 16: loader 0 issues five pieces per tile, the others four -- the counted
// vmcnt waits are instantiated per loader).  The W fragment reads of the third column block run past the 72 rows that were
// loaded (into the next ring slot / the slack behind the ring): whatever they find only reaches accumulator columns that are
// never stored (n_lim).  bf16 operands only; M and N edges are clamped like in gemm_ws_kernel.
// KT (round 5, second step): k-tiles per ring slot = per block barrier.  KT = 2: three slots of two k-tiles; the consumers request
// the fragments of a slot's second k-tile before the first one's MFMAs start (its LDS latency disappears behind them) and meet
// the loaders half as often.
template <typename OutT, bool PROF, int KT>
struct Ws72 {
    static constexpr int BM = 64, BN = 72, BK = 64, STAGES = KT == 1 ? 4 : 3, NLOAD = 4, NCONS = 6;
    static constexpr int TILE_BYTES = (BM + BN) * 128;
    static constexpr int SLOT_BYTES = KT * TILE_BYTES;
    // + the rows 72..95 read behind the last slot; and never less than 84 KiB: two blocks must NOT fit a CU.  Launched into an idle
    // GPU the 256 blocks land one per CU, but behind another kernel the dispatcher hands a CU that drained early a second block
    // while others still wait for their first (LDS and waves would allow it) -- and the doubled CUs set the launch time again:
    // 8.8 us by in-kernel stamps (isolated launches) against 10.4 us back to back (profiles/r05_gemm_ws72.txt)
    static constexpr int RING_BYTES = STAGES * SLOT_BYTES + 24 * 128;
    static constexpr int SMEM_BYTES = RING_BYTES > 84 * 1024 ? RING_BYTES : 84 * 1024;
    static_assert((STAGES - 2) * KT == 2, "the loaders' counted waits below are written for two k-tiles in flight behind the current slot");

    template <int LW>  // W pieces of this loader wave (3 for loader 0, 2 for the others)
    static __device__ __forceinline__ void loader(const GemmArgs& a, int block_id, int lw, int lane, int m0, int n0, int kt0, int nk,
                                                  unsigned char* smem, float* ln_sm) {
        constexpr int LA = 2, LPT = LA + LW;
        static_assert((STAGES - 1) * KT * LPT < 56, "ring depth");
        auto stamp = [&](int i) {
            if (PROF && lane == 0 && lw == 0 && i < 32) a.prof[(size_t)block_id * 64 + 32 + i] = wall_clock64();
        };
        const bf16* __restrict__ A = (const bf16*)a.A;
        const bf16* __restrict__ W = (const bf16*)a.W;
        const bf16* gA[LA];
        const bf16* gW[LW];
        int pA[LA], pW[LW];
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            pA[j] = lw * LA + j;
            const int row = pA[j] * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            gA[j] = A + (size_t)m * a.lda + c * 8 + (size_t)kt0 * BK;
        }
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            pW[j] = (lw == 0 ? 0 : 1 + 2 * lw) + j;
            const int row = pW[j] * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int n = n0 + row;
            n = n < a.N ? n : a.N - 1;
            gW[j] = W + (size_t)n * a.ldw + c * 8 + (size_t)kt0 * BK;
        }
        const int nst = (nk + KT - 1) / KT;
        auto issue = [&](int st, int slot) {  // the k-tiles of slot st that exist (the last slot of an odd count holds one)
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const int kt = st * KT + t;
                if (kt < nk) {
                    unsigned char* base = smem + slot * SLOT_BYTES + t * TILE_BYTES;
#pragma unroll
                    for (int j = 0; j < LA; ++j)
                        __builtin_amdgcn_global_load_lds((glb_void_t*)(gA[j] + (size_t)kt * BK), (lds_void_t*)(base + pA[j] * 1024), 16, 0, 0);
#pragma unroll
                    for (int j = 0; j < LW; ++j)
                        __builtin_amdgcn_global_load_lds((glb_void_t*)(gW[j] + (size_t)kt * BK), (lds_void_t*)(base + BM * 128 + pW[j] * 1024),
                                                         16, 0, 0);
                }
            }
        };
        // deferred LayerNorm of the A rows: as in gemm_ws_body (loader lw reduces the partial sums of rows lw * 16 + (lane & 15))
        float2 lp[8];
        if (a.ln_stats) {
            int m = m0 + lw * 16 + (lane & 15);
            m = m < a.M ? m : a.M - 1;
            const float2* sp = (const float2*)a.ln_stats + ((size_t)(m >> 5) * 32 + (lane >> 4) * 8) * 32 + (m & 31);
#pragma unroll
            for (int i = 0; i < 8; ++i) lp[i] = sp[(size_t)i * 32];
        }
        int issued = 0;
        for (; issued < STAGES - 1 && issued < nst; ++issued) issue(issued, issued);
        stamp(1);
        int slot_next = issued % STAGES;
        const int st_ln = nst >= 2 ? nst - 2 : 0;
        for (int st = 0; st < nst; ++st) {
            if (st < 12) stamp(2 + 2 * st);
            {   // this wave's pieces of slot st have landed once no more than the pieces of the k-tiles issued BEHIND it are in flight
                int done = (st + 1) * KT, all = issued * KT;
                done = done < nk ? done : nk;
                all = all < nk ? all : nk;
                const int y = all - done;  // 0, 1 or 2 k-tiles
                if (y <= 0) wait_vmcnt<0>();
                else if (y == 1) wait_vmcnt<LPT>();
                else wait_vmcnt<2 * LPT>();
            }
            if (st < 12) stamp(3 + 2 * st);
            __builtin_amdgcn_s_barrier();
            if (issued < nst) {
                issue(issued, slot_next);
                ++issued;
                slot_next = slot_next + 1 == STAGES ? 0 : slot_next + 1;
            }
            if (a.ln_stats && st == st_ln) {
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
                if (a.ln_err && lane < 16 && m0 + lw * 16 + lane < a.M && mean * mean > 64.0f * var) atomicOr(a.ln_err, 4u);
                if (lane < 16) {
                    ln_sm[lw * 16 + lane] = mean;
                    ln_sm[64 + lw * 16 + lane] = rstd;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        stamp(28);
    }

    static __device__ __forceinline__ void body(const GemmArgs& a, int block_id, int nblocks, unsigned char* smem, float* ln_sm) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
        const int ntiles = tiles_m * tiles_n;
        int bid = block_id;  // XCD-aware order: an XCD's blocks are a contiguous range of n-major tiles (the row tiles of a column
                             // tile share its W rows through that XCD's L2)
        {
            const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, i = bid >> 3;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        }
        const int split = bid / ntiles, tile = bid - split * ntiles;
        const int tile_n = tile / tiles_m, tile_m = tile - tile_n * tiles_m;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const int nk_all = (a.kloop ? a.kloop : a.ldw) / BK;
        const int per = (nk_all + a.splitk - 1) / a.splitk;
        const int kt0 = split * per;
        int nk = nk_all - kt0;
        nk = nk > per ? per : nk;
        if (nk <= 0) return;
        auto stamp = [&](int i) {
            if (PROF && lane == 0 && wave == 0 && i < 32) a.prof[(size_t)block_id * 64 + i] = wall_clock64();
        };
        stamp(0);
        if (wave >= NCONS) {
            const int lw = wave - NCONS;
            if (lw == 0)
                loader<3>(a, block_id, lw, lane, m0, n0, kt0, nk, smem, ln_sm);
            else
                loader<2>(a, block_id, lw, lane, m0, n0, kt0, nk, smem, ln_sm);
            return;
        }
        // ---------------- consumer waves: 2 x 3, one 32 x 32 accumulator each
        const int wm = wave / 3, wn = wave - 3 * wm;
        const int half = lane >> 5, l31 = lane & 31;
        const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
        unsigned aoff[4], woff[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            aoff[ks] = lds0 + lds_off(wm * 32 + l31, 2 * ks + half);
            woff[ks] = lds0 + BM * 128 + lds_off(wn * 32 + l31, 2 * ks + half);
        }
        f32x16_t acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
        // Everything the epilogue needs is fetched NOW -- the consumers wait ~1.5 us for the first tile anyway: the lane's bias and
        // (deferred LayerNorm) column sum (as dependent loads behind the main loop they cost 0.5 us per launch), and the kernel
        // arguments of the store (destination, row stride, slab offset, activation: behind the loop their scalar loads were another
        // 0.4 us, tools/attic/r05_gemm_stamps.py).  The kernel only takes plain row-major destinations (gemm_use_ws72), so the epilogue is
        // 16 stores per lane and nothing else.
        const int n_lim = n0 + BN < a.N ? n0 + BN : a.N;
        const int ncol = n0 + wn * 32 + l31;
        const bool col_ok = ncol < n_lim;
        const int ncl = col_ok ? ncol : n_lim - 1;
        const float bias_v = (a.bias && split == 0) ? a.bias[ncl] : 0.f;
        const float ln_cs = a.ln_stats ? a.ln_colsum[ncl] : 0.f;
        const int ldc = (int)a.seg[0].st;
        const int row_base = m0 + wm * 32 + 4 * half;
        OutT* cptr = (OutT*)a.seg[0].ptr + (a.out_slabs ? (size_t)split * a.slab_stride : 0) + (size_t)row_base * ldc + ncl;
        const float* rptr = a.residual ? a.residual + (size_t)row_base * a.ldr + ncl : nullptr;
        const int ldr = a.ldr;
        const int mrem = a.M - row_base;
        const int act = a.act;
        const bool has_ln = a.ln_stats != nullptr;
        const int nst = (nk + KT - 1) / KT;
        unsigned boff = 0;
        for (int st = 0; st < nst; ++st) {
            if (st < 12) stamp(2 + 2 * st);
            __builtin_amdgcn_s_barrier();
            if (st < 12) stamp(3 + 2 * st);
            u32x4_t fa[KT][4][1], fw[KT][4][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ds_read128<0>(fa[0][q][0], aoff[q] + boff);
                ds_read128<0>(fw[0][q][0], woff[q] + boff);
            }
            const bool two = KT == 2 && st * KT + 1 < nk;  // wave-uniform
            if (two) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ds_read128<0>(fa[KT - 1][q][0], aoff[q] + boff + TILE_BYTES);
                    ds_read128<0>(fw[KT - 1][q][0], woff[q] + boff + TILE_BYTES);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == 0) wait_lgkm<14>();
                    if (q == 1) wait_lgkm<12>();
                    if (q == 2) wait_lgkm<10>();
                    if (q == 3) wait_lgkm<8>();
                    __builtin_amdgcn_sched_barrier(0);
                    FragMma<bf16, 1, 1>::run(acc, fa[0][q], fw[0][q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == 0) wait_lgkm<6>();
                    if (q == 1) wait_lgkm<4>();
                    if (q == 2) wait_lgkm<2>();
                    if (q == 3) wait_lgkm<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    FragMma<bf16, 1, 1>::run(acc, fa[KT - 1][q], fw[KT - 1][q]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == 0) wait_lgkm<6>();
                    if (q == 1) wait_lgkm<4>();
                    if (q == 2) wait_lgkm<2>();
                    if (q == 3) wait_lgkm<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    FragMma<bf16, 1, 1>::run(acc, fa[0][q], fw[0][q]);
                }
            }
            boff = boff + SLOT_BYTES == STAGES * SLOT_BYTES ? 0u : boff + SLOT_BYTES;
        }
        stamp(28);
        if (has_ln) {
            const float* strip = ln_sm + wm * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 mq = *(const float4*)(strip + 8 * q + 4 * half);
                const float4 rq = *(const float4*)(strip + 64 + 8 * q + 4 * half);
                acc[0][0][4 * q + 0] = rq.x * (acc[0][0][4 * q + 0] - mq.x * ln_cs);
                acc[0][0][4 * q + 1] = rq.y * (acc[0][0][4 * q + 1] - mq.y * ln_cs);
                acc[0][0][4 * q + 2] = rq.z * (acc[0][0][4 * q + 2] - mq.z * ln_cs);
                acc[0][0][4 * q + 3] = rq.w * (acc[0][0][4 * q + 3] - mq.w * ln_cs);
            }
        }
        if (PROF && wave == 0 && lane == 0) a.prof[(size_t)block_id * 64 + 24] = wall_clock64();
        float v[16];
        switch (act) {
            case ACT_GELU_ERF:
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = act_fn<ACT_GELU_ERF, true>(acc[0][0][r] + bias_v);
                break;
            case ACT_GELU_TANH:
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = act_fn<ACT_GELU_TANH, true>(acc[0][0][r] + bias_v);
                break;
            case ACT_LEAKY:
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = act_fn<ACT_LEAKY, true>(acc[0][0][r] + bias_v);
                break;
            default:
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[0][0][r] + bias_v;
                break;
        }
        if (PROF && wave == 0 && lane == 0) a.prof[(size_t)block_id * 64 + 25] = wall_clock64();
        if (col_ok) {
            if (rptr) {  // all residual loads in flight together; rows past M read the last valid row (never stored)
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int off = (r & 3) + 8 * (r >> 2);
                    off = off < mrem ? off : mrem - 1;
                    rv[r] = rptr[(size_t)off * ldr];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += rv[r];
            }
            if (mrem > 27) {  // the lane's last row is row_base + 27: no per-row checks for interior tiles
#pragma unroll
                for (int r = 0; r < 16; ++r) store_from_f32<OutT>(cptr + (size_t)((r & 3) + 8 * (r >> 2)) * ldc, v[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = (r & 3) + 8 * (r >> 2);
                    if (off < mrem) store_from_f32<OutT>(cptr + (size_t)off * ldc, v[r]);
                }
            }
        }
        if (PROF) {
            if (wave == 0 && lane == 0) a.prof[(size_t)block_id * 64 + 26] = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(29);
        }
    }
};

template <typename OutT, bool PROF = false, int KT = 2>
__global__ __launch_bounds__(640) void gemm_ws72_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[Ws72<OutT, PROF, KT>::SMEM_BYTES];
    __shared__ __attribute__((aligned(16))) float ln_sm[128];
    Ws72<OutT, PROF, KT>::body(a, blockIdx.x, gridDim.x, smem, ln_sm);
}

template <typename OutT> static int launch_ws72(const GemmArgs& a, hipStream_t s) {
    const int tiles = ceil_div(a.M, 64) * ceil_div(a.N, 72);
    static const int kt_env = getenv("DIMX_WS72_KT") ? atoi(getenv("DIMX_WS72_KT")) : 1;  // k-tiles per barrier; 2 measured SLOWER (1624 vs 1653 clips/s, profiles/r05_gemm_ws72.txt)
    // the deferred-LayerNorm epilogue writes ln_sm two stages before the last barrier: with two k-tiles per barrier a block
    // needs at least two stages (4 k-tiles) for that barrier to exist (K = 128 raced); shorter blocks keep KT = 1
    const int nk_blk = a.K / 64 / (a.splitk > 0 ? a.splitk : 1);
    const int kt = (kt_env == 2 && !(a.ln_stats && nk_blk < 4)) ? 2 : 1;
    if (a.prof) {
        if (kt == 1) hipLaunchKernelGGL((gemm_ws72_kernel<OutT, true, 1>), dim3(tiles * a.splitk), dim3(640), 0, s, a);
        else hipLaunchKernelGGL((gemm_ws72_kernel<OutT, true, 2>), dim3(tiles * a.splitk), dim3(640), 0, s, a);
    } else {
        if (kt == 1) hipLaunchKernelGGL((gemm_ws72_kernel<OutT, false, 1>), dim3(tiles * a.splitk), dim3(640), 0, s, a);
        else hipLaunchKernelGGL((gemm_ws72_kernel<OutT, false, 2>), dim3(tiles * a.splitk), dim3(640), 0, s, a);
    }
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

// ------------------------------------------------------------------------------------------------
// Attention / GEMM co-residency probe (tools/attic/fuse_probe.py; VERDICT round 2, item 2: "hide the decode step's GEMM chain
// under its attention: horizontal fusion over two half-batches").  Streams do not overlap the decode kernels and CU masks
// starve the HBM stream (round 2); what is left to try is ONE launch whose blocks take either role: blocks of role 0 run the
// loader/consumer decode GEMM (gemm_ws_body, 8 waves, 64 KiB of LDS), blocks of role 1 the one-query decode attention
// (decode_attn_body with 8 waves = 8 (clip, head) pairs per block; scores in the first 8 x npad floats of the same dynamic
// LDS).  Both fit a CU together (2 x 64 KiB LDS, 16 waves of <= 128 VGPRs).  Roles alternate along the dispatch order of
// an XCD and flip parity every 32 blocks, so that under breadth-first placement every CU receives one block of each kind;
// hw_id (optional) records HW_REG_HW_ID per block so that the achieved placement can be counted, not assumed.
template <typename OutT>
__global__ __launch_bounds__(512) void fused_probe_kernel(const GemmArgs g, const DecodeAttnArgs d, int n_gemm, int n_attn, int npad,
                                                         unsigned* hw_id) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ __attribute__((aligned(16))) float ln_sm[128];
    __shared__ float red_m[8], red_l[8], red_acc[8 * 64];
    const int b = blockIdx.x;
    const int x = b & 7, q = b >> 3;                 // XCD and position in the XCD's dispatch sequence
    const int parity = (q ^ (q >> 5)) & 1;           // alternate roles; flip every 32 positions (one round of the XCD's CUs)
    // the two parity classes alternate in an XCD's sequence, so q / 2 positions of either class come before position q:
    // class-local index id in [0, grid / 2).  A class that has run out of work serves the other class's indices beyond
    // grid / 2 (which that class's own half of the grid cannot hold); what is left over returns at once.
    const int halfgrid = gridDim.x >> 1;
    int role = parity, id = (q >> 1) * 8 + x;
    {
        const int n_role = role ? n_attn : n_gemm, n_other = role ? n_gemm : n_attn;
        if (id >= n_role) {
            id = halfgrid + (id - n_role);
            role ^= 1;
            if (id >= n_other) return;
        }
    }
    if (hw_id && threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hw_id[b] = (v & 0xffffffu) | ((xcc & 7u) << 24) | ((unsigned)role << 28);
    }
    if (role == 0) gemm_ws_body<bf16, OutT, 4, false, 0>(g, id, n_gemm, dyn, ln_sm);
    else decode_attn_body<bf16, false, true, 1, 8>(d, id, (float*)dyn, npad, red_m, red_l, red_acc);
}

template <typename T, typename OutT, int STAGES, int ABLW = 0> static int launch_ws(const GemmArgs& a, hipStream_t s) {
    const int tiles = ceil_div(a.M, 64) * ceil_div(a.N, 64);
    if (a.prof && STAGES == 4)
        hipLaunchKernelGGL((gemm_ws_kernel<T, OutT, 4, true, ABLW>), dim3(tiles * a.splitk), dim3(512), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_ws_kernel<T, OutT, STAGES, false, ABLW>), dim3(tiles * a.splitk), dim3(512), 0, s, a);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int STAGES, int KPI>
static int launch_glds_kpi(const GemmArgs& a, hipStream_t s) {
    const int tiles = ceil_div(a.M, BM) * ceil_div(a.N, BN);
    dim3 grid(tiles * a.splitk), block(WM * WN * 64);
    hipLaunchKernelGGL((gemm_glds_kernel<T, OutT, BM, BN, WM, WN, STAGES, 0, false, KPI>), grid, block, 0, s, a);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int STAGES, int ABL = 0>
static int launch_glds(const GemmArgs& a, hipStream_t s) {
    const int tiles = ceil_div(a.M, BM) * ceil_div(a.N, BN);
    dim3 grid(tiles * a.splitk), block(WM * WN * 64);
    if (a.tile_map == 1) grid = dim3(8 * ((ceil_div(a.M, BM) + 3) / 4) * (ceil_div(a.N, BN) / 2));
    // one 32-row slice per wave (MI == 1) is where staging pays (128x128 on 8 waves: -12..-39 %); with two slices per
    // wave (the 4-wave 128x128 and the 256-wide tiles) it measured 30-40 % slower than direct stores
    if constexpr (BM * BN >= 128 * 128 && ABL == 0 && BM / (WM * 32) == 1) {
        if (a.stage_out) {
            hipLaunchKernelGGL((gemm_glds_kernel<T, OutT, BM, BN, WM, WN, STAGES, ABL, true>), grid, block, 0, s, a);
            DIMX_HIP(hipGetLastError());
            return DIMX_OK;
        }
    }
    hipLaunchKernelGGL((gemm_glds_kernel<T, OutT, BM, BN, WM, WN, STAGES, ABL>), grid, block, 0, s, a);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    const int tiles = ceil_div(a.M, BM) * ceil_div(a.N, BN);
    dim3 grid(tiles), block(WM * WN * 64);
    if (a.conv_T > 0)
        hipLaunchKernelGGL((gemm_kernel<T, OutT, BM, BN, WM, WN, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gemm_kernel<T, OutT, BM, BN, WM, WN, false>), grid, block, 0, s, a);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

// cfg ids (GemmArgs.cfg, 0 = automatic): tile / waves / ring depth of the LDS-DMA kernel
//  1: 128x128 2 stages   2: 128x128 3 stages   3: 64x64 4 stages   4: 64x64 3 stages   5: 64x64 2 stages
//  6: 128x64 3 stages    7: 128x64 2 stages    8: 64x128 3 stages
template <typename T, typename OutT> static int launch_by_cfg(const GemmArgs& a, int cfg, hipStream_t s) {
    DIMX_REQUIRE(!a.w_tiled || (cfg >= 34 && cfg <= 40), DIMX_ERR_ARG, "gemm: block-tiled W is read by the decode kernel only (cfg %d)", cfg);
    DIMX_REQUIRE(!a.ln_stats || (((cfg >= 34 && cfg <= 37) || cfg == 72) && a.splitk == 1 && a.ln_colsum && a.ln_C > 0 &&
                                 a.K >= 16 * Elem<T>::kPerChunk),
                 DIMX_ERR_ARG, "gemm: the deferred-LayerNorm epilogue exists in the decode kernel only, without split-K, from two "
                               "k-tiles up (cfg %d, K %d)", cfg, a.K);
    switch (cfg) {
        case 1: return launch_glds<T, OutT, 128, 128, 2, 2, 2>(a, s);
        case 3:
            if (a.prof) return launch_glds<T, OutT, 64, 64, 2, 2, 4, 3>(a, s);
            return launch_glds<T, OutT, 64, 64, 2, 2, 4>(a, s);
        case 4: return launch_glds<T, OutT, 64, 64, 2, 2, 3>(a, s);
        case 14: return launch_glds<T, OutT, 128, 128, 4, 2, 2>(a, s);  // 8 waves
#ifdef DIMX_GEMM_TUNING  // measured dead ends, kept for A/B runs: DIMX_TUNING=1 python __graft_entry__.py build
        case 18: return launch_glds<T, OutT, 256, 128, 4, 2, 2>(a, s);  // 8 waves, 96 KB ring
        case 19: return launch_glds<T, OutT, 256, 256, 4, 2, 2>(a, s);  // 8 waves, 128 KB ring
        case 31: return launch_glds<T, OutT, 256, 128, 4, 2, 3>(a, s);  // 8 waves, 144 KB ring, 2 tiles in flight
        case 29: return launch_glds<T, OutT, 128, 128, 4, 2, 4>(a, s);  // 8 waves, 128 KB ring, 3 tiles in flight
        case 30: return launch_glds<T, OutT, 128, 128, 4, 2, 3>(a, s);  // 8 waves, 96 KB ring
        case 32: return launch_glds_kpi<T, OutT, 64, 64, 2, 2, 2, 2>(a, s);  // 64x64, 2 k-tiles per iteration, 64 KB
        case 33: return launch_glds_kpi<T, OutT, 64, 64, 2, 2, 3, 2>(a, s);  // ... 3 groups deep, 96 KB
        case 36: return launch_ws<T, OutT, 5>(a, s);  // loader/consumer kernel, 80 KB ring (two blocks per CU just fit)
        case 37: return launch_ws<T, OutT, 8>(a, s);  // ... 128 KB ring
        case 38: return launch_ws<T, OutT, 4, 1>(a, s);  // ablations of 34 (wrong results by construction)
        case 39: return launch_ws<T, OutT, 4, 2>(a, s);
        case 40: return launch_ws<T, OutT, 4, 3>(a, s);
#endif
        case 72:  // 64x72, 6 consumer + 4 loader waves, one block per CU (bf16 operands)
            if constexpr (sizeof(T) == 2) return launch_ws72<OutT>(a, s);
            break;
        case 34: return launch_ws<T, OutT, 4>(a, s);  // 64x64, 4 consumer + 4 loader waves, 64 KB ring
        case 35: return launch_ws<T, OutT, 6>(a, s);  // ... 96 KB ring (one block per CU)
        default: break;
    }
    set_error("gemm: unknown cfg %d", cfg);
    return DIMX_ERR_ARG;
}

static inline void cfg_tile(int cfg, int& bm, int& bn) {
    switch (cfg) {
        case 1: case 2: case 11: case 14: case 15: case 22: case 23: case 25: case 26: case 29: case 30: bm = 128; bn = 128; break;
        case 6: case 7: case 12: case 13: case 24: bm = 128; bn = 64; break;
        case 16: bm = 256; bn = 64; break;
        case 17: case 18: case 27: case 31: bm = 256; bn = 128; break;
        case 19: case 28: bm = 256; bn = 256; break;
        case 8: bm = 64; bn = 128; break;
        case 72: bm = 64; bn = 72; break;
        default: bm = 64; bn = 64; break;
    }
}

template <typename T, typename OutT> static int launch_typed(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.splitk = 1;
    const int bk = 8 * Elem<T>::kPerChunk;
    const long tiles128 = (long)ceil_div(a.M, 128) * ceil_div(a.N, 128);
    const int kext = a.kloop ? a.kloop : a.ldw;
    const bool pipelined = a.conv_T == 0 && a.K % bk == 0 && a.K == kext && !a.force_simple;
    if (!pipelined) {
        // gather / ragged-K path: register-staged kernel
        if (tiles128 >= 256) return launch_cfg<T, OutT, 128, 128, 2, 2>(a, s);
        return launch_cfg<T, OutT, 64, 64, 2, 2>(a, s);
    }
    int cfg = a.cfg;
    // measured on MI355X: 128x128 8-wave for large M; bf16 from 288 tiles (the dW of a 1152 x 4608 projection at 4 800 rows:
    // 122.9 us on the 64 x 64 kernel, 80.4 on this one; 228 tiles and fewer: level or behind -- profiles/r03_train_gemm.txt)
    if (cfg == 0) cfg = tiles128 >= (sizeof(T) == 2 ? 288 : 512) ? 14 : gemm_cfg_small();
    if (a.cfg == 0 && sizeof(T) == 4 && a.out_slabs) cfg = gemm_cfg_small();  // f32 slabs: one tile shape whatever M is (see gemm_plan_splits)
    if (a.cfg == 0 && gemm_use_ws72(a0)) cfg = 72;
    DIMX_REQUIRE(cfg != 72 || (sizeof(T) == 2 && a.N % 72 == 0 && !a.w_tiled && a.nseg == 1 && a.seg[0].sd == 1 && a.seg[0].sh == 0 &&
                               a.seg[0].sb == a.seg[0].st * (long)a.rowT && a.rowadd_mode == 0),
                 DIMX_ERR_ARG, "gemm: cfg 72 needs bf16 operands, N %% 72 == 0 (N=%d) and a plain row-major destination", a.N);
    if (a.out_slabs) {
        a.splitk = gemm_plan_splits(a0);
        a.residual = nullptr;
    }
    static const int tile_map_env = getenv("DIMX_TILE_MAP") ? atoi(getenv("DIMX_TILE_MAP")) : 0;
    {
        int bm, bn;
        cfg_tile(cfg, bm, bn);
        const int tm = ceil_div(a.M, bm), tn = ceil_div(a.N, bn);
        a.tile_map = (tile_map_env == 1 && a.splitk == 1 && tm >= 32 && tn >= 2 && tn % 2 == 0) ? 1 : 0;
    }
    // large-M outputs leave through LDS as 16-byte row pieces (epilogue_staged) when the output map allows it
    a.stage_out = 0;
    a.vt_pack4 = 0;
    static const bool no_stage = getenv("DIMX_NO_STAGE") != nullptr;
    if (!no_stage && !a.out_slabs && a.M >= 2048 && (a.rowT == 1 || a.rowT >= 32)) {
        const int epo = 16 / (int)sizeof(OutT);
        (void)epo;
        bool ok = a.N % 4 == 0 && (a.nseg == 1 || a.seg_width % 32 == 0), any_row = false, pack = true;
        ok = ok && (!a.bias || ((uintptr_t)a.bias % 16) == 0);
        ok = ok && (!a.residual || (a.ldr % 4 == 0 && ((uintptr_t)a.residual % 16) == 0));
        ok = ok && (!a.rowadd_mode || (a.ld_rowadd % 4 == 0 && ((uintptr_t)a.rowadd % 16) == 0));
        for (int i = 0; i < a.nseg; ++i) {
            const OutSeg& g = a.seg[i];
            if (g.sd == 1) {
                any_row = true;
                ok = ok && ((uintptr_t)g.ptr % 16) == 0 && g.sb % 4 == 0 && g.st % 4 == 0 && g.sh % 4 == 0 && g.D % 4 == 0;
            } else {
                pack = pack && g.st == 1 && a.rowT % 4 == 0 && a.M % 4 == 0 && g.sb % 4 == 0 && g.sd % 4 == 0 &&
                       g.sh % 4 == 0 && ((uintptr_t)g.ptr % (4 * sizeof(OutT))) == 0;
            }
        }
        if (ok && any_row) {
            a.stage_out = 1;
            a.vt_pack4 = pack ? 1 : 0;
        }
    }
    return launch_by_cfg<T, OutT>(a, cfg, s);
}

int gemm_plan_splits(const GemmArgs& a) {
    if (!a.out_slabs) return 1;
    if (gemm_use_x3(a)) {
        int bn = 0, sp = 1;
        (void)gemm_x3_plan(a, &bn, &sp);
        return sp;
    }
    if (a.force_splitk > 0) return a.force_splitk;
    if (!a.allow_splitk) return 1;
    const int bk = a.in_dtype == DIMX_BF16 ? 64 : 32;
    const int kext = a.kloop ? a.kloop : a.ldw;
    if (a.conv_T != 0 || a.K % bk != 0 || a.K != kext || a.force_simple) return 1;  // register-staged kernel: no split
    int cfg = a.cfg, bm, bn;
    if (cfg == 0) cfg = (long)ceil_div(a.M, 128) * ceil_div(a.N, 128) >= 512 ? 14 : 4;
    // f32 parity mode: tile AND split count must not depend on M (a rank's shard reproduces the whole batch's rows bit for bit):
    // the slab path always plans for the 64 x 64 tile (launch_typed forces the same kernel), ADVICE round 4
    if (a.cfg == 0 && a.in_dtype != DIMX_BF16) cfg = 4;
    if (a.cfg == 0 && gemm_use_ws72(a)) cfg = 72;
    cfg_tile(cfg, bm, bn);
    const int tiles = ceil_div(a.M, bm) * ceil_div(a.N, bn);
    const int nk = kext / bk;
    if (a.in_dtype != DIMX_BF16) {
        // f32 (exact-f32 MFMA at the vector rate): a 64 x 64 tile's k-tile is 16 dependent 32x32x2 MFMAs per wave = 0.49 us of
        // matrix time (0.6 - 0.73 us measured with the hand-off) -- the block is MFMA-bound, two blocks on a CU take twice as
        // long, and the launch lasts ceil(blocks / CUs) rounds of ceil(nk / splits) k-tiles.  288 blocks (the bf16 sweet spot,
        // where co-resident blocks overlap their latencies for free) is the WORST choice here: 1.125 rounds cost two.  Pick the
        // split count that minimises rounds x k-tiles (+ a quarter k-tile per slab for its write and the consumer's read);
        // tools/bench_f32_decode_gemm.py: qkv 25.6 -> 19.5 us at 3 splits, K = 4608 45 -> 36.6 us at 8, 768-wide q 11.1 -> ~6.
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
                n_cu <= 0)
                n_cu = 256;
        }
        // The count must NOT depend on M: a rank's shard of a batch has to reproduce the rows of the whole batch bit for bit
        // (test_c4_*), and the split decides the summation order.  It is planned for the design point M = 256.
        const int tiles_ref = ceil_div(256, bm) * ceil_div(a.N, bn);
        int best = 1;
        float best_cost = 1e30f;
        for (int sp = 1; sp <= 8 && sp <= nk; ++sp) {
            const float cost = (float)(ceil_div(tiles_ref * sp, n_cu) * ceil_div(nk, sp)) + 0.25f * (float)sp;
            if (cost < best_cost) {
                best_cost = cost;
                best = sp;
            }
        }
        return best;
    }
    if (tiles >= 256) return 1;
    if (cfg == 72) {
        // one block per CU and no more (round 5, profiles/r05_gemm_blocks.txt): the largest split count that keeps the launch within
        // 256 blocks with at least min_nk k-tiles per block
        static const int min_nk72 = getenv("DIMX_SPLIT_MINNK") ? atoi(getenv("DIMX_SPLIT_MINNK")) : 6;
        int sp = 256 / tiles;
        const int max_sp = nk / min_nk72 > 0 ? nk / min_nk72 : 1;
        sp = sp > max_sp ? max_sp : sp;
        sp = sp > 8 ? 8 : sp;
        return sp < 1 ? 1 : sp;
    }
    // measured (tools/bench_gemm.py under rocprofv3): ~288 blocks (one per CU + a few) is the sweet spot
    static const int target = getenv("DIMX_SPLIT_TARGET") ? atoi(getenv("DIMX_SPLIT_TARGET")) : 288;
    static const int min_nk = getenv("DIMX_SPLIT_MINNK") ? atoi(getenv("DIMX_SPLIT_MINNK")) : 6;  // k-tiles per split at least (swept 3..9)
    int sp = (target + tiles / 2) / tiles;
    const int max_sp = nk / min_nk > 0 ? nk / min_nk : 1;
    sp = sp > max_sp ? max_sp : sp;
    sp = sp > 8 ? 8 : sp;
    return sp < 1 ? 1 : sp;
}

static int gemm_cfg_small() {
    // measured on MI355X: 64x64 with a 4-deep ring for decode, loader + consumer waves (34) over the 4-wave kernel (3):
    // qkv 10.3 -> 8.8, ff1 13.1 -> 12.1, ff2 11.9 -> 11.0 us, +1.0..1.5 % end to end on the same box; deeper rings
    // (36: 5, 35: 6, 37: 8 slots) are slower (tools/gemm_ab.py)
    static const int cfg_small = getenv("DIMX_GEMM_CFG_SMALL") ? atoi(getenv("DIMX_GEMM_CFG_SMALL")) : 34;
    return cfg_small;
}

// the 64 x 72 one-block-per-CU decode kernel: bf16 operands, the LDS-DMA path, a decode-sized M, N a multiple of 72 whose
// tiles fit the chip in one round (DIMX_NO_WS72=1: the 64 x 64 kernel everywhere, for A/B runs)
static bool gemm_use_ws72(const GemmArgs& a) {
    static const bool off = getenv("DIMX_NO_WS72") != nullptr;
    if (off || a.in_dtype != DIMX_BF16 || gemm_cfg_small() != 34) return false;
    const int kext = a.kloop ? a.kloop : a.ldw;
    if (a.conv_T != 0 || a.K % 64 != 0 || a.K != kext || a.force_simple || a.w_tiled) return false;
    if (a.M > 256 || a.N % 72 != 0) return false;
    // its epilogue stores to a plain row-major destination only (what gemm_set_plain_out builds; the decode step's GEMMs)
    if (a.nseg != 1 || a.seg[0].sd != 1 || a.seg[0].sh != 0 || a.seg[0].sb != a.seg[0].st * (long)a.rowT || a.rowadd_mode != 0) return false;
    return ceil_div(a.M, 64) * (a.N / 72) <= 256;
}

bool gemm_decode_has_ln_epilogue() {
    const int c = gemm_cfg_small();
    return c >= 34 && c <= 37;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    const int epc = a.in_dtype == DIMX_BF16 ? 8 : 4;
    const int bk = 8 * epc;
    const size_t es = dtype_size(a.in_dtype);
    DIMX_REQUIRE(a.A && a.W && a.M > 0 && a.N > 0 && a.K > 0, DIMX_ERR_ARG, "gemm: null operand or empty shape");
    DIMX_REQUIRE(a.ldw % epc == 0 && a.ldw >= a.K && (a.kloop ? a.kloop : a.ldw) % bk == 0 &&
                     (a.kloop == 0 || (a.kloop >= a.K && a.kloop <= a.ldw)),
                 DIMX_ERR_ARG, "gemm: W must be K-padded to %d (ldw=%d kloop=%d K=%d)", bk, a.ldw, a.kloop, a.K);
    DIMX_REQUIRE((a.lda * es) % 16 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0, DIMX_ERR_ARG,
                 "gemm: A/W rows must be 16-byte aligned (lda=%d)", a.lda);
    DIMX_REQUIRE(a.K % epc == 0, DIMX_ERR_ARG, "gemm: K=%d must be a multiple of %d", a.K, epc);
    DIMX_REQUIRE(a.rowT >= 1 && a.nseg >= 1 && a.nseg <= 3 && a.seg_width > 0, DIMX_ERR_ARG, "gemm: bad output map");
    if (a.conv_T > 0) {
        DIMX_REQUIRE(a.conv_C % bk == 0 && a.K == 5 * a.conv_C && a.M % a.conv_T == 0, DIMX_ERR_ARG,
                     "gemm(conv5): C=%d must be a multiple of %d, K=5C, M %% T == 0", a.conv_C, bk);
    }
    for (int i = 0; i < a.nseg; ++i)
        DIMX_REQUIRE(a.seg[i].ptr && a.seg[i].D > 0, DIMX_ERR_ARG, "gemm: output segment %d unset", i);
    if (a.out_slabs)
        DIMX_REQUIRE(a.out_dtype == DIMX_F32 && a.act == ACT_NONE && !a.residual && a.rowadd_mode == 0 && a.nseg == 1 &&
                         a.conv_T == 0 && a.K == (a.kloop ? a.kloop : a.ldw),
                     DIMX_ERR_ARG, "gemm: slab output needs a plain f32 GEMM without activation/residual");
    if (a.w_tiled)
        DIMX_REQUIRE(a.N % 8 == 0 && a.conv_T == 0 && !a.force_simple && a.K % bk == 0 && a.M < 4096 && a.kloop == 0, DIMX_ERR_ARG,
                     "gemm: block-tiled W needs N %% 8 == 0, K %% %d == 0 and a decode-sized M (N=%d K=%d M=%d)", bk, a.N, a.K, a.M);
    {   // DIMX_GEMM_LOG=1: one line per launch (tools/gemm_in_situ.py pairs them with a kernel trace)
        static const bool log = getenv("DIMX_GEMM_LOG") != nullptr;
        if (log)
            fprintf(stderr, "dimx-gemm M=%d N=%d K=%d in=%d out=%d bias=%d res=%d act=%d rowadd=%d nseg=%d slabs=%d big=%d\n", a.M, a.N, a.K,
                    a.in_dtype, a.out_dtype, a.bias ? 1 : 0, a.residual ? 1 : 0, a.act, a.rowadd_mode, a.nseg, a.out_slabs ? 1 : 0,
                    (a.in_dtype == DIMX_BF16 && a.cfg == 0 && gemm256_eligible(a)) ? 1 : 0);
    }
    if (a.in_dtype == DIMX_BF16) {
        if (a.cfg == 0 && gemm256_eligible(a)) return launch_gemm256(a, s);
        if (a.out_dtype == DIMX_BF16) return launch_typed<bf16, bf16>(a, s);
        return launch_typed<bf16, float>(a, s);
    }
    DIMX_REQUIRE(a.out_dtype == DIMX_F32, DIMX_ERR_ARG, "gemm: f32 inputs produce f32 outputs");
    if (gemm_use_x3(a)) return launch_gemm_x3(a, s);   // round 6: decode-sized f32 GEMMs on the bf16 matrix cores, f32-equivalent (gemm_x3.hip)
    return launch_typed<float, float>(a, s);
}


// which: 0 = both roles in one launch, 1 = GEMM blocks only, 2 = attention blocks only (same kernel, same geometry)
int launch_fused_probe(const GemmArgs& g0, const DecodeAttnArgs& d, int which, unsigned* hw_id, hipStream_t s) {
    GemmArgs g = g0;
    DIMX_REQUIRE(g.in_dtype == DIMX_BF16 && d.dtype == DIMX_BF16 && d.q_f32 && !d.knew, DIMX_ERR_ARG, "fused_probe: bf16, cross form");
    DIMX_REQUIRE(g.K % 64 == 0 && g.ldw == g.K && d.n_keys <= 2048, DIMX_ERR_ARG, "fused_probe: shape");
    g.splitk = 1;
    const int n_gemm = which == 2 ? 0 : ceil_div(g.M, 64) * ceil_div(g.N, 64);
    const int n_attn = which == 1 ? 0 : ceil_div(d.B * d.H, 8);
    const int npad = (d.n_keys + 15) / 16 * 16;
    DIMX_REQUIRE(8 * npad * 4 <= 4 * 16384, DIMX_ERR_ARG, "fused_probe: too many keys for the shared LDS window");
    const int total = n_gemm + n_attn;
    const int grid = (total + 15) / 16 * 16;  // whole (XCD round, parity pair) groups: surplus blocks return at once
    const size_t lds = 4 * 16384;
#define FP(OT)                                                                                                      \
    do {                                                                                                            \
        (void)hipFuncSetAttribute((const void*)fused_probe_kernel<OT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((fused_probe_kernel<OT>), dim3(grid), dim3(512), lds, s, g, d, n_gemm, n_attn, npad, hw_id);  \
    } while (0)
    if (g.out_dtype == DIMX_BF16) FP(bf16); else FP(float);
#undef FP
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace dimx
