// train_attn.hip -- attention of the training step on the matrix cores.  SURVEY 8 row f3; the mathematics is x-transformers'
// Attend (masked_fill(-max) before a float32 softmax) and its adjoint.  Two element types through one body:
//   bf16  (perf mode)    v_mfma_f32_32x32x16_bf16, operands rounded to bf16 on the way into LDS, f32 accumulation / statistics
//   float (parity mode)  v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation -- the numerics of the one-wave-per-row
//                        VALU kernels of train_kernels.hip (kept as the plain reference form, DIMX_TRAIN_ATTN_VALU=1) at a
//                        thirtieth of their time
//
//   forward   O = softmax(scale . Q K^T + masks) V, LSE_i = max_i + log sum_i kept for the backward pass
//   dQ        dS = P o (dO V^T - delta) . scale,  dQ = dS K          (P recomputed from the LSE, delta_i = dO_i . O_i)
//   dK, dV    dK = dS^T Q,  dV = P^T dO
//
// All three kernels give one wave a 32-row strip, 4 waves per block, the other operand
// streamed through LDS in 64-row tiles converted from the f32 activations on the way in (through registers: the loads of tile
// t + 1 are issued before tile t is multiplied).  The products are arranged so that
// no probability ever changes lanes:
//   * forward / dQ compute S^T = K Q^T (A = K rows from LDS, B = Q rows in registers): a lane owns ONE query (lane & 31) and
//     16 keys of every 32-key block, so the softmax is lane-local and P^T / dS^T are already B operands of the second product
//     (O^T = V^T P^T, dQ^T = K^T dS^T), whose A operand is read from a TRANSPOSED LDS tile with the keys in the order the
//     accumulator rows have them: k-step j covers keys {16j + 4(lane>>5) + r, 16j + 8 + 4(lane>>5) + r}, r = 0..3.
//   * dK / dV compute S = Q K^T (A = Q rows from LDS, B = K rows in registers): a lane owns ONE key and 16 queries per block,
//     P and dS are B operands of dV^T = dO^T P and dK^T = Q^T dS with dO^T / Q^T read from transposed LDS tiles.
// Outputs leave as 16-byte stores (an accumulator holds 4 consecutive head columns of one row).
// Deterministic: no atomics, every output element is written by exactly one lane.
#include <atomic>

#include "train.hpp"

namespace dimx {

namespace {

constexpr float kNegMaxF = -3.402823466e+38f;
constexpr float kInf = __builtin_inff();

// per element type: LDS row stride (elements; rows stay 16-byte aligned and spread over the banks), head columns one k-step of
// the first product covers (a lane holds half of them as one 16-byte fragment)
template <typename T> struct Cfg;
template <> struct Cfg<bf16> {
    typedef uint16_t E;
    static constexpr int kLd = 72, kDS = 16, kNS = 4;
};
template <> struct Cfg<float> {
    typedef float E;
    static constexpr int kLd = 68, kDS = 8, kNS = 8;
};
template <typename T> constexpr int tile_elems() { return 64 * Cfg<T>::kLd; }

// one k-step: bf16 = one 32x32x16 MFMA over the 8 + 8 elements of the two lane halves; f32 = four 32x32x2 MFMAs, the j-th over
// element j of both halves (any pairing of the k index works as long as A and B agree)
template <typename T> __device__ __forceinline__ void mma(f32x16_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma<bf16>(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x16_t& acc, const uint4& a, const uint4& b) {
    const float4 fa = __builtin_bit_cast(float4, a), fb = __builtin_bit_cast(float4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
}

__device__ __forceinline__ uint4 pack8(const float4& a, const float4& b) {
    return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}

// Staging of a 64-row x 64-column block of an f32 [L, ld] matrix (rows r0.., rows >= L read as zero) goes through registers in
// two halves, so that the global loads of tile t + 1 are in flight while tile t is being multiplied:
//   RowsReg: thread t owns 16 consecutive columns of row t / 4            -> row-major LDS tile
//   ColsReg: thread t owns column t % 64 of 16 rows (4 quads of 4)        -> TRANSPOSED tile dst[column][row]
struct RowsReg {
    float4 v[4];
};
struct ColsReg {
    float x[4][4];
};
__device__ __forceinline__ void load_rows(RowsReg& r, const float* __restrict__ src, int ld, int r0, int L, int tid) {
    const int row = tid >> 2, c = (tid & 3) * 16;
    if (r0 + row < L) {
        const float4* p = (const float4*)(src + (size_t)(r0 + row) * ld + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.v[i] = p[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <typename T> __device__ __forceinline__ void store_rows(typename Cfg<T>::E* dst, const RowsReg& r, int tid) {
    const int row = tid >> 2, c = (tid & 3) * 16;
    if constexpr (sizeof(T) == 2) {
        uint4* d = (uint4*)(dst + row * Cfg<T>::kLd + c);
        d[0] = pack8(r.v[0], r.v[1]);
        d[1] = pack8(r.v[2], r.v[3]);
    } else {
        float4* d = (float4*)(dst + row * Cfg<T>::kLd + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = r.v[i];
    }
}
__device__ __forceinline__ void load_cols(ColsReg& r, const float* __restrict__ src, int ld, int r0, int L, int tid) {
    const int d = tid & 63, quad = tid >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 16 * p + 4 * quad + i;
            r.x[p][i] = row < L ? src[(size_t)row * ld + d] : 0.f;
        }
    }
}
template <typename T> __device__ __forceinline__ void store_cols(typename Cfg<T>::E* dst, const ColsReg& r, int tid) {
    const int d = tid & 63, quad = tid >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if constexpr (sizeof(T) == 2)
            *(uint2*)(dst + d * Cfg<T>::kLd + 16 * p + 4 * quad) =
                make_uint2(pack_bf16x2(r.x[p][0], r.x[p][1]), pack_bf16x2(r.x[p][2], r.x[p][3]));
        else
            *(float4*)(dst + d * Cfg<T>::kLd + 16 * p + 4 * quad) = make_float4(r.x[p][0], r.x[p][1], r.x[p][2], r.x[p][3]);
    }
}

// B operand of the first product straight from an f32 row in global memory: k-step s = columns kDS s + (kDS / 2) (lane >> 5) ..
template <typename T>
__device__ __forceinline__ void frag_rows_global(uint4 (&f)[Cfg<T>::kNS], const float* __restrict__ row_ptr, bool valid, int hl) {
#pragma unroll
    for (int s = 0; s < Cfg<T>::kNS; ++s) {
        if (!valid) {
            f[s] = make_uint4(0u, 0u, 0u, 0u);
        } else if constexpr (sizeof(T) == 2) {
            const float4* p = (const float4*)(row_ptr + 16 * s + 8 * hl);
            f[s] = pack8(p[0], p[1]);
        } else {
            f[s] = __builtin_bit_cast(uint4, *(const float4*)(row_ptr + 8 * s + 4 * hl));
        }
    }
}

// A operand of the first product from a row-major tile: row 32 rb + (lane & 31), the same columns
template <typename T> __device__ __forceinline__ uint4 frag_a(const typename Cfg<T>::E* tile, int rb, int s, int l31, int hl) {
    return *(const uint4*)(tile + (32 * rb + l31) * Cfg<T>::kLd + Cfg<T>::kDS * s + (Cfg<T>::kDS / 2) * hl);
}

// Second product over the 64 rows of a tile: acc[db] += tileT[32 db + lane & 31][row] . x[row], x = the two 32-row accumulator
// blocks of the first product.  An accumulator register i of block blk is row 32 blk + 8 (i >> 2) + 4 (lane >> 5) + (i & 3):
//   bf16: k-step j takes registers 8 (j & 1) .. + 7 of block j >> 1 = rows {16 j + 4 hl + r, 16 j + 8 + 4 hl + r}, r = 0..3 -- the A
//         operand reads those two 4-row groups of the transposed tile
//   f32:  registers 4 g .. 4 g + 3 of block blk = rows 32 blk + 8 g + 4 hl + r are the B operands of four 32x32x2 MFMAs as they
//         stand; the A operand is the 16-byte read of the same four rows
template <typename T>
__device__ __forceinline__ void second_product(f32x16_t (&acc)[2], const typename Cfg<T>::E* tileT, const f32x16_t (&x)[2], int l31,
                                               int hl) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x16_t& a = x[j >> 1];
            const int o = 8 * (j & 1);
            const uint4 b = make_uint4(pack_bf16x2(a[o], a[o + 1]), pack_bf16x2(a[o + 2], a[o + 3]), pack_bf16x2(a[o + 4], a[o + 5]),
                                       pack_bf16x2(a[o + 6], a[o + 7]));
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const uint16_t* p = tileT + (32 * db + l31) * Cfg<T>::kLd + 16 * j + 4 * hl;
                const uint2 lo = *(const uint2*)p, hi = *(const uint2*)(p + 8);
                mma<T>(acc[db], make_uint4(lo.x, lo.y, hi.x, hi.y), b);
            }
        }
    } else {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16_t& a = x[blk];
                const uint4 b = __builtin_bit_cast(uint4, make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]));
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    mma<T>(acc[db], *(const uint4*)(tileT + (32 * db + l31) * Cfg<T>::kLd + 32 * blk + 8 * g + 4 * hl), b);
            }
        }
    }
}

__device__ __forceinline__ float other_half(float x) { return __shfl_xor(x, 32, 64); }

// keep / valid bit per key of a 64-key tile (wave-uniform 64-bit words)
__device__ __forceinline__ void key_bits(const TrAttn& a, int b, int k0, int lane, uint64_t& keep, uint64_t& valid) {
    const int key = k0 + lane;
    const bool v = key < a.Lk;
    bool kp = v;
    if (v && a.kmask) kp = a.kmask[(size_t)b * a.Lk + key] != 0;
    if (kp && a.kmask2) kp = a.kmask2[(size_t)b * a.Lk + key] != 0;
    keep = __ballot(kp);
    valid = __ballot(v);
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(TrAttn a, const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ o,
                                                            float* __restrict__ lse) {
    typedef typename Cfg<T>::E E;
    constexpr int NS = Cfg<T>::kNS;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    E* Ks = (E*)dyn_lds;
    E* Vt = Ks + tile_elems<T>();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hl = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0b = blockIdx.x * 128, q0w = q0b + 32 * w, qi = q0w + l31;
    const bool wave_on = q0w < a.Lq, q_ok = qi < a.Lq;
    uint4 Qf[NS];
    frag_rows_global<T>(Qf, q + ((size_t)b * a.Lq + (q_ok ? qi : 0)) * a.ldq + h * 64, q_ok, hl);
    const float* kb = k + (size_t)b * a.Lk * a.ldk + h * 64;
    const float* vb = v + (size_t)b * a.Lk * a.ldv + h * 64;
    f32x16_t accO[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) accO[0][i] = accO[1][i] = 0.f;
    float m = kNegMaxF, lsum = 0.f;
    int nkt = (a.Lk + 63) >> 6;
    if (a.causal) nkt = min(nkt, (min(q0b + 127, a.Lq - 1) >> 6) + 1);
    RowsReg kr;
    ColsReg vr;
    if (nkt > 0) {
        load_rows(kr, kb, a.ldk, 0, a.Lk, tid);
        load_cols(vr, vb, a.ldv, 0, a.Lk, tid);
    }
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        store_rows<T>(Ks, kr, tid);
        store_cols<T>(Vt, vr, tid);
        uint64_t keep, valid;
        key_bits(a, b, kt * 64, lane, keep, valid);
        __syncthreads();
        if (kt + 1 < nkt) {  // tile kt + 1 travels while tile kt is multiplied
            load_rows(kr, kb, a.ldk, (kt + 1) * 64, a.Lk, tid);
            load_cols(vr, vb, a.ldv, (kt + 1) * 64, a.Lk, tid);
        }
        if (!wave_on || (a.causal && kt * 64 > q0w + 31)) continue;
        f32x16_t st[2];
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kb2][i] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) mma<T>(st[kb2], frag_a<T>(Ks, kb2, s, l31, hl), Qf[s]);
        }
        float mt = kNegMaxF;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kl = 32 * kb2 + 8 * (i >> 2) + 4 * hl + (i & 3);
                const bool kp = ((keep >> kl) & 1) && !(a.causal && kt * 64 + kl > qi);
                const bool vd = (valid >> kl) & 1;
                const float s = vd ? (kp ? st[kb2][i] * a.scale : kNegMaxF) : -kInf;
                st[kb2][i] = s;
                mt = fmaxf(mt, s);
            }
        }
        mt = fmaxf(mt, other_half(mt));
        const float m_new = fmaxf(m, mt);
        const float alpha = __expf(m - m_new);
        m = m_new;
        float ps = 0.f;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __expf(st[kb2][i] - m_new);
                st[kb2][i] = p;
                ps += p;
            }
        }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            accO[0][i] *= alpha;
            accO[1][i] *= alpha;
        }
        second_product<T>(accO, Vt, st, l31, hl);
    }
    if (!q_ok) return;
    const float ltot = lsum + other_half(lsum);
    const float inv = 1.f / ltot;
    float* op = o + ((size_t)b * a.Lq + qi) * a.ldo + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(op + 32 * db + 8 * g + 4 * hl) = make_float4(accO[db][4 * g] * inv, accO[db][4 * g + 1] * inv,
                                                                    accO[db][4 * g + 2] * inv, accO[db][4 * g + 3] * inv);
    }
    if (hl == 0) lse[((size_t)b * a.H + h) * a.Lq + qi] = m + logf(ltot);
}

// ------------------------------------------------------------------------------------------------ dQ
// (the kernel also PRODUCES delta_i = dO_i . O_i for its query rows -- each lane half sums 32 columns of its row in f32, the halves
// meet through one shuffle -- and stores it for the dK / dV kernel that follows on the stream: one launch per attention less than the
// separate one-thread-per-row pass of round 3)
template <typename T>
__global__ __launch_bounds__(256) void attn_dq_mfma_kernel(TrAttn a, const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ o,
                                                           const float* __restrict__ d_o, const float* __restrict__ lse,
                                                           float* __restrict__ delta, float* __restrict__ dq, int lddq) {
    typedef typename Cfg<T>::E E;
    constexpr int NS = Cfg<T>::kNS;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    E* Ks = (E*)dyn_lds;
    E* Vs = Ks + tile_elems<T>();
    E* Kt = Vs + tile_elems<T>();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hl = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0b = blockIdx.x * 128, q0w = q0b + 32 * w, qi = q0w + l31;
    const bool wave_on = q0w < a.Lq, q_ok = qi < a.Lq;
    uint4 Qf[NS], Gf[NS];
    frag_rows_global<T>(Qf, q + ((size_t)b * a.Lq + (q_ok ? qi : 0)) * a.ldq + h * 64, q_ok, hl);
    frag_rows_global<T>(Gf, d_o + ((size_t)b * a.Lq + (q_ok ? qi : 0)) * a.ldo + h * 64, q_ok, hl);
    const float L = q_ok ? lse[((size_t)b * a.H + h) * a.Lq + qi] : kInf;
    float dl = 0.f;
    {
        const float4* op4 = (const float4*)(o + ((size_t)b * a.Lq + (q_ok ? qi : 0)) * a.ldo + h * 64 + 32 * hl);
        const float4* gp4 = (const float4*)(d_o + ((size_t)b * a.Lq + (q_ok ? qi : 0)) * a.ldo + h * 64 + 32 * hl);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 x = op4[c], g = gp4[c];
            dl += x.x * g.x + x.y * g.y + x.z * g.z + x.w * g.w;
        }
        dl += __shfl_xor(dl, 32);
        if (!q_ok) dl = 0.f;
        if (q_ok && hl == 0) delta[((size_t)b * a.H + h) * a.Lq + qi] = dl;
    }
    const float* kb = k + (size_t)b * a.Lk * a.ldk + h * 64;
    const float* vb = v + (size_t)b * a.Lk * a.ldv + h * 64;
    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
    int nkt = (a.Lk + 63) >> 6;
    if (a.causal) nkt = min(nkt, (min(q0b + 127, a.Lq - 1) >> 6) + 1);
    RowsReg kr, vr;
    ColsReg kc;
    if (nkt > 0) {
        load_rows(kr, kb, a.ldk, 0, a.Lk, tid);
        load_rows(vr, vb, a.ldv, 0, a.Lk, tid);
        load_cols(kc, kb, a.ldk, 0, a.Lk, tid);
    }
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        store_rows<T>(Ks, kr, tid);
        store_rows<T>(Vs, vr, tid);
        store_cols<T>(Kt, kc, tid);
        uint64_t keep, valid;
        key_bits(a, b, kt * 64, lane, keep, valid);
        __syncthreads();
        if (kt + 1 < nkt) {
            load_rows(kr, kb, a.ldk, (kt + 1) * 64, a.Lk, tid);
            load_rows(vr, vb, a.ldv, (kt + 1) * 64, a.Lk, tid);
            load_cols(kc, kb, a.ldk, (kt + 1) * 64, a.Lk, tid);
        }
        if (!wave_on || (a.causal && kt * 64 > q0w + 31)) continue;
        f32x16_t st[2], dp[2];
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kb2][i] = dp[kb2][i] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                mma<T>(st[kb2], frag_a<T>(Ks, kb2, s, l31, hl), Qf[s]);
                mma<T>(dp[kb2], frag_a<T>(Vs, kb2, s, l31, hl), Gf[s]);
            }
        }
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kl = 32 * kb2 + 8 * (i >> 2) + 4 * hl + (i & 3);
                const bool kp = ((keep >> kl) & 1) && !(a.causal && kt * 64 + kl > qi);
                const float p = kp ? __expf(st[kb2][i] * a.scale - L) : 0.f;  // a masked score has no gradient (masked_fill)
                st[kb2][i] = p * (dp[kb2][i] - dl) * a.scale;
            }
        }
        second_product<T>(acc, Kt, st, l31, hl);
    }
    if (!q_ok) return;
    float* op = dq + ((size_t)b * a.Lq + qi) * lddq + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(op + 32 * db + 8 * g + 4 * hl) =
                make_float4(acc[db][4 * g], acc[db][4 * g + 1], acc[db][4 * g + 2], acc[db][4 * g + 3]);
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <typename T>
__global__ __launch_bounds__(256) void attn_dkv_mfma_kernel(TrAttn a, const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ d_o,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv) {
    typedef typename Cfg<T>::E E;
    constexpr int NS = Cfg<T>::kNS;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    float* Ls = (float*)dyn_lds;
    float* Ds = Ls + 64;
    E* Qs = (E*)(dyn_lds + 512);
    E* Gs = Qs + tile_elems<T>();
    E* Qt = Gs + tile_elems<T>();
    E* Gt = Qt + tile_elems<T>();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hl = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int k0b = blockIdx.x * 128, k0w = k0b + 32 * w, kj = k0w + l31;
    const bool wave_on = k0w < a.Lk, k_ok = kj < a.Lk;
    bool kept = k_ok;
    if (kept && a.kmask) kept = a.kmask[(size_t)b * a.Lk + kj] != 0;
    if (kept && a.kmask2) kept = a.kmask2[(size_t)b * a.Lk + kj] != 0;
    uint4 Kf[NS], Vf[NS];
    frag_rows_global<T>(Kf, k + ((size_t)b * a.Lk + (k_ok ? kj : 0)) * a.ldk + h * 64, k_ok, hl);
    frag_rows_global<T>(Vf, v + ((size_t)b * a.Lk + (k_ok ? kj : 0)) * a.ldv + h * 64, k_ok, hl);
    const float* qb = q + (size_t)b * a.Lq * a.ldq + h * 64;
    const float* gb = d_o + (size_t)b * a.Lq * a.ldo + h * 64;
    const float* lp = lse + ((size_t)b * a.H + h) * a.Lq;
    const float* dp_ = delta + ((size_t)b * a.H + h) * a.Lq;
    f32x16_t accK[2], accV[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) accK[0][i] = accK[1][i] = accV[0][i] = accV[1][i] = 0.f;
    const int nqt = (a.Lq + 63) >> 6;
    const int qt0 = a.causal ? min(k0b >> 6, nqt) : 0;  // a causal query tile below the block's first key sees none of its keys
    RowsReg qr, gr;
    ColsReg qc, gc;
    float lreg = kInf, dreg = 0.f;
    auto load_tile = [&](int qt) {
        load_rows(qr, qb, a.ldq, qt * 64, a.Lq, tid);
        load_rows(gr, gb, a.ldo, qt * 64, a.Lq, tid);
        load_cols(qc, qb, a.ldq, qt * 64, a.Lq, tid);
        load_cols(gc, gb, a.ldo, qt * 64, a.Lq, tid);
        if (tid < 64) {
            const int qi = qt * 64 + tid;
            lreg = qi < a.Lq ? lp[qi] : kInf;
            dreg = qi < a.Lq ? dp_[qi] : 0.f;
        }
    };
    if (qt0 < nqt) load_tile(qt0);
    for (int qt = qt0; qt < nqt; ++qt) {
        __syncthreads();
        store_rows<T>(Qs, qr, tid);
        store_rows<T>(Gs, gr, tid);
        store_cols<T>(Qt, qc, tid);
        store_cols<T>(Gt, gc, tid);
        if (tid < 64) {
            Ls[tid] = lreg;
            Ds[tid] = dreg;
        }
        __syncthreads();
        if (qt + 1 < nqt) load_tile(qt + 1);
        if (!wave_on || (a.causal && qt * 64 + 63 < k0w)) continue;
        f32x16_t st[2], dp[2];
#pragma unroll
        for (int qb2 = 0; qb2 < 2; ++qb2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[qb2][i] = dp[qb2][i] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                mma<T>(st[qb2], frag_a<T>(Qs, qb2, s, l31, hl), Kf[s]);
                mma<T>(dp[qb2], frag_a<T>(Gs, qb2, s, l31, hl), Vf[s]);
            }
        }
#pragma unroll
        for (int qb2 = 0; qb2 < 2; ++qb2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = 32 * qb2 + 8 * g + 4 * hl;
                const float4 L4 = *(const float4*)(Ls + ql), D4 = *(const float4*)(Ds + ql);
                const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * g + r;
                    const bool kp = kept && !(a.causal && kj > qt * 64 + ql + r);
                    const float p = kp ? __expf(st[qb2][i] * a.scale - Lr[r]) : 0.f;
                    st[qb2][i] = p;
                    dp[qb2][i] = p * (dp[qb2][i] - Dr[r]) * a.scale;
                }
            }
        }
        second_product<T>(accV, Gt, st, l31, hl);
        second_product<T>(accK, Qt, dp, l31, hl);
    }
    if (!k_ok) return;
    float* kp_ = dk + ((size_t)b * a.Lk + kj) * lddk + h * 64;
    float* vp_ = dv + ((size_t)b * a.Lk + kj) * lddv + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(float4*)(kp_ + 32 * db + 8 * g + 4 * hl) =
                make_float4(accK[db][4 * g], accK[db][4 * g + 1], accK[db][4 * g + 2], accK[db][4 * g + 3]);
            *(float4*)(vp_ + 32 * db + 8 * g + 4 * hl) =
                make_float4(accV[db][4 * g], accV[db][4 * g + 1], accV[db][4 * g + 2], accV[db][4 * g + 3]);
        }
    }
}

// true once per (instantiation, device): one bit per device ordinal (a process-wide flag left the second GPU of a
// process without its dynamic-LDS limit)
template <typename T, int WHICH> static bool first_launch_on_this_device() {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    const unsigned long long bit = 1ull << dev;
    return (done.fetch_or(bit) & bit) == 0;
}

template <typename T> static size_t lds_bytes(int tiles, int extra) { return (size_t)tiles * tile_elems<T>() * sizeof(typename Cfg<T>::E) + extra; }

template <typename T> static int fwd_typed(const TrAttn& t, const float* q, const float* k, const float* v, float* o, float* lse, hipStream_t s) {
    const size_t lds = lds_bytes<T>(2, 0);
    if (first_launch_on_this_device<T, 0>())   // the attribute belongs to (function, device): a second GPU of the process needs its own call
        (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd_mfma_kernel<T>, dim3((t.Lq + 127) / 128, t.H, t.B), dim3(256), lds, s, t, q, k, v, o, lse);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

template <typename T>
static int bwd_typed(const TrAttn& t, const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                     float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, hipStream_t s) {
    const size_t lds_q = lds_bytes<T>(3, 0), lds_kv = lds_bytes<T>(4, 512);
    if (first_launch_on_this_device<T, 1>()) {
        (void)hipFuncSetAttribute((const void*)attn_dq_mfma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        (void)hipFuncSetAttribute((const void*)attn_dkv_mfma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    }
    hipLaunchKernelGGL(attn_dq_mfma_kernel<T>, dim3((t.Lq + 127) / 128, t.H, t.B), dim3(256), lds_q, s, t, q, k, v, o, d_o, lse, delta, dq,
                       lddq);
    hipLaunchKernelGGL(attn_dkv_mfma_kernel<T>, dim3((t.Lk + 127) / 128, t.H, t.B), dim3(256), lds_kv, s, t, q, k, v, d_o, lse, delta,
                       dk, lddk, dv, lddv);
    DIMX_HIP(hipGetLastError());
    return DIMX_OK;
}

}  // namespace

// t.mfma: 1 = bf16 operands (perf mode), 2 = exact-f32 MFMA (parity mode)
int tr_attn_fwd_mfma(const TrAttn& t, const float* q, const float* k, const float* v, float* o, float* lse, hipStream_t s) {
    return t.mfma == 2 ? fwd_typed<float>(t, q, k, v, o, lse, s) : fwd_typed<bf16>(t, q, k, v, o, lse, s);
}

int tr_attn_bwd_mfma(const TrAttn& t, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                     const float* lse, float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, hipStream_t s) {
    return t.mfma == 2 ? bwd_typed<float>(t, q, k, v, o, d_o, lse, delta, dq, lddq, dk, lddk, dv, lddv, s)
                       : bwd_typed<bf16>(t, q, k, v, o, d_o, lse, delta, dq, lddq, dk, lddk, dv, lddv, s);
}

}  // namespace dimx
