// common.hpp -- shared host/device helpers for libdimx_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/dimx.h"

namespace dimx {

// ---------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);
const char* get_error();

#define DIMX_HIP(expr)                                                                       \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            dimx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return DIMX_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

#define DIMX_REQUIRE(cond, code, ...)       \
    do {                                    \
        if (!(cond)) {                      \
            dimx::set_error(__VA_ARGS__);   \
            return (code);                  \
        }                                   \
    } while (0)

#define DIMX_TRY(expr)            \
    do {                          \
        int _s = (expr);          \
        if (_s != DIMX_OK) return _s; \
    } while (0)

// ---------------------------------------------------------------- element types
struct bf16 {
    uint16_t x;
};

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
    static constexpr int kId = DIMX_F32;
};
template <> struct Elem<bf16> {
    static constexpr int kPerChunk = 8;
    static constexpr int kId = DIMX_BF16;
};

inline size_t dtype_size(int dt) { return dt == DIMX_BF16 ? 2 : 4; }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf16_to_f32(uint16_t u) {
    return __builtin_bit_cast(float, (uint32_t)u << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {  // round-to-nearest-even (v_cvt_pk_bf16_f32)
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// one v_cvt_pk_bf16_f32 (round-to-nearest-even, both halves); the scalar form above costs cvt + shift + or per pair
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
inline uint16_t host_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16>(const bf16* p) { return bf16_to_f32(p->x); }

template <typename T> __device__ __forceinline__ void store_from_f32(T* p, float v);
template <> __device__ __forceinline__ void store_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_from_f32<bf16>(bf16* p, float v) { p->x = f32_to_bf16(v); }

// ---------------------------------------------------------------- activations
enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_GELU_TANH = 2, ACT_GELU_ERF = 3 };

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_LEAKY: return x > 0.f ? x : 0.2f * x;
        case ACT_GELU_TANH: {
            // reference code/utils/base_model_util.py:81-94
            const float c = 0.7978845608028654f;
            float inner = c * (x + 0.044715f * (x * x * x));
            return x * (0.5f * (1.0f + tanhf(inner)));
        }
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
        default: return x;
    }
}

// Wave-wide reductions on the DPP / readlane paths instead of __shfl_xor (which hipcc lowers to ds_bpermute_b32: an LDS
// round trip of ~100 cycles per step, six dependent steps per reduction -- the tail of every LayerNorm, softmax and
// sampler launch of the decode step).  Bit-identical to the ASCENDING xor butterfly (xor 1, 2, 4, ...): quad_perm IS
// xor 1 / xor 2; after those steps a quad holds one value, so the half-row and row mirrors deliver exactly the partner
// group's partial sum (a + b == b + a), and the four row totals are combined as (t0 + t1) + (t2 + t3) like the butterfly
// does in every lane (verified bit for bit by tools/ubench/dpp_check.hip).  All 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ float lane_f32(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// v of lane (l ^ O) for O in {1, 2, 4, 8, 16, 32}, entirely on the vector ALU (what __shfl_xor means, without the
// ds_bpermute round trip): quad_perm for 1 / 2, two bank-masked row shifts for 4, row_ror:8 for 8, the gfx950 lane-swap
// instructions v_permlane16_swap / v_permlane32_swap for 16 / 32 (forms verified against __shfl_xor on the GPU:
// tools/ubench/dpp_check.hip).  All 64 lanes must be active.
__device__ __forceinline__ int wave_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int O> __device__ __forceinline__ int xor_lane_i32(int v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "xor distance");
    if constexpr (O == 1) {
        return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    } else if constexpr (O == 2) {
        return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    } else if constexpr (O == 4) {  // quads 0, 2 read lane + 4 (row_shl:4), quads 1, 3 read lane - 4 (row_shr:4)
        const int r = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);
    } else if constexpr (O == 8) {
        return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);  // row_ror:8
    } else if constexpr (O == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return ((wave_lane() >> 4) & 1) ? (int)r[0] : (int)r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return wave_lane() >= 32 ? (int)r[0] : (int)r[1];
    }
}
template <int O> __device__ __forceinline__ float xor_lane_f32(float v) {
    return __builtin_bit_cast(float, xor_lane_i32<O>(__builtin_bit_cast(int, v)));
}

// sum over aligned groups of 8 (16) lanes, every lane of the group gets it; == the xor-1,2,4(,8) butterfly bit for bit
// when the operation is commutative, which + and max are
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f32<DPP_XOR1>(v);
    v += dpp_f32<DPP_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v = group8_sum(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    return v;
}
// (value, index) of the wave's maximum, ties to the smaller index, in every lane: four in-row steps on DPP, the two
// cross-row steps on the lane-swap instructions (no LDS round trip left)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#define DIMX_ARGMAX_STEP(OV, OI)                          \
    do {                                                  \
        const float ov_ = (OV);                           \
        const int oi_ = (OI);                             \
        if (ov_ > v || (ov_ == v && oi_ < i)) {           \
            v = ov_;                                      \
            i = oi_;                                      \
        }                                                 \
    } while (0)
    DIMX_ARGMAX_STEP(dpp_f32<DPP_XOR1>(v), dpp_i32<DPP_XOR1>(i));
    DIMX_ARGMAX_STEP(dpp_f32<DPP_XOR2>(v), dpp_i32<DPP_XOR2>(i));
    DIMX_ARGMAX_STEP(dpp_f32<DPP_HALF_MIRROR>(v), dpp_i32<DPP_HALF_MIRROR>(i));
    DIMX_ARGMAX_STEP(dpp_f32<DPP_ROW_MIRROR>(v), dpp_i32<DPP_ROW_MIRROR>(i));
    DIMX_ARGMAX_STEP(xor_lane_f32<16>(v), xor_lane_i32<16>(i));
    DIMX_ARGMAX_STEP(xor_lane_f32<32>(v), xor_lane_i32<32>(i));
#undef DIMX_ARGMAX_STEP
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}
// The wave sum of the f32 parity mode and of the VQ search keeps its historical association (xor 32, 16, ... 1: the order
// oracle/vq_argmin.c restates); the bf16 perf mode takes the DPP form, which associates as xor 1, 2, ... 32 -- the two
// differ in the last bit for about a third of all inputs (tools/ubench/dpp_check.hip).
__device__ __forceinline__ float wave_sum(float v) {  // the xor 32, 16, ... 1 butterfly, step for step
    v += xor_lane_f32<32>(v);
    v += xor_lane_f32<16>(v);
    v += xor_lane_f32<8>(v);
    v += xor_lane_f32<4>(v);
    v += xor_lane_f32<2>(v);
    v += xor_lane_f32<1>(v);
    return v;
}
__device__ __forceinline__ float wave_sum_fast(float v) {
    v = row16_sum(v);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
template <bool FAST> __device__ __forceinline__ float wave_sum_sel(float v) { return FAST ? wave_sum_fast(v) : wave_sum(v); }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- GEMM launcher
struct OutSeg {
    void* ptr;
    long sb, sh, st, sd;  // element strides for (batch row, head, time, channel)
    int D;                // channels per head inside the segment
};

struct GemmArgs {
    int in_dtype, out_dtype;
    const void* A;
    int lda;
    const void* W;
    int ldw;  // row stride of W (elements), >= the k extent the loop runs over
    int kloop;  // k extent of the main loop (multiple of the k-tile; 0 = ldw).  ldw > kloop lets the caller pad the
                // row stride so that consecutive rows do not camp on the same L2 channels
    int M, N, K;
    // temporal k=5 convolution gather on A (conv_T > 0)
    int conv_T;
    const int32_t* conv_lens;
    int conv_C;
    // epilogue
    const float* bias;
    int act;
    const float* residual;
    int ldr;
    const float* rowadd;  // [rows, ld_rowadd] table added after the activation
    int rowadd_mode;      // 0 none, 1 row t, 2 row b + rowadd_off, 3 fixed row rowadd_off
    float rowadd_scale;
    int rowadd_off, ld_rowadd;
    int rowadd_div;       // mode 2: table row = b / rowadd_div + rowadd_off (several samples per clip share a row)
    // rows decompose as m = b*rowT + t (rowT = 1 -> b = m, t = 0); used by rowadd and the output map
    int rowT;
    // output: N columns split into nseg equal segments of seg_width columns; element (m, n) of
    // segment s goes to seg[s].ptr + b*sb + t*st + (nn / D)*sh + (nn % D)*sd, nn = n - s*seg_width
    OutSeg seg[3];
    int nseg, seg_width;
    // scheduling knobs
    int allow_splitk;  // perf mode: split K with f32 atomics when the grid is small (in-place residual only)
    int splitk;        // set by the launcher
    int force_simple;  // tests: force the register-staged kernel
    int cfg;           // tile/stage configuration id of the LDS-DMA kernel (0 = automatic)
    int force_splitk;  // tuning: requested split count (0 = automatic)
    int out_slabs;     // split-K partial sums go to f32 slabs C + split*slab_stride (plain stores, fixed order
                       // reduction by the consumer kernel); requires act none, no residual/rowadd, plain output
    long slab_stride;  // elements between slabs
    int stage_out;     // set by the launcher: large-M outputs leave through LDS as 8/16-byte row-contiguous pieces
    unsigned long long* prof;  // tuning only (tools/gemm_phases.py): per-block wall-clock stamps, ABL = 3 instantiation
    // deferred LayerNorm of the A operand (gemm_ws_kernel only; see ChainArgs.defer): A holds bf16(x) un-normalised, W the
    // gamma-scaled weights; acc <- rstd[m] * (acc - mean[m] * ln_colsum[n]) before bias / activation.  ln_stats =
    // [M / 32][32][32][2] per-slice {sum x, sum (x - slice mean)^2}, 32 equal column slices of ln_C columns.
    const float* ln_stats;
    const float* ln_colsum;
    int ln_C;
    unsigned* ln_err;  // optional: bit 2 is set when a row's |mean| exceeds 8 standard deviations (deferred form too coarse)
    int w_tiled;       // W is stored as [N/8][ldw/BK] blocks of 8 rows x 128 B (1 KiB, contiguous): an LDS-DMA piece then reads one
                       // contiguous KiB instead of 8 row segments (tools/ubench/cu_load_rate.hip: 122-143 vs 70-78 GB/s per CU).
                       // gemm_ws_kernel only.  Measured: -0.2 us per decode GEMM, nothing end to end (1467 vs 1467 clips/s) -- the
                       // model keeps row-major weights; the switch stays for tools/gemm_ab.py and the kernel test.
    const void* w3;    // round 6, f32 parity mode: W as three bf16 planes [3][N][ldw] whose sum is W exactly (split_x3); with it a
    long w3_plane;     // GEMM of the DECODE STEP (x3_decode) runs on the bf16 matrix cores, f32-equivalent (gemm_x3_kernel).  Elements between planes.
    int x3_decode;     // set by the decode step (dimx_generate) and by dimx_op_gemm_x3 only: the kernel choice must not depend on M -- a
                       // rank's shard of a batch reproduces the rows of the whole batch bit for bit whatever the batch size
    int tile_map;      // set by the launcher (prefill): 1 = every XCD works on one half of the N tiles of a quarter of the
                       // M tiles, so that its share of W (N/2 x K) stays L2-resident while the A panels stream through
    int vt_pack4;      // set by the launcher: transposed (time-contiguous) segments take 4 packed rows per store
};

void gemm_args_init(GemmArgs& a);
// plain row-major output helper
void gemm_set_plain_out(GemmArgs& a, void* C, int ldc);
int launch_gemm(const GemmArgs& a, hipStream_t s);
// true when small-M GEMMs run on the loader/consumer kernel, the only one with the deferred-LayerNorm epilogue (ln_stats)
bool gemm_decode_has_ln_epilogue();
// the 256 x 256 phase-pipelined prefill kernel (gemm256.hip): taken by launch_gemm when eligible
bool gemm256_eligible(const GemmArgs& a);
int launch_gemm256(const GemmArgs& a, hipStream_t s);
int launch_gemm256_segs(const GemmArgs& a, const OutSeg* segs, int nseg, int seg_width, hipStream_t s);
// the split-bf16 decode GEMM of the f32 parity mode (gemm_x3.hip): eligibility, tile / split plan (functions of N, K and the
// slab flag only), launch
bool gemm_use_x3(const GemmArgs& a);
bool gemm_x3_plan(const GemmArgs& a, int* bn_out, int* sp_out);
int launch_gemm_x3(const GemmArgs& a, hipStream_t s);
// W [N][ldw] f32 -> three bf16 planes [3][N][ldw] with plane0 + plane1 + plane2 == W exactly
int launch_split_x3(const float* w, void* planes, size_t n, hipStream_t s);
// number of K splits launch_gemm will use for an out_slabs GEMM (the consumer needs it)
int gemm_plan_splits(const GemmArgs& a);

// ---------------------------------------------------------------- other launchers
int launch_layernorm(int out_dtype, const float* x, void* y, const float* gamma, const float* beta, int M,
                     int C, hipStream_t s);
// x[row] += sum_s slabs[s][row] (fixed order), then y = LayerNorm(x); decode-step residual + pre-norm
int launch_add_slabs_layernorm(int out_dtype, float* x, const float* slabs, int nslab, long slab_stride, void* y,
                               const float* gamma, int M, int C, hipStream_t s);
int launch_instnorm(int out_dtype, const float* x, void* y, const int32_t* lens, int B, int T, int C,
                    hipStream_t s);

struct AttnArgs {
    int dtype;
    const void *q, *k, *vt;
    void* o;
    long q_sb, q_st, q_sh;  // q element (b,i,h,d) at q + b*q_sb + i*q_st + h*q_sh + d
    long k_sb, k_st, k_sh;
    long v_sb, v_sh, v_sd;  // vt element (b,h,d,j) at vt + b*v_sb + h*v_sh + d*v_sd + j
    int v_rows;             // 1: `vt` is V itself, row-major like k: element (b,j,h,d) at vt + b*v_sb + j*v_st + h*v_sh + d (bf16, D <= 64)
    long v_st;
    long o_sb, o_st, o_sh;
    int B, H, Lq, Lk, D;
    float scale;
    int causal;
    const int32_t* lens;   // [B] optional
    const uint8_t* kmask;  // [B, kmask_ld] optional
    int kmask_ld;
    int dbg;               // attention_tr.hip ablation bits (DIMX_ATTN_DBG, tuning only: results are wrong when set)
    // attention_tr.hip with a key mask: scratch for the packed 64-bit validity words, B * ceil(Lk / 64) of them, owned by the
    // CALLER (its workspace arena: one buffer per handle and call, so launches on other streams / devices never share it and
    // nothing is allocated inside a launch or a stream capture -- ADVICE round 4)
    unsigned long long* kwords;
    size_t kwords_cap;     // in words
    int kwords_ready;      // the words in `kwords` are already packed for this kmask (launch_pack_key_words): no pack launch -- the decode
                           // loop packs its context mask once per generation, not once per step and layer
};
// key validity words of attention_tr.hip: word[b][t] bit j <-> key 64 t + j of clip b is a key; words [B][ceil(Lk / 64)]
int launch_pack_key_words(const uint8_t* kmask, int kmask_ld, const int32_t* lens, int B, int Lk, unsigned long long* words, size_t cap,
                          hipStream_t s);
int launch_attention(const AttnArgs& a, hipStream_t s);
int launch_attention_tr(const AttnArgs& a, hipStream_t s);
// mlp_fused.hip: x <- x + W2 . gelu(W1 . LayerNorm(x) + b1) + b2 in one launch (bf16 operands, C = 384); the weights are packed once
// on the host into the kernel's LDS chunk images
size_t mlp_fused_packed_bytes(int C, int F);
int mlp_fused_pack(const float* w1, const float* b1, const float* w2, int C, int F, uint16_t* out);
int launch_mlp_fused(float* x, const void* packed, const float* b2, const float* ln_g, const float* ln_b, int M, int C, int F, int act,
                     hipStream_t s);  // attention_tr.hip: bf16, D 48 / 64, q / k / v row-major

struct DecodeAttnArgs {
    int dtype;             // storage type of q / caches / out
    const void* q;         // [B, q_ld] : head h at h*64
    int q_ld;
    int q_f32;             // q / knew / vnew are f32 split-K slabs [nslab][B, ld]: summed in slab order on load
    int nslab;
    long slab_stride;      // elements between slabs
    int force_nsplit;      // tests: waves per (clip, head) (0 = automatic)
    int clip_blocks;       // round 6 (two-engine experiment): one 12-wave block per clip that owns its CU (LDS request > half a CU),
                           // the layer kernel's attention shape as a launch of its own; bf16, H == 12, f32 slab q only
    int rows_per_clip;     // cross attention with several query rows per clip (multi-sample generation):
                           // rows r*S .. r*S+S-1 of q/out share clip r's K/V cache and mask (0/1 = one row)
    const void* knew;      // self-attn: this step's k/v rows [B, *] (nullptr for cross attention)
    const void* vnew;
    int kv_ld;
    void* kcache;          // [B,H,Tmax,64]
    void* vcache;
    int Tmax;
    void* out;             // [B, o_ld]
    int o_ld;
    int B, H;
    const int32_t* step;   // device scalar: self -> number of cached keys before this step
    int n_keys;            // cross: number of keys (T); self: ignored
    const uint8_t* kmask;  // cross: [B, kmask_ld] optional
    int kmask_ld;
    float scale;
};
int launch_decode_attn(const DecodeAttnArgs& a, hipStream_t s);
// attention / GEMM co-residency probe (gemm.hip fused_probe_kernel; tools/attic/fuse_probe.py)
int launch_fused_probe(const GemmArgs& g, const DecodeAttnArgs& d, int which, unsigned* hw_id, hipStream_t s);

int launch_vq_argmin(const float* z, int N, const float* Et /*[128][512]*/, const float* ee /*[512]*/,
                     int32_t* idx, float* best_d, float* margin, hipStream_t s);

// ---------------------------------------------------------------- XCD-local chain kernels (chain.hip)
struct ChainGemmDesc {
    const void* W;  // [N][ldw] bf16, rows = output columns
    int N, K, ldw;
    int cols, rows_pad, ncb, nkt;  // filled by the planner: columns per CU, 8-row padded, 32-column blocks, k-tiles
};
struct ChainArgs {
    int B;  // rows (clips), <= 256: group g = rows [32g, 32g + 32) lives on XCD g
    // first projection (optional, g1.W != nullptr): A1 [B, lda1] bf16 . W1^T -> xr [B, C] f32 (C = g1.N)
    ChainGemmDesc g1;
    const void* A1;
    int lda1;
    float* xr;
    // row phase: x [B, C] f32 += xr (+ slabs[0..nslab) of a preceding chip-wide split-K GEMM); y = LayerNorm(x) * gamma
    float* x;
    int C;
    const float* slabs;
    int nslab;
    long slab_stride;
    void* y;  // bf16 [B, C]
    const float* gamma;
    // second projection (optional): y . W2^T -> out2 [B, ld_out2] f32 (K = C)
    ChainGemmDesc g2;
    float* out2;
    int ld_out2;
    // synchronisation: this launch site's per-XCD arrival counters [8][16] (zeroed once per generate call), the
    // device step counter (epoch of the monotonic counters), error flags (bit 0 placement, bit 1 barrier timeout)
    unsigned* counters;
    unsigned* seen;  // [8][32] this launch site's (XCD, CU slot) claim stamps (zeroed with the counters)
    const int32_t* step;
    unsigned* err;
    unsigned long long* prof;  // tuning only: [256 blocks][16] wall-clock stamps (100 MHz) of the phase boundaries
    // Deferred LayerNorm (defer = 1, needs g1): the CU that owns a column slice of the first projection adds the residual
    // itself, writes x (f32) and y = bf16(x) UN-normalised plus the partial row sums {sum x, sum x^2} of its columns into
    // stats[group][CU][row][2]; the consumer of y multiplies with gamma-scaled weights W' = gamma o W and corrects its
    // result: LN(x) . W^T = rstd * (x . W'^T - mean * colsum(W')).  No row phase and no second group barrier; with no
    // second projection the kernel has no group barrier at all (the next launch -- ff1 -- is the consumer).
    int defer;
    float* stats;            // [8][32][32][2] f32
    const float* colsum2;    // [g2.N] f32: row sums of the gamma-scaled second projection (defer && g2)
    int nbar;                                                // set by the launcher
    int fault;  // test hook (dimx_debug_chain_fault): the blocks of XCD 2i+1 claim XCD 2i's slots -- a non-bijective placement
    int offA1, offW1, offRed1, offA2, offW2, offRed2;        // LDS plan, set by the launcher
};
// The attention half of a decoder layer as ONE launch (chain.hip, xcd_layer_kernel; round 5): self attention -> out-projection +
// residual -> (deferred LayerNorm) cross-q projection -> cross attention -> out-projection + residual.  Clip i of group g is CU
// slot i of XCD g for both attentions and all three projections split their columns over the XCD's 32 CUs, so every hand-off
// stays inside the XCD (four group barriers); the projections' weight slices are requested while the attentions stream.
struct LayerChainArgs {
    int B, C;                       // clips (<= 256), model dim
    DecodeAttnArgs sa, ca;          // self / cross attention, one query row per (clip, head); out = o (bf16 [B, ld_o]); ca.q = qc
    ChainGemmDesc g_so, g_cq, g_co; // self out-projection [C][inner], cross q (gamma-scaled) [inner][C], cross out-projection [C][inner]
    const void* o;                  // attention output rows = A operand of the two out-projections
    int ld_o;
    float* x;                       // [B, C] f32 residual stream
    void* y;                        // [B, C] bf16(x), un-normalised
    float* stats;                   // [8][32][32][2] partial row sums (deferred LayerNorm)
    const float* colsum_cq;         // [inner] row sums of the gamma-scaled q projection
    float* qc;                      // [B, ld_qc] f32 cross-attention queries
    int ld_qc;
    unsigned* counters;             // this launch site's per-XCD arrival counters [8][16] + claim stamps [8][32]
    unsigned* seen;
    const int32_t* step;            // epoch of the monotonic counters: launches on one counter set must see 0, 1, 2, ... (the decode
    int epoch_add;                  // step counter in dimx_generate; nullptr + epoch_add = the call index in dimx_op_layer_chain)
    unsigned* err;
    int fault;
    unsigned long long* prof;       // tuning only: [256][16] phase stamps
    int sc_stride;                  // floats per (clip, head) score row in LDS
    int perm;                       // round 6 experiment (DIMX_LAYER_PERM=1): wave w of CU slot i streams head w of clip (i + w) % 32 instead of
                                    // clip i -- a static spread of every clip's 12 heads over 12 CUs of its XCD (full groups only)
    int abl;                        // tuning only (DIMX_LAYER_ABL, dimx_op_layer_chain; results are then WRONG): bit 0 = the three weight
                                    // slices are not fetched (round 6: what do the 8 x replicated slices cost the launch?)
    int off_base, off_A1, off_W1, off_A2, off_W2, off_W3, off_A3;  // LDS plan (bytes), set by the launcher
};
bool layer_chain_supported(const LayerChainArgs& a, int cu_count);
int launch_layer_chain(const LayerChainArgs& a, hipStream_t s);
size_t chain_plan(ChainArgs& a);
bool chain_supported(const ChainArgs& a, int cu_count);
int launch_chain(const ChainArgs& a, hipStream_t s);

// elementwise helpers (elementwise.hip)
int launch_cast_pad(int out_dtype, const float* x, int ldx, const float* coladd, void* y, int ldy, int M, int K,
                    hipStream_t s, const uint8_t* zero_rows = nullptr);
int launch_gather_rows(int out_dtype, const float* table, int ld_table, int rows, const int32_t* idx, void* y,
                       int ldy, int M, int C, hipStream_t s);
int launch_context_concat(int out_dtype, const float* x_s, const float* patch, const float* audio, void* ctx,
                          int M, int dim, int dim_a, hipStream_t s);
int launch_finalize_idx(int32_t* idx, const int32_t* lens, int B, int T, int32_t pad_value, hipStream_t s,
                        int fqn = 1);
int launch_shift_tokens(const int32_t* z, int32_t* inp, int32_t* tgt, int B, int T, hipStream_t s);
int launch_ce_argmax(const float* logits, const int32_t* target, float* row_loss, int32_t* argmax_tok, int R,
                     int V, hipStream_t s);
int launch_sample(const float* logits, int ld_logits, int R, int top_k, float temperature, const float* noise,
                  uint64_t seed, const int32_t* step_dev, uint64_t step_host, int32_t* tokens, int tok_ld,
                  int tok_col_from_step, int nslab, long slab_stride, float* logits_out, int logits_out_ld, int row0,
                  int rows_total, const float* emb_table, int emb_C, float* x_next, int32_t* step_rw, unsigned* done_ctr,
                  hipStream_t s, const float* pos_table = nullptr, float pos_scale = 0.f, int pos_rows = 0,
                  const int32_t* dev_params = nullptr,
                  void* y_next = nullptr, const float* y_gamma = nullptr, int y_dtype = DIMX_F32);
// generate(): zero the per-group step / done counters and store temperature + seed next to them (read by the sampler)
int launch_gen_params(int32_t* base, int groups, float temperature, uint64_t seed, int row_off, int rows_total,
                      hipStream_t s);
int launch_embed_step(const float* table, int C, int rows, const int32_t* start, const int32_t* tokens, int tok_ld,
                      const int32_t* step_dev, float* x, int B, int start_div, hipStream_t s,
                      const float* pos_table = nullptr, float pos_scale = 0.f);
int launch_add_pos_rows(float* x, const float* pos, int M, int n, int C, float scale, hipStream_t s);
int launch_mask_lens(const uint8_t* mask, int32_t* lens, int B, int T, hipStream_t s);
int launch_legacy_scramble(int out_dtype, const float* E, const int32_t* idx, const int32_t* lens, void* out, int B,
                           int T, int fqn, int zdim, int n_embed, hipStream_t s);
int launch_step_inc(int32_t* step_dev, hipStream_t s);
int launch_copy_rows_step(const float* src, float* dst, int B, int V, int n, const int32_t* step_dev, hipStream_t s);
int launch_mask_and(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, hipStream_t s);

}  // namespace dimx
