// train.hpp -- launchers of csrc/train_kernels.hip (the non-GEMM operators of the training step)
#pragma once
#include "common.hpp"

namespace dimx {

constexpr int kTrSlabs = 32;  // row slabs of the deterministic column reductions
constexpr int kLnBlocks = 600; // blocks (= partial rows) of the fused LayerNorm adjoint

struct TrAttn {
    int B, H, Lq, Lk;
    int ldq, ldk, ldv, ldo;  // row strides of the [B, L, *] f32 operands; head h at column h * 64 (dim_head 64)
    float scale;
    int causal;
    const uint8_t* kmask;    // [B, Lk] keep-mask (padding), optional
    const uint8_t* kmask2;   // [B, Lk] second keep-mask (the mask_prob draw), optional
    int mfma;                // train_attn.hip on the matrix cores: 1 = bf16 operands (perf mode), 2 = exact f32 (parity mode); 0 = the f32 VALU kernels
};

// operand copies of up to kPrepMax weight matrices in ONE launch: for each [N][K] f32 source the cast [N][Kp] (zero columns
// K..Kp) and the transposed cast [K][Np] (zero columns N..Np)
constexpr int kPrepMax = 64;
struct PrepDesc {
    const float* src;
    void* w;
    void* wt;
    int N, K, Kp, Np;
    int lds;      // row stride of src (elements)
    int tile0;    // first 32 x 32 tile of this matrix in the launch
    int tiles_k;  // tiles along K
    int gelu;     // 1: the copies hold erf-GELU(src) (the feed-forward activation as the next Linear's operand)
};
struct PrepTable {
    PrepDesc d[kPrepMax];
    int n;
    int total_tiles;
};
int tr_prep_weights(int out_dtype, const PrepTable& t, hipStream_t s);
// the same pair of copies for ONE matrix (an activation [rows][cols] with row stride lds): o [rows][Kp], t [cols][Mp]
int tr_prep_pair(int out_dtype, const float* src, int lds, int rows, int cols, void* o, int Kp, void* t, int Mp, hipStream_t s, int gelu = 0);
struct PrepFused {
    const float* src;
    const float* src2;
    const float* gamma;
    void* o;
    void* t;
    float* colpart;
    const float* beta;
    int lds, lds2, Kp, Mp, rows, cols, mode, chunk;
};
int tr_prep_fused(int out_dtype, int mode, const float* src, int lds, const float* src2, int lds2, const float* gamma, int rows, int cols,
                  void* o, int Kp, void* t, int Mp, float* colpart, int* n_part, hipStream_t s, const float* beta = nullptr);
// VQ-VAE decoder / legacy generator glue (train_kernels.hip)
int tr_im2col5(const float* x, float* x5, int B, int n, int C, hipStream_t s);
int tr_col2im5(const float* dx5, float* dx, int B, int n, int C, hipStream_t s);
int tr_lrelu_inorm_fwd(const float* x, float* y, int B, int n, int C, hipStream_t s);
int tr_lrelu_inorm_bwd(const float* x, const float* dy, float* dx, int B, int n, int C, hipStream_t s);
int tr_add_clip_rows(const float* a, const float* rows, float* y, int M, int n, int C, hipStream_t s);
int tr_argmax512(const float* logits, int32_t* idx, int B, int n_in, int n_out, int t_off, hipStream_t s);
int tr_cont_loss(const float* pred, const float* v_tgt, const uint8_t* mask, int B, int T, int n, float* rown, float* dpred, float* loss_out,
                 hipStream_t s);
int tr_pad_head_rows(const float* src, float* dst, int groups, int cols, int unpad, hipStream_t s);
int tr_pad_head_cols(const float* src, float* dst, int rows, int groups, int unpad, hipStream_t s);
int tr_emb_relu(const float* table, const int32_t* ids, float* e, int B, int E, hipStream_t s);
int tr_emb_relu_bwd(const float* table, const int32_t* ids, const float* de, float* dtable, int B, int E, hipStream_t s);
int tr_prepend_row(const float* first, const float* rest, float* joint, int B, int T, int C, int split, hipStream_t s);
int tr_prepend_tokens(const int32_t* z, const uint8_t* mask, int32_t* z_ext, uint8_t* m_ext, int B, int T, hipStream_t s);
int tr_gather128(const float* book, const int32_t* idx, float* out, int R, hipStream_t s);
// SLM pre-training glue (train_kernels.hip)
int tr_copy_bt(const float* src, long src_bs, int src_ld, float* dst, long dst_bs, int dst_ld, int B, int T, int C, const float* addrow,
               int accumulate, hipStream_t s);
int tr_copy_bt_u8(const uint8_t* src, long src_bs, uint8_t* dst, long dst_bs, int B, int T, int invert, hipStream_t s);
int tr_mask_tokens(const int32_t* z, const uint8_t* sel, int32_t* out, long n, hipStream_t s);
size_t tr_nce_scratch_floats(int B, int C);
int tr_nce(const float* xs, const float* xl, const uint8_t* mask, int B, int T, int C, float* scr, float* out, float* dxs, float* dxl,
           hipStream_t s);
int tr_transpose_pad(int out_dtype, const float* in, int ld_in, void* out, int ld_out, int R, int C, hipStream_t s);
int tr_attn_fwd(const TrAttn& t, const float* q, const float* k, const float* v, float* o, float* lse, hipStream_t s);
int tr_attn_bwd(const TrAttn& t, const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, hipStream_t s);
int tr_attn_fwd_mfma(const TrAttn& t, const float* q, const float* k, const float* v, float* o, float* lse, hipStream_t s);
int tr_attn_bwd_mfma(const TrAttn& t, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                     const float* lse, float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, hipStream_t s);
int tr_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, int accumulate, int M, int C, hipStream_t s);
int tr_xhat(const float* x, float* xh, int M, int C, hipStream_t s);
int tr_layernorm_bwd_fused(const float* x, const float* gamma, const float* dy, float* dx, int accumulate, int M, int C, float* part_g,
                           float* part_b, int* nrows, hipStream_t s);
int tr_colsum_partial(const float* dy, int M, int C, float* part, int* nrows, hipStream_t s);
// deferred column reductions: out[c] = sum_i part[i][c], i < nslab, for up to kFinMax entries in one launch
constexpr int kFinMax = 96;
struct FinDesc {
    const float* part;
    float* out;
    int C, nslab;
    int blk0;     // first block of this entry in the launch (one block per 64 columns)
};
struct FinTable {
    FinDesc d[kFinMax];
    int n;
    int total_blocks;
};
int tr_multi_finish(const FinTable& t, hipStream_t s);
int tr_gelu_fwd(const float* pre, float* out, long n, hipStream_t s);
int tr_gelu_bwd(const float* pre, const float* dh, float* dpre, long n, hipStream_t s);
int tr_add(float* y, const float* a, long n, hipStream_t s);
int tr_add_rows(const float* a, int lda, const float* row, const float* table, float scale, int T, float* y, int ldy, int M, int C,
                hipStream_t s);
int tr_zero_rows(float* y, const uint8_t* keep, int M, int C, hipStream_t s);
int tr_copy_cols(const float* src, int lds_, float* dst, int ldd, int M, int C, int accumulate, hipStream_t s);
int tr_pos_grad(const float* dx, float* dtable, int B, int T, int C, float scale, hipStream_t s, int accumulate = 0);
int tr_cross_entropy(const float* logits, const int32_t* target, float* row_loss, float* dlogits, int R, float* loss_out, hipStream_t s);
int tr_onehot_t(int out_dtype, const int32_t* tokens, void* out, int ld_out, int M, int rows, hipStream_t s);
int tr_grad_norm(const float* g, long n, float max_norm, float* part, float* norm_out, hipStream_t s);
int tr_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
             const float* clip, hipStream_t s);

}  // namespace dimx
