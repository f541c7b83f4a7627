// model.hpp -- handle layout of libdimx_hip: packed device weights + stage planners.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

namespace dimx {

struct Linear {
    void* w = nullptr;  // [N][Kp] in the handle's operand type (bf16 / f32)
    const float* bias = nullptr;
    int N = 0, K = 0, Kp = 0;
    // round 6, f32 parity mode, decoder projections: W as three bf16 planes [3][N][Kp] whose sum is W exactly -- the decode-sized
    // GEMMs then run on the bf16 matrix cores, f32-equivalent (gemm_x3.hip)
    void* w3 = nullptr;
};

struct VQBlock {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    Linear qkv, out, l1, l2;
    const void* mlp = nullptr;   // bf16 mode, hidden 384: the MLP sublayer's weights as mlp_fused.hip chunk images
};

struct VQNet {
    Linear vm, conv, le, post;      // encoder
    VQBlock enc[8];
    Linear pre, dconv, dle, rev;    // decoder
    VQBlock dec[8];
    const float* pe_enc = nullptr;  // [5000][hidden]
    const float* pe_dec = nullptr;
    const float* E = nullptr;       // [n_embed][zdim]
    const float* Et = nullptr;      // [zdim][n_embed]
    const float* ee = nullptr;      // [n_embed]
};

struct XAttn {
    const float* ln_g;
    Linear qkv;  // self: fused [3*inner][dim]; cross: q only [inner][dim]
    Linear kv;   // cross: fused [2*inner][dim]
    Linear out;
    // decoder cross attention, bf16 mode: q projection with the pre-norm's gamma folded in (W' = gamma o W) and the row
    // sums of W' -- the deferred-LayerNorm form of the decode step (ChainArgs.defer)
    Linear q_ln;
    const float* q_ln_colsum = nullptr;
};
struct XFF {
    const float* ln_g;
    Linear f1, f2;
    Linear f1_ln;  // decoder, bf16 mode: gamma-scaled first projection + row sums (GemmArgs.ln_stats)
    const float* f1_ln_colsum = nullptr;
    const void* mlp = nullptr;   // encoders, bf16 mode, dim 384: mlp_fused.hip chunk images
};
struct XEnc {
    Linear proj_in;
    const float* pos_emb = nullptr;  // [max_seq_len][dim]
    XAttn attn[8];
    XFF ff[8];
    const float* final_g = nullptr;
};
struct XDec {
    const float* tok_emb = nullptr;  // [num_tokens][dec_dim]
    const float* pos_emb = nullptr;  // [max_seq_len][dec_dim] (legacy decoder: use_abs_pos_emb=True)
    XAttn self_[8], cross[8];
    XFF ff[8];
    const float* final_g = nullptr;
    Linear logits;
    Linear cross_kv_all;  // [depth * 2 * inner][Kp]: every layer's [Wk; Wv] stacked (cross[l].kv are views into it)
};

// geometry of the three network families of a handle (variant 0 = SLMFT, 1 = legacy ListenerGenerator,
// 2 = SLM pre-training model: SLMFT's dims, bidirectional encoders + encoder_l, decoder with abs. pos. embedding)
struct VQGeom {
    int in_dim, in_pad, hidden, heads, inter, layers, out_dim, fqn, zdim, n_embed;
    bool has_decoder;
};
struct EncGeom {
    int dim_in, in_pad, dim, heads, dim_head, depth, ff_mult, causal;
};
struct DecGeom {
    int dim, heads, dim_head, depth, ff_mult, abs_pos, num_tokens, ctx_dim;
};

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

// simple bump planner over the caller's workspace; base == nullptr -> dry run (size only)
struct Arena {
    unsigned char* base;
    size_t off;
    size_t cap;
    bool overflow;
    Arena(void* b, size_t c) : base((unsigned char*)b), off(0), cap(c), overflow(false) {}
    void* take(size_t bytes) {
        off = align_up(off, 256);
        void* p = base ? base + off : nullptr;
        off += bytes;
        if (base && off > cap) overflow = true;
        return p;
    }
};

struct GraphKey {
    void* ws;
    int B, T, top_k;
    float temperature;
    const void* noise;
    uint64_t seed;
    const void *start, *mask, *tokens, *logits_out;
    int groups;
    bool operator==(const GraphKey& o) const {
        return groups == o.groups && ws == o.ws && B == o.B && T == o.T && top_k == o.top_k && temperature == o.temperature &&
               noise == o.noise && seed == o.seed && start == o.start && mask == o.mask && tokens == o.tokens &&
               logits_out == o.logits_out;
    }
};

}  // namespace dimx

namespace dimx {
void train_forget(dimx_ctx* h);   // train.hip: drop the training plan cached for a handle (dimx_destroy)
}

struct dimx_ctx {
    int device = 0;
    dimx_dims d;
    int variant = 0;
    dimx::VQGeom vqg[2];
    dimx::EncGeom encg[2];  // SLMFT: encoder_s, encoder_joint; legacy: generator.encoder only
    dimx::DecGeom decg;
    int mode = 0;
    int at = DIMX_F32;  // operand storage type of GEMM/attention inputs
    std::map<std::string, dimx::HostTensor> host;
    std::vector<std::string> required;
    int packed_mask = 0;  // COMP_* bits of the components whose packed device copies are current
    std::vector<void*> dev_allocs;
    dimx::VQNet vq[2];  // 0 speaker, 1 listener
    dimx::XEnc enc_s, enc_joint, enc_l;  // enc_l: SLM pre-training variant only
    dimx::XDec dec;
    const float *patch_s = nullptr, *patch_dec_s = nullptr, *norm_s_g = nullptr, *norm_s_b = nullptr;
    const float *patch_l = nullptr, *patch_dec_l = nullptr, *norm_l_g = nullptr, *norm_l_b = nullptr;  // SLM
    const float *norm_j_g = nullptr, *norm_j_b = nullptr;                                              // SLM `norm`
    // encode_ctx -> decode hand-off
    bool ctx_ready = false;
    int ctx_B = 0, ctx_T = 0, ctx_for_generate = 0;
    void* ctx_ws = nullptr;
    // generate step graph
    static constexpr int kMaxGroups = 8;
    hipGraphExec_t graph_exec[kMaxGroups] = {};
    hipGraphExec_t graph_multi[kMaxGroups] = {};  // the same step captured graph_unroll times back to back
    int graph_unroll = 16;                        // DIMX_GRAPH_UNROLL (1 = one launch per step)
    // prefill (VQ encode, encoders + context, VQ decode) as clip groups on several streams (round 5, bf16 mode; DIMX_PREFILL_GROUPS)
    static constexpr int kPreGroups = 4;
    int prefill_groups = 0;             // 0 = automatic, 1 = one batch on the caller's stream, 2..4
    hipStream_t pre_stream[kPreGroups - 1] = {};
    hipEvent_t pre_fork = nullptr, pre_join[kPreGroups - 1] = {};
    hipStream_t grp_stream[kMaxGroups] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxGroups] = {};
    int gen_groups = 1;  // independent clip groups decoded concurrently on separate streams (DIMX_GEN_GROUPS;
                         // measured on MI355X: 1 is fastest, concurrent step graphs do not overlap usefully)
    hipStream_t cap_stream = nullptr;  // capture happens here (the caller's stream may be the null stream)
    dimx::GraphKey graph_key{};
    bool graph_valid = false;
    int use_graph = 1;
    // XCD-local chain kernels of the decode step (chain.hip): bf16 mode, one sample per clip, 256-CU device
    int use_chain = 1;                  // DIMX_NO_CHAIN=1 keeps the one-kernel-per-op step
    int cu_count = 0;
    float* chain_stats_dev = nullptr;   // [8][32][32][2] partial row sums of the deferred-LayerNorm chain kernels
    int defer_ln = 1;                   // DIMX_NO_DEFER_LN=1 keeps the row-phase LayerNorm inside the chain kernels
    unsigned long long* layer_prof_dev = nullptr;  // DIMX_LAYER_PROF=1 (tuning): [8 layers][256 blocks][16] phase stamps of xcd_layer_kernel
    int multi_tr = 1;                   // round 6: several samples per clip (best-of-N), bf16: the decode cross attention on the MFMA prefill
                                        // attention kernel (attention_tr.hip) instead of the VALU multi-query kernel (DIMX_NO_MULTI_TR=1: the latter)
    int use_layer_chain = 1;            // DIMX_NO_LAYER_CHAIN=1 keeps the attention half of a layer as four launches (round 5)
    unsigned* chain_err_dev = nullptr;  // bit 0: two blocks claimed one (XCD, CU slot), bit 1: a group barrier timed out
    unsigned* chain_err_host = nullptr; // pinned copy, refreshed at the end of every generate call
    hipEvent_t chain_err_ev = nullptr;
    int chain_faults = 0;               // generate calls whose chain kernels reported a fault and that were re-run without them
    int chain_fault_inject = 0;         // test hook: that many of the next generate calls launch the chain kernels with fault = 1
    // sampler generator window of a sharded batch (dimx_set_shard): this handle generates clips
    // [shard_row_off, shard_row_off + B) of shard_rows_total (0 = the call's own B)
    int shard_row_off = 0, shard_rows_total = 0;
};
