/* dimx.h -- C-ABI of libdimx_hip.so, the MI355X (gfx950) implementation of the DIM-Listener
 * hot path.  Plain pointers and sizes only; no torch types.  Every function returns 0 on
 * success and a negative dimx_status on failure (never throws); dimx_last_error() gives the
 * thread-local message.  All device work is enqueued asynchronously on the hipStream_t passed
 * in (as void*; NULL = the null stream).  A handle is bound to one device and is not
 * re-entrant across streams.  Two exceptions to "asynchronously": dimx_generate reads its chain kernels' fault word before it
 * returns (bf16 mode), and dimx_vq_encode / dimx_encode_ctx / dimx_vq_decode run batches of >= 16 384 rows as
 * clip groups on the handle's own side streams (fork / join events around them: every result is complete in stream order when
 * the call's stream reaches the join) -- if the call's stream still has work in flight when such a call is made, the call waits
 * for it on the host before it forks (DIMX_PREFILL_GROUPS=1 keeps one batch on the caller's stream and never waits).
 *
 * Each stage entry point replaces one method of the reference's Python operator surface
 * (paths relative to /root/reference):
 *
 *   dimx_vq_encode   <- SLMFT.forward_vq / VQAutoEncoder.encode
 *                       (code/seq2seq_pretrain.py:480-494, code/models/stage1_BIWI.py:22-27,307-317)
 *   dimx_vq_argmin   <- VectorQuantizer.forward, argmin part (code/models/lib/quantizer.py:35-47)
 *   dimx_vq_decode   <- SLMFT.forward_vq_decoder / VQAutoEncoder.decode
 *                       (code/seq2seq_pretrain.py:454-464, code/models/stage1_BIWI.py:29-37,376-393)
 *   dimx_encode_ctx  <- SLMFT.forward_encoder + the context concat of forward_decoder
 *                       (code/seq2seq_pretrain.py:431-446)
 *   dimx_decode_tf   <- AutoregressiveWrapper.forward via forward_decoder(mode='train')
 *                       (code/seq2seq_pretrain.py:448)
 *   dimx_generate    <- AutoregressiveWrapper.generate via forward_decoder(mode='val')
 *                       (code/seq2seq_pretrain.py:450)
 *
 * The dimx_op_* functions expose the individual HIP kernels for unit parity tests.
 */
#ifndef DIMX_H
#define DIMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dimx_ctx* dimx_handle;

typedef enum {
    DIMX_OK = 0,
    DIMX_ERR_ARG = -1,      /* bad argument (null pointer, shape out of range, misaligned) */
    DIMX_ERR_HIP = -2,      /* a HIP runtime call failed */
    DIMX_ERR_WEIGHT = -3,   /* unknown / missing / wrongly shaped weight tensor */
    DIMX_ERR_STATE = -4,    /* call order violated (e.g. decode before encode_ctx) */
    DIMX_ERR_WORKSPACE = -5 /* workspace too small */
} dimx_status;

/* numeric modes */
#define DIMX_MODE_PARITY_F32 0 /* f32 storage, f32-input MFMA: matches the CPU oracle to ~1e-5 */
#define DIMX_MODE_PERF_BF16 1  /* bf16 GEMM/attention operands, f32 accumulate + f32 residual stream */

/* element types for dimx_op_* */
#define DIMX_F32 0
#define DIMX_BF16 1

typedef struct {
    /* VQ-VAE (reference code/config.yaml:15-30) */
    int vq_in_dim, vq_hidden, vq_layers, vq_heads, vq_inter, vq_n_embed, vq_zdim;
    /* seq2seq (reference code/seq2seq_pretrain.py:369-418) */
    int dim_in, dim, dim_a, enc_depth, dec_depth, heads, dim_head, num_tokens, max_seq_len, ff_mult;
    /* 0 = DIM-Listener SLMFT (code/seq2seq_pretrain.py:325); 1 = legacy ListenerGenerator (code/seq2seq.py:138,
     * driven by code/x_engine.py): speaker VQ 824->768 with 8 codes per frame, x-tf encoder/decoder dim 512,
     * depth 6, heads 8, decoder with absolute positional embedding */
    int variant;
    /* speaker VQ-VAE geometry of variant 1 (reference code/config_speaker_old.yaml:15-30) */
    int spk_in_dim, spk_hidden, spk_heads, spk_inter, spk_face_quan_num;
} dimx_dims;

typedef struct {
    const char* name;   /* state-dict key, e.g. "listener_vq.encoder.vertice_mapping.0.weight" */
    const float* data;  /* HOST pointer, float32, contiguous row-major; caller keeps ownership */
    int ndim;
    int64_t shape[4];
} dimx_weight_desc;

int dimx_version(void);
const char* dimx_last_error(void);
void dimx_default_dims(dimx_dims* d);
/* the legacy ListenerGenerator geometry (variant 1) */
void dimx_legacy_dims(dimx_dims* d);

int dimx_create(dimx_handle* h, int device_id, const dimx_dims* dims, int numeric_mode);
int dimx_destroy(dimx_handle h);
int dimx_numeric_mode(dimx_handle h);

/* Upload + pack weights (fused QKV, conv tap-major, bf16 copies, K padding).  May be called
 * several times; every key of the hot path must have been supplied before the first stage call
 * that needs it.  Unknown keys that belong to the reference surface but not to the path
 * (encoder_l.*, norm_l.*, norm.*, patch_embed_l, patch_embed_dec_l, *.project_out.weight / .bias) are
 * accepted and ignored.  OPTIONAL tensors (SURVEY A.2 [XT?]: details of x-transformers that differ between releases):
 * <encoder>.project_in.bias and <decoder>.to_logits.bias are applied when supplied and stay until dimx_begin_checkpoint (a call never
 * drops a tensor, so a chunked or key-sorted loader gives the same result in any order); a LayerNorm bias under *.attn_layers.* must
 * be all zero (DIMX_ERR_WEIGHT otherwise).  Any other unknown key is DIMX_ERR_WEIGHT.  Synchronous. */
int dimx_load_weights(dimx_handle h, const dimx_weight_desc* descs, int n);
/* A new checkpoint begins (what nn.Module.load_state_dict of another checkpoint means, code/finetune_s2s_pretrain.py:57): the
 * optional tensors of the previous one are forgotten; the required tensors stay until overwritten.  Synchronous. */
int dimx_begin_checkpoint(dimx_handle h);
/* number of hot-path tensors still missing (0 = ready) */
int dimx_missing_weights(dimx_handle h);

/* Bytes of caller-provided device workspace needed by any stage call at (B, T); 0 = shape not supported
 * (B < 1, T < 1 or T > max_seq_len). */
size_t dimx_workspace_bytes(dimx_handle h, int B, int T);
/* same, when dimx_generate will draw n_samples sequences per clip */
size_t dimx_workspace_bytes_samples(dimx_handle h, int B, int T, int n_samples);

/* Stream behaviour of the three prefill-sized stages (dimx_vq_encode, dimx_vq_decode(_latent), dimx_encode_ctx): from 16 384
 * rows (B x T) up they run as 2-4 clip groups, group 0 on `stream`, the others on side streams of the handle between a fork and a
 * join event (DIMX_PREFILL_GROUPS=1: never).  Consequences a pipelining caller must know: (1) when `stream` still has work in flight
 * the call WAITS FOR IT ON THE HOST before it forks (hipStreamSynchronize); (2) a `stream` that is being captured keeps one batch,
 * and the hipStreamQuery the fork uses must not run while ANOTHER thread captures in the global capture mode (use
 * hipStreamCaptureModeThreadLocal there, or DIMX_PREFILL_GROUPS=1); (3) on an error return the side streams have been joined:
 * nothing of the call still writes into `ws` once `stream` has drained.
 *
 * which: 0 = speaker VQ-VAE, 1 = listener VQ-VAE.
 * x: [B,T,56] f32, valid frames left-aligned; lens: [B] int32 device (NULL = all T).
 * pe_mode 0: every clip gets positional row 0 (the reference's batch-1 calls in forward_vq);
 * pe_mode 1: clip b gets row b + batch_row_offset (public batched VQAutoEncoder.encode).
 * idx: [B,T] int32, frames t >= lens[b] are set to pad_value.  z_out (optional) [B,T,128] f32. */
int dimx_vq_encode(dimx_handle h, int which, const float* x, const int32_t* lens, int B, int T,
                   int pe_mode, int batch_row_offset, int32_t pad_value, int32_t* idx, float* z_out,
                   void* ws, size_t ws_bytes, void* stream);

/* Nearest codebook entry, first index on ties.  z: [N,128] f32.  best_d / margin optional [N]. */
int dimx_vq_argmin(dimx_handle h, int which, const float* z, int N, int32_t* idx, float* best_d,
                   float* margin, void* stream);

/* idx: [B,L] int32 in [0,512).  Clip b is decoded with positional row b + batch_row_offset and
 * InstanceNorm / attention span the full L (reference behaviour on padded batches).
 * rows_per_clip S > 1 (multi-sample generation): rows b*S .. b*S+S-1 are samples of clip b and all use
 * positional row b + batch_row_offset, like S separate reference forwards would.  out: [B,L,56] f32. */
int dimx_vq_decode(dimx_handle h, int which, const int32_t* idx, int B, int L, int batch_row_offset,
                   int rows_per_clip, float* out, void* ws, size_t ws_bytes, void* stream);

/* VQAutoEncoder.decode on latents the caller supplies (reference code/models/stage1_BIWI.py:29-37: decode(quant)
 * applies the decoder to WHATEVER [B,128,L] tensor it is given, quantised or not): z [B,L,128] f32 (time-major,
 * i.e. quant.permute(0,2,1)), clip b decoded with positional row b + batch_row_offset.  out: [B,L,56] f32. */
int dimx_vq_decode_latent(dimx_handle h, int which, const float* z, int B, int L, int batch_row_offset,
                          float* out, void* ws, size_t ws_bytes, void* stream);

/* SLMFT.forward_encoder on its own (reference code/seq2seq_pretrain.py:431-442): x_s_out [B,T,384] f32 =
 * norm_s(encoder_joint(encoder_s(v_speaker + patch_embed_s))) with the causal attn_mask and the padding mask.
 * No audio, no context, no K/V projection; invalidates a context built earlier in the same workspace. */
int dimx_encode_speaker(dimx_handle h, const float* v_speaker, const uint8_t* mask, int B, int T,
                        float* x_s_out, void* ws, size_t ws_bytes, void* stream);

/* Multi-GPU sharding of one evaluation batch: this handle generates clips
 * [row_offset, row_offset + B) of a batch of rows_total clips.  Only the sampler's counter-based generator
 * depends on it (its counter is indexed by the GLOBAL sequence row, so the shards of a batch draw exactly what a
 * single process would draw for the same seed).  rows_total = 0 restores the default (the call's own B). */
int dimx_set_shard(dimx_handle h, int row_offset, int rows_total);

/* Speaker encoder stack + context assembly + cross-attention K/V projection for all decoder
 * layers.  v_speaker [B,T,56] f32, v_audio [B,T,768] f32, mask [B,T] uint8 (1 = valid frame).
 * The result lives in the workspace (same ws must be passed to the decode call that follows).
 * x_s_out (optional): [B,T,384] f32 = norm_s(encoder_joint(encoder_s(.))).
 * for_generate: 0 -> cross K/V laid out for dimx_decode_tf, 1 -> for dimx_generate. */
int dimx_encode_ctx(dimx_handle h, const float* v_speaker, const float* v_audio, const uint8_t* mask,
                    int B, int T, int for_generate, float* x_s_out, void* ws, size_t ws_bytes,
                    void* stream);
/* Variant 1 (legacy ListenerGenerator, reference code/seq2seq.py:220-249) gives the same entry point this
 * meaning: v_speaker [B,T,824] f32 with the valid frames of each clip FIRST (the reference indexes
 * v_speaker[i][mask[i]]; the host compacts), v_audio ignored (may be NULL), mask [B,T] uint8 with
 * popcount(mask[b]) = number of valid frames; the speaker VQ-VAE encoder (batch-1 per clip, positional row 0),
 * the quantiser, the reference's channel-major re-view and the 6-layer bidirectional encoder run inside;
 * x_s_out (optional): [B,T,512] f32 encoder output.  dimx_decode_tf then takes kv_mask = NULL and
 * dimx_generate produces T (not T-1) tokens per sequence: tokens [B*S, T], logits_out [B*S, T, 512]. */

/* Variant 2 = the SLM pre-training model (reference code/seq2seq_pretrain.py:58-323; dims of the default
 * geometry with dimx_dims.variant = 2): bidirectional encoders incl. encoder_l, decoder with absolute positional
 * embedding.  dimx_slm_encode replaces SLM.forward_encoder (:200-221): v_speaker / v_listener [B,T,56] f32,
 * mask [B,T] uint8 (1 = valid), mask_speaker / mask_listener [B,T] uint8 (1 = frame masked out: its input row is
 * zeroed after the patch embedding was added; NULL = none); outputs f32 x_s, x_l [B,T,384], x_joint [B,2T,384]
 * (speaker half first).  2T must not exceed max_seq_len. */
int dimx_slm_encode(dimx_handle h, const float* v_speaker, const float* v_listener, const uint8_t* mask,
                    const uint8_t* mask_speaker, const uint8_t* mask_listener, int B, int T, float* x_s,
                    float* x_l, float* x_joint, void* ws, size_t ws_bytes, void* stream);

/* Cross-attention context from a given encoder output (SLM.forward_decoder :223-229, SLMFT :445-446):
 * context = cat(x + patch_embed_dec_{s (which_patch 0) | l (1, variant 2 only)}, v_audio) and the K/V projection
 * of every decoder layer.  x: f32 [B, ldx_rows, 384] of which the first T rows of every clip are used
 * (ldx_rows = T for a plain [B,T,384]; 2T to address a half of x_joint).  Then dimx_decode_tf / dimx_generate
 * as after dimx_encode_ctx. */
int dimx_set_context(dimx_handle h, const float* x, int ldx_rows, int which_patch, const float* v_audio, int B,
                     int T, int for_generate, void* ws, size_t ws_bytes, void* stream);

/* Test hook for variant 1: the x_speaker tensor of code/seq2seq.py:224-241 ([B,T,1024] f32, optional) and
 * the speaker code indices ([B,T*8] int32, -100 beyond the clip length, optional). */
int dimx_legacy_speaker_features(dimx_handle h, const float* v_speaker, const uint8_t* mask, int B, int T,
                                 float* x_speaker_out, int32_t* idx_out, void* ws, size_t ws_bytes,
                                 void* stream);

/* Teacher-forced decoder pass.  z_l [B,T] int32 (-100 = ignore), ctx_mask [B,T] uint8,
 * kv_mask [B,T-1] uint8 keep-mask for self-attention keys (NULL = keep all).
 * logits [B,T-1,512] f32; row_loss (optional) [B,T-1] f32 = per-position cross entropy (0 where the
 * target is -100); argmax_tok (optional) [B,T-1] int32. */
int dimx_decode_tf(dimx_handle h, const int32_t* z_l, const uint8_t* ctx_mask, const uint8_t* kv_mask,
                   int B, int T, float* logits, float* row_loss, int32_t* argmax_tok, void* ws,
                   size_t ws_bytes, void* stream);

/* Autoregressive generation of T-1 tokens from start[B] with KV cache.
 * temperature <= 0 or exp_noise == NULL && seed == 0 -> greedy argmax.
 * exp_noise: [T-1,B,512] f32 Exp(1) samples (token = argmax softmax(top_k(logits)/temp)/noise);
 * if NULL and seed != 0 the noise is drawn on device from a counter-based generator.
 * n_samples S (1, 2, 4, 5, 8, 10): S independent samples per clip in one pass -- rows b*S+s of tokens /
 * exp_noise / logits_out belong to clip b; the clip's context K/V is streamed once for all S samples (the
 * reference's best-of-10 protocol, code/x_engine_pt.py:257, as one batched generation).  The workspace must
 * come from dimx_workspace_bytes_samples(h, B, T, S).
 * tokens: [B*S,T-1] int32.  logits_out (optional): [B*S,T-1,512] f32 raw logits of every step. */
int dimx_generate(dimx_handle h, const int32_t* start, const uint8_t* ctx_mask, int B, int T, int n_samples,
                  float temperature, int top_k, const float* exp_noise, uint64_t seed, int32_t* tokens,
                  float* logits_out, void* ws, size_t ws_bytes, void* stream);

/* In the bf16 mode dimx_generate runs part of the decode step as XCD-local chain kernels (B <= 256, one sample per clip,
 * 256-CU device) that rely on their 256 blocks being co-resident, one per CU.  They verify that and dimx_generate checks
 * their flags BEFORE it returns: with the chain path active the call therefore waits for its own generation to finish
 * (everything else stays asynchronous); on a fault it regenerates the same batch on the one-kernel-per-op step, keeps the
 * chain path off for this handle and counts the event here (0 = never happened).  The same check covers the deferred
 * LayerNorm's precision guard (a residual row whose |mean| exceeds 8 standard deviations: the batch is regenerated with the
 * row-phase LayerNorm, the deferred form stays off for the handle, the event is counted here too). */
int dimx_chain_faults(dimx_handle h);
/* Test hook: the next n_calls dimx_generate calls launch their chain kernels with a deliberately non-bijective
 * (XCD, CU slot) placement (the blocks of every odd XCD claim the slots of its even neighbour). */
int dimx_debug_chain_fault(dimx_handle h, int n_calls);

/* ---- training step: reference train_epoch, code/x_engine_pt.py:9-60 driven by
 * code/finetune_s2s_pretrain.py:105-143 (AdamW lr 1e-5, clip 1.0, VQ-VAEs frozen) --------------------------------------------
 * The reference differentiates SLMFT.forward(mode='train') with autograd; here forward AND backward of the teacher-forced stack
 * (encoder_s -> encoder_joint -> norm_s -> context -> decoder -> cross entropy, code/seq2seq_pretrain.py:431-450,496-514) are
 * hand-written HIP kernels (csrc/train.hip, train_kernels.hip; every Linear and both of its adjoints on the library's GEMM).
 * Parameters, gradients and the AdamW moments are FLAT f32 device arenas owned by the caller, laid out as reported by
 * dimx_train_param_info (the trainable tensors this path reaches; unused reference parameters -- encoder_l.*, norm_l.*,
 * project_out, ... -- get no gradient from autograd either and are not in the arena).  The handle must have been given the
 * full state dict (dimx_load_weights); its numeric mode selects the GEMM operands (f32-exact or bf16 with f32 accumulation). */
int dimx_train_num_params(dimx_handle h);
int64_t dimx_train_total(dimx_handle h);   /* floats in an arena (tensors are 16-byte aligned inside it) */
int dimx_train_param_info(dimx_handle h, int i, const char** name, int64_t* offset, int64_t* numel);
size_t dimx_train_workspace_bytes(dimx_handle h, int B, int T);
/* One forward + backward pass.  params / grads: arenas (grads is overwritten); v_speaker [B,T,56], v_audio [B,T,768] f32;
 * mask [B,T] uint8 (1 = valid frame); z_l [B,T] int32 listener codes, -100 on padding (from dimx_vq_encode); kv_mask [B,T-1]
 * uint8 keep-mask of AutoregressiveWrapper's mask_prob draw (NULL = keep all).  loss_out: 2 device floats {mean cross
 * entropy over the valid targets, 1 / number of valid targets}; logits_out optional [B,T-1,512] f32. */
int dimx_train_forward_backward(dimx_handle h, const float* params, float* grads, const float* v_speaker, const float* v_audio,
                                const uint8_t* mask, const int32_t* z_l, const uint8_t* kv_mask, int B, int T, float* loss_out,
                                float* logits_out, void* ws, size_t ws_bytes, void* stream);
/* How this handle's training steps were launched: out3 = {steps replayed from the captured hipGraph, steps launched kernel by
 * kernel, nodes of the captured graph}.  Below 4 096 rows (B x T) a call whose arguments (every pointer, B, T) equal the
 * previous call's is captured once and replayed afterwards; from 4 096 rows up the step is launched kernel by kernel with its
 * weight-gradient GEMMs on a side stream (the faster choice there: csrc/train.hip).  DIMX_TRAIN_GRAPH=0|1 / DIMX_TRAIN_SIDE=0|1
 * force either; results are bit-identical in every combination. */
int dimx_train_graph_stats(dimx_handle h, int64_t* out3);
/* The legacy generator's training step (reference loop code/x_engine.py:8-36 over ListenerGenerator.forward,
 * code/seq2seq.py:235-278): handle of variant 1.  The arenas follow dimx_train_param_info of that handle: generator.*, the
 * listener VQ-VAE's DECODER (listener_vq.decoder.*) and the listener-id conditioning (listener_embeddings.weight [100,256],
 * fc_listener.*) -- what the reference trains on this call (code/seq2seq.py:165-176; speaker_ids is None in the loop).
 * x_speaker [B,T,1024]: features of the frozen speaker VQ-VAE (dimx_legacy_speaker_features); z_l [B,T] listener codes, -100 on
 * padding; v_listener [B,T,56]; mask [B,T]; listener_ids [B] int32 or NULL; codebook [512,128] and pe [>=B rows of 384] are the
 * listener VQ-VAE's frozen codebook and its decoder's positional buffer (device pointers of the module's tensors).
 * loss_out: 4 device floats {cross entropy, 1 / valid targets, continuous loss, 1 / selected rows}; pred_out optional
 * [B,T-1,56]; logits_out optional [B,T (with ids) or T-1,512]. */
size_t dimx_train_legacy_workspace_bytes(dimx_handle h, int B, int T);
int dimx_train_legacy_forward_backward(dimx_handle h, const float* params, float* grads, const float* x_speaker, const int32_t* z_l,
                                       const float* v_listener, const uint8_t* mask, const int32_t* listener_ids, const float* codebook,
                                       const float* pe, int B, int T, float* loss_out, float* pred_out, float* logits_out, void* ws,
                                       size_t ws_bytes, void* stream);
/* The SLM pre-training step (reference loop code/train_s2s_pretrain.py:41-64 -> x_engine_pt.train_epoch over
 * SLM.forward, code/seq2seq_pretrain.py:300-323): handle of variant 2.  The arenas follow dimx_train_param_info of that handle:
 * everything the reference trains there (:98-113) -- the patch embeddings, norm / norm_s / norm_l, encoder_s / encoder_l /
 * encoder_joint, decoder_joint (with its absolute positional table) and BOTH VQ-VAE decoders; the VQ encoders and codebooks are
 * frozen.  v_speaker / v_listener [B,T,56], v_audio [B,T,768] f32; mask [B,T] uint8 (1 = valid frame); mask_speaker /
 * mask_listener [B,T] uint8 (1 = frame masked out: random_masking_unstructured, :170-183); z_s / z_l [B,T] int32 codes of the
 * frozen VQ encoders (dimx_vq_encode); codebook_* [512,128] and pe_* [>= B rows of 384]: the VQ-VAEs' frozen codebooks and their
 * decoders' positional buffers.  2T <= max_seq_len.  loss_out: 10 device floats {l_ce_s, 1 / targets, l_ce_l, 1 / targets,
 * l_cont_s, 1 / rows, l_cont_l, 1 / rows, nce, c_acc}; the reference's total is [0] + [2] + [4] + [6] + [8]. */
size_t dimx_train_slm_workspace_bytes(dimx_handle h, int B, int T);
int dimx_train_slm_forward_backward(dimx_handle h, const float* params, float* grads, const float* v_speaker, const float* v_listener,
                                    const float* v_audio, const uint8_t* mask, const uint8_t* mask_speaker, const uint8_t* mask_listener,
                                    const int32_t* z_s, const int32_t* z_l, const float* codebook_s, const float* codebook_l,
                                    const float* pe_s, const float* pe_l, int B, int T, float* loss_out, void* ws, size_t ws_bytes,
                                    void* stream);
/* Gradient clipping (torch.nn.utils.clip_grad_norm_, max_norm <= 0: none) + one torch.optim.AdamW step over a flat arena.
 * step: 1-based step count (bias correction).  scratch: >= 1026 device floats; scratch[1024] = gradient norm before clipping,
 * scratch[1025] = the clip coefficient applied. */
int dimx_train_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float max_norm, float* scratch, void* stream);
/* The attention operator of the training step alone (unit parity): q [B,Lq,H*64], k / v [B,Lk,H*64] f32, head h at columns
 * 64 h; kmask / kmask2 [B,Lk] uint8 keep-masks (NULL = keep all), causal: key j > query i masked; masked scores are filled
 * with -FLT_MAX before the softmax (x-transformers' Attend).  Writes o [B,Lq,H*64] and lse [B,H,Lq]; with d_o also delta
 * [B,H,Lq], dq, dk, dv.  mfma = 1: the bf16 matrix-core kernels the perf mode trains with (csrc/train_attn.hip), 2: the same
 * kernels on exact-f32 MFMA (what the parity mode trains with), 0: the one-wave-per-row f32 VALU kernels (the plain form). */
int dimx_op_train_attention(int mfma, const float* q, const float* k, const float* v, const float* d_o, const uint8_t* kmask,
                            const uint8_t* kmask2, int B, int H, int Lq, int Lk, int causal, float scale, float* o, float* lse,
                            float* delta, float* dq, float* dk, float* dv, void* stream);

/* ---- kernel-level entry points (unit parity tests) -------------------------------------- */

/* C = epilogue(A[M,K] . W[N,K]^T).  A/W element type `in_dtype`, C element type `out_dtype`.
 * W must be K-padded to a multiple of 64 (bf16) / 32 (f32) elements with zeros (ldw = padded K).
 * act: 0 none, 1 LeakyReLU(0.2), 2 GELU-tanh, 3 GELU-erf.  bias [N] f32 / residual [M,ldr] f32
 * optional.  conv_T > 0: A is [B*conv_T, C] and the GEMM is a k=5 replicate-padded temporal
 * convolution with K = 5*C, W tap-major [N][5][C]; conv_lens optional [B] int32.
 * flags: bit 0 = allow split-K with f32 atomics (only taken when residual == C, i.e. in-place accumulation
 * onto the residual stream, small M); bit 1 = force the register-staged (non LDS-DMA) kernel; bit 2 = C is [splits][M, ldc] f32
 * split-K slabs (count: dimx_op_gemm_slabs). */
int dimx_op_gemm(int in_dtype, int out_dtype, const void* A, int lda, const void* W, int ldw, void* C,
                 int ldc, int M, int N, int K, const float* bias, int act, const float* residual,
                 int ldr, int conv_T, const int32_t* conv_lens, int flags, void* stream);
/* The f32 parity mode's decode GEMM alone (csrc/gemm_x3.hip; reference arithmetic: the fp32 Linear layers of the decoder,
 * code/seq2seq_pretrain.py:413-418): an f32 number is the exact sum of three bf16 numbers, so C = A . W^T is computed on the bf16
 * matrix cores from the three planes of each operand with f32 accumulation -- an f32 GEMM at 6/16 of the f32-MFMA cost.
 * dimx_op_split_x3: w [n] f32 (device) -> planes [3][n] bf16 (device), plane0 + plane1 + plane2 == w exactly.
 * dimx_op_gemm_x3: A [M, lda] f32, planes of W [N, K] (K % 32 == 0, N a multiple of 36 / 64 / 72 / 96; planned for M = 256, any M runs), C [M, ldc] f32 or,
 * with flags bit 2, the split-K slabs [dimx_op_gemm_slabs(.., flags | 16)][M, ldc]; bias / act / residual as in dimx_op_gemm. */
int dimx_op_split_x3(const float* w, void* planes, long n, void* stream);
int dimx_op_gemm_x3(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K, const float* bias, int act,
                    const float* residual, int ldr, int flags, void* stream);
/* split-K slabs an out_slabs dimx_op_gemm call with these arguments writes (the f32 kernels plan the count from (N, K) themselves;
 * flags as in dimx_op_gemm: bit 0 allow split-K, bit 4 the split-bf16 kernel of the f32 parity mode, bits 16..23 a forced count) */
int dimx_op_gemm_slabs(int in_dtype, int M, int N, int K, int flags);
/* The cross-attention K/V projection as dimx_encode_ctx(for_generate=1) launches it: [M = B*rowT, K] . W[N, K]^T, the
 * N columns being nlayers x (K | V) segments of H*64 columns; segment i goes to the i-th [B, H, Tp, 64] cache inside
 * `out`.  bf16 with nlayers = 4 is the fused all-layers launch of the 256 x 256 kernel; f32 supports nlayers = 1. */
int dimx_op_gemm_headmajor(int dtype, const void* A, int lda, const void* W, int ldw, void* out, int M, int N, int K,
                           int rowT, int Tp, int nlayers, void* stream);
/* y = LayerNorm(x) over the last dim (C in {384,1152}), eps 1e-5; beta optional. */
int dimx_op_layernorm(int out_dtype, const float* x, void* y, const float* gamma, const float* beta,
                      int M, int C, void* stream);
/* y[b,t,c] = (x - mean_bc) / sqrt(var_bc + 1e-5) with statistics over t < len_b (biased var). */
int dimx_op_instnorm(int out_dtype, const float* x, void* y, const int32_t* lens, int B, int T, int C,
                     void* stream);
/* Flash-style attention on packed [B,L,H*D] q/k and a transposed v [B,H,D,Lk_pad] (as the QKV GEMM
 * writes it).  D in {48,64}.  out [B,Lq,H*D]. */
int dimx_op_attention(int dtype, const void* q, const void* k, const void* vt, void* out, int B, int H,
                      int Lq, int Lk, int D, int ldq, int ldk, int ld_vt, int ldo, float scale,
                      int causal, const int32_t* lens, const uint8_t* kmask, void* stream);
/* The same attention with v row-major like k ([B,Lk,H*D] bf16; what the perf mode's fused q/k/v projection writes since round 3:
 * three row-contiguous destinations for the 256x256 GEMM, the transposition happens on the way into LDS).  bf16, D in {48,64}. */
int dimx_op_attention_rowv(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq, int Lk, int D, int ldq,
                           int ldk, int ldv, int ldo, float scale, int causal, const int32_t* lens, const uint8_t* kmask,
                           void* stream);
/* One-query (autoregressive step) attention over a [B,H,Tmax,64] K/V cache, n_keys keys per (clip,head);
 * q/out are [B,H*64].  kmask optional [B,n_keys].  Cross-attention form (no cache append).
 * nsplit: waves per (clip, head) sharing the keys (1, 2, 4; 0 = automatic).  q_is_f32: q is a single f32 slab
 * (the form dimx_generate launches) instead of the cache's element type. */
int dimx_op_decode_attn(int dtype, const void* q, const void* kcache, const void* vcache, void* out, int B, int H,
                        int Tmax, int n_keys, float scale, const uint8_t* kmask, int nsplit, int q_is_f32,
                        void* stream);
/* Tuning probe (tools/attic/fuse_probe.py): ONE launch whose blocks run either the decode GEMM C[M,N] = act(A[M,K] . W[N,K]^T + bias)
 * (bf16 operands, the loader/consumer kernel of the decode step) or the cross-attention form of the decode attention over
 * [B,H,Tmax,64] bf16 caches (q f32 [B,H*64]), so that both kinds are co-resident on every CU.  which: 0 both, 1 GEMM blocks only,
 * 2 attention blocks only.  hw_id (optional, [grid] uint32): per block HW_REG_HW_ID[23:0] | XCC id << 24 | role << 28. */
int dimx_op_fused_probe(const void* A, const void* W, const float* bias, void* C, int out_dtype, int M, int N, int K, int act,
                        const void* q, const void* kcache, const void* vcache, void* out, int B, int H, int Tmax, int n_keys,
                        float scale, const uint8_t* kmask, int which, uint32_t* hw_id, void* stream);
/* Self-attention form of the step kernel, exactly as dimx_generate launches it: qkv [B, ld] holds this step's
 * q | k | v (H*64 each; f32 when q_is_f32, else the cache type), *step_dev keys are already cached; the kernel
 * appends k/v at position *step_dev and attends over *step_dev + 1 keys.  out [B, H*64] in the cache type. */
int dimx_op_decode_attn_self(int dtype, const void* qkv, int ld, void* kcache, void* vcache, void* out, int B, int H,
                             int Tmax, const int32_t* step_dev, float scale, int q_is_f32, void* stream);
/* Decode-step residual + pre-norm: x[M,C] += sum_s slabs[s] (fixed order), y = LayerNorm(x) * gamma (no bias). */
/* The prefill's fused feed-forward sublayer alone (csrc/mlp_fused.hip; unit parity): x [M,C] f32 on the device is replaced by
 * x + W2 . gelu(W1 . LayerNorm(x) + b1) + b2 with bf16 operands and f32 accumulation.  w1 [F,C], b1 [F] (or NULL), w2 [C,F] are HOST
 * f32 arrays (packed into the kernel's chunk images by the call); b2 [C], ln_g [C], ln_b [C] (or NULL) are device f32.  C = 384,
 * F a multiple of 32; act 2 = tanh-GELU (VQ-VAE blocks, code/models/lib/base_models.py:56-68), 3 = erf-GELU (x-transformers
 * FeedForward).  Synchronises the stream. */
int dimx_op_mlp_fused(float* x, const float* w1_host, const float* b1_host, const float* w2_host, const float* b2, const float* ln_g,
                      const float* ln_b, int M, int C, int F, int act, void* stream);
/* The same in its parts (what dimx_load_weights / the forward do): the size of the packed weights, the host-side packing, and the
 * asynchronous launch on weights that already sit on the device (packed: dimx_mlp_fused_packed_bytes(C, F) bytes, 16-byte aligned). */
size_t dimx_mlp_fused_packed_bytes(int C, int F);
int dimx_mlp_fused_pack(const float* w1_host, const float* b1_host, const float* w2_host, int C, int F, void* out_host, size_t out_bytes);
int dimx_op_mlp_fused_packed(float* x, const void* packed, const float* b2, const float* ln_g, const float* ln_b, int M, int C, int F,
                             int act, void* stream);
int dimx_op_add_slabs_layernorm(int out_dtype, float* x, const float* slabs, int nslab, long slab_stride, void* y,
                                const float* gamma, int M, int C, void* stream);
/* One XCD-local chain launch of the decode step (csrc/chain.hip; bf16 only, B <= 256, 256-CU device):
 *   [W1 != NULL]  xr = A1[B,K1] . W1[C,K1]^T ;   x[B,C] += xr + sum_s slabs[s][B,C] ;   y = bf16(LayerNorm(x) * gamma) ;
 *   [W2 != NULL]  out2[B,N2] = y . W2[N2,C]^T   (f32).
 * A1 / W1 / W2 / y are bf16, x / slabs / gamma / out2 f32.  scratch: >= 2048 + B*C*4 bytes of device memory; on return
 * (after the stream has drained) ((uint32_t*)scratch)[129] holds the kernel's error flags (0 = ok, bit 0 = two blocks
 * claimed the same (XCD, CU slot), bit 1 = a group barrier timed out, bit 2 (deferred form only) = a row's |mean| exceeds 8
 * standard deviations, i.e. the deferred LayerNorm's bf16(x) operand is too coarse for it). */
int dimx_op_chain(const void* A1, int K1, const void* W1, float* x, const float* slabs, int nslab, const float* gamma,
                  void* y, const void* W2, int N2, float* out2, int B, int C, void* scratch, void* stream);
/* Deferred-LayerNorm form of the chain launch (what generate() runs in the bf16 mode):
 *   x[B,C] += A1[B,K1] . W1[C,K1]^T ;  y = bf16(x) (NOT normalised) ;  stats[8][32][32][2] = {sum x, sum (x - slice mean)^2} of
 *   the CU's column slice per (row group, CU, row), combined by the consumers with the parallel-variance formula ;  [W2s != NULL]  out2[B,N2] = rstd * (y . W2s^T - mean * colsum2), i.e. LayerNorm(x) * gamma . W2^T
 *   for W2s = gamma o W2 (columns scaled) and colsum2[n] = sum_k W2s[n][k].  scratch / error flags as dimx_op_chain. */
int dimx_op_chain_ln(const void* A1, int K1, const void* W1, float* x, void* y, float* stats, const void* W2s,
                     const float* colsum2, int N2, float* out2, int B, int C, void* scratch, void* stream);
/* C[M,N] = act(LayerNorm-corrected A . Ws^T + bias): the consumer side of dimx_op_chain_ln (decode-step ff1).  A = bf16(x)
 * un-normalised [M,K], Ws = gamma o W bf16 [N,K], stats as written by dimx_op_chain_ln over K columns, colsum[n] = sum_k
 * Ws[n][k]; out_dtype DIMX_BF16 / DIMX_F32; M <= 256. */
int dimx_op_gemm_ln(int out_dtype, const void* A, const void* Ws, void* C, int M, int N, int K, const float* bias, int act,
                    const float* stats, const float* colsum, void* stream);
/* The attention half of one decoder layer of the decode step as ONE XCD-local launch (csrc/chain.hip xcd_layer_kernel; what
 * dimx_generate runs per layer for 128 < B <= 256 clips in the bf16 mode; x-transformers Decoder layer, reference
 * code/seq2seq_pretrain.py:413-419, one step of AutoregressiveWrapper.generate :450):
 *   o = self-attention(q, k, v of this step summed from the nslab f32 slabs qkv[s][B, 3*768]; keys = *step cached rows of
 *       sk / sv [B,12,T,64] bf16, the new row is appended);   x += o . Wso^T;   y = bf16(x);   qc = LN(x) . Wq^T in its deferred
 *   form (w_cq = gamma o Wq, colsum_cq its row sums);   o = cross-attention(qc, ck / cv [B,12,Tp,64] bf16, n_keys, kmask [B,n_keys]);
 *   x += o . Wco^T;   y = bf16(x);   stats = the partial row sums of x for the consumer's deferred LayerNorm.
 * w_so / w_co [1152][768], w_cq [768][1152] bf16 row-major; x [B,1152] f32, y [B,1152] bf16, o [B,768] bf16, qc [B,768] f32,
 * stats [8][32][32][2] f32; *step = cached self-attention keys; scratch >= 4096 bytes, zeroed by the caller before the first call
 * (arrival counters, claim stamps, error flags at word 768: 0 = ok); call_index = 0, 1, 2, ... on the same scratch (the epoch
 * of its monotonic counters). */
int dimx_op_layer_chain(const float* qkv, int nslab, long slab_stride, void* sk, void* sv, int T, const void* ck, const void* cv,
                        int Tp, int n_keys, const uint8_t* kmask, const void* w_so, const void* w_cq, const float* colsum_cq,
                        const void* w_co, float* x, void* y, void* o, float* qc, float* stats, int B, const int32_t* step,
                        int call_index, float scale, void* scratch, void* prof, void* stream);
/* tokens = sampler(logits[R,512]) -- see dimx_generate. */
int dimx_op_sample(const float* logits, int R, int top_k, float temperature, const float* exp_noise,
                   uint64_t seed, uint64_t step, int32_t* tokens, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIMX_H */
