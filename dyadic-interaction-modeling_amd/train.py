"""Training step of the DIM-Listener fine-tuning model: reference ``train_epoch``
(code/x_engine_pt.py:9-60) as driven by code/finetune_s2s_pretrain.py:105-143 (AdamW lr 1e-5, clip 1.0, frozen VQ-VAEs).

What is differentiated is the teacher-forced path of ``SLMFT.forward(mode='train')`` (code/seq2seq_pretrain.py:496-514):
encoder_s -> encoder_joint -> norm_s -> context -> AutoregressiveWrapper.forward -> cross entropy.  The continuous
loss has no gradient path in the reference either (its ``pred`` comes from an argmax / one-hot of the logits, :454-464),
and both VQ-VAEs are frozen (:348-366), so the listener code targets and the decoded motion are taken from the HIP engine
(no graph).

Two implementations of that step live in the package:
  * ``dimx.train_hip.HipTrainer`` -- forward + backward + clip + AdamW on hand-written HIP kernels (csrc/train.hip), the
    one ``x_engine_pt.train_epoch`` uses when it is handed a HipTrainer;
  * this module -- the same mathematics on PyTorch-ROCm autograd (rocBLAS / hipBLASLt GEMMs), kept as the
    ``torch.optim`` route of ``train_epoch`` and as the functional layers the LEGACY generator's loop (``legacy_loss``,
    reference code/x_engine.py:8-36) differentiates.  After ``optimizer.step()`` the engine notices the changed
    parameters and re-packs them before its next launch.

Multi-GPU: one process per GPU, replicated weights, per-rank batch shard; ``all_reduce_grads`` averages the gradients in
~64 MiB flat buckets over RCCL (xGMI ring: a few large collectives instead of one per tensor) before clipping.
"""
import math

import torch
import torch.nn.functional as F

from . import dist as ddist

FROZEN_PREFIXES = ("speaker_vq.", "listener_vq.")


def trainable_parameters(model):
    """(name, parameter) of everything the reference trains: all but the two VQ-VAEs (and their ``pe`` buffers)."""
    return [(n, p) for n, p in model.named_parameters() if not n.startswith(FROZEN_PREFIXES)]


def set_trainable(model, flag=True):
    for _, p in trainable_parameters(model):
        p.requires_grad_(flag)
    return model


# ---------------------------------------------------------------------------------------------------------------------
# x-transformers 1.30.16 layers, functional, on a {key: tensor} view of the module's own parameters
# ---------------------------------------------------------------------------------------------------------------------
def _ln(x, w, b=None):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def _attention(x, ctx, P, pre, heads, key_mask=None, attn_mask=None):
    B, n, _ = x.shape
    q = F.linear(x, P[pre + "to_q.weight"]).view(B, n, heads, -1).transpose(1, 2)
    k = F.linear(ctx, P[pre + "to_k.weight"]).view(B, ctx.shape[1], heads, -1).transpose(1, 2)
    v = F.linear(ctx, P[pre + "to_v.weight"]).view(B, ctx.shape[1], heads, -1).transpose(1, 2)
    dots = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    neg = -torch.finfo(dots.dtype).max
    if key_mask is not None:
        dots = dots.masked_fill(~key_mask[:, None, None, :], neg)
    if attn_mask is not None:
        dots = dots.masked_fill(~attn_mask, neg)
    out = torch.matmul(dots.softmax(dim=-1, dtype=torch.float32).to(dots.dtype), v)
    return F.linear(out.transpose(1, 2).reshape(B, n, -1), P[pre + "to_out.weight"])


def _ff(x, P, pre):
    h = F.gelu(F.linear(x, P[pre + "ff.0.0.weight"], P[pre + "ff.0.0.bias"]))
    return F.linear(h, P[pre + "ff.2.weight"], P[pre + "ff.2.bias"])


def xt_encoder(P, pre, x, mask, causal, depth, heads):
    """ContinuousTransformerWrapper(..., return_embeddings=True) with an Encoder of (attention, feed-forward) x depth."""
    T = x.shape[1]
    h = F.linear(x, P[pre + "project_in.weight"], P.get(pre + "project_in.bias"))   # optional tensor: models._adopt_optional_tensors
    h = h + P[pre + "pos_emb.emb.weight"][:T] * (h.shape[-1] ** -0.5)
    am = torch.ones(T, T, dtype=torch.bool, device=x.device).tril() if causal else None
    L = pre + "attn_layers.layers."
    for i in range(depth):
        h = h + _self(h, P, L, 2 * i, heads, mask, am)
        h = h + _ff(_ln(h, P[L + "%d.0.0.weight" % (2 * i + 1)]), P, L + "%d.1." % (2 * i + 1))
    return _ln(h, P[pre + "attn_layers.final_norm.weight"])


def _self(h, P, L, li, heads, key_mask, attn_mask):
    y = _ln(h, P[L + "%d.0.0.weight" % li])
    return _attention(y, y, P, L + "%d.1." % li, heads, key_mask, attn_mask)


def xt_decoder_logits(P, pre, tokens, context, context_mask, self_kv_mask, depth, heads, pos_emb=False):
    """TransformerWrapper(num_tokens, attn_layers=Decoder(cross_attend=True)) on a token prefix -> logits."""
    n = tokens.shape[1]
    h = P[pre + "token_emb.emb.weight"][tokens]
    if pos_emb:
        h = h + P[pre + "pos_emb.emb.weight"][:n] * (h.shape[-1] ** -0.5)
    causal = torch.ones(n, n, dtype=torch.bool, device=tokens.device).tril()
    L = pre + "attn_layers.layers."
    for i in range(depth):
        h = h + _self(h, P, L, 3 * i, heads, self_kv_mask, causal)
        y = _ln(h, P[L + "%d.0.0.weight" % (3 * i + 1)])
        h = h + _attention(y, context, P, L + "%d.1." % (3 * i + 1), heads, context_mask, None)
        h = h + _ff(_ln(h, P[L + "%d.0.0.weight" % (3 * i + 2)]), P, L + "%d.1." % (3 * i + 2))
    h = _ln(h, P[pre + "attn_layers.final_norm.weight"])
    return F.linear(h, P[pre + "to_logits.weight"], P.get(pre + "to_logits.bias"))


def slmft_loss(P, dims, v_speaker, v_audio, mask, z_l, kv_mask=None):
    """Differentiable l_ce_l of SLMFT.forward(mode='train').  P: {key: tensor} (parameters carry the graph); z_l [B,T]
    listener codes with -100 on padding (from the frozen VQ-VAE); kv_mask [B,T-1] keep-mask (AutoregressiveWrapper's
    mask_prob draw) or None.  Returns (loss, logits)."""
    x = v_speaker + P["patch_embed_s"]
    x = xt_encoder(P, "encoder_s.", x, mask, True, dims.enc_depth, dims.heads)
    x = xt_encoder(P, "encoder_joint.", x, mask, True, dims.enc_depth, dims.heads)
    x_s = _ln(x, P["norm_s.weight"], P["norm_s.bias"])
    ctx = torch.cat([x_s + P["patch_embed_dec_s"], v_audio], dim=-1)
    inp = z_l[:, :-1].clamp(min=0)                                   # pad_value 0 where the target is ignored
    logits = xt_decoder_logits(P, "decoder_joint.net.", inp, ctx, mask, kv_mask, dims.dec_depth, dims.heads)
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), z_l[:, 1:].reshape(-1), ignore_index=-100)
    return loss, logits


# ---------------------------------------------------------------------------------------------------------------------
# legacy ListenerGenerator (reference code/seq2seq.py:138-278) -- what code/x_engine.py:8-36 trains
# ---------------------------------------------------------------------------------------------------------------------
LEGACY_TRAINABLE_PREFIXES = ("generator.", "listener_vq.decoder.", "speaker_embeddings.", "listener_embeddings.",
                             "fc_speaker.", "fc_listener.")


def legacy_trainable_parameters(model):
    """reference code/seq2seq.py:165-176: the speaker VQ-VAE and the listener VQ-VAE's encoder + codebook are frozen,
    the listener VQ-VAE's DECODER, the generator and the id-embedding layers train."""
    return [(n, p) for n, p in model.named_parameters() if n.startswith(LEGACY_TRAINABLE_PREFIXES)]


def set_legacy_trainable(model, flag=True):
    for _, p in legacy_trainable_parameters(model):
        p.requires_grad_(flag)
    return model


def _gelu_tanh(x):
    return x * (0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x * x * x))))


def vq_decoder(P, pre, zq, heads, layers, pe):
    """Differentiable TransformerDecoder of the listener VQ-VAE (reference code/models/stage1_BIWI.py:376-393): linear,
    Conv1d(k5, replicate) + LeakyReLU(0.2) + InstanceNorm over time, linear, + pe[batch row], ``layers`` pre-LN
    {attention with packed qkv and scale hidden^-0.5, tanh-GELU MLP} blocks, bias-free output map.  zq [B,L,128]."""
    B, n, _ = zq.shape
    c = pre + "decoder."
    h = F.linear(zq, P[c + "decoder_linear_embedding_pre.net.weight"], P[c + "decoder_linear_embedding_pre.net.bias"])
    x = F.conv1d(F.pad(h.transpose(1, 2), (2, 2), mode="replicate"), P[c + "expander.0.0.weight"], P[c + "expander.0.0.bias"])
    h = F.instance_norm(F.leaky_relu(x, 0.2), eps=1e-5).transpose(1, 2)
    h = F.linear(h, P[c + "decoder_linear_embedding.net.weight"], P[c + "decoder_linear_embedding.net.bias"])
    h = h + pe[:B]
    H = h.shape[-1]
    for i in range(layers):
        a = "%sdecoder_transformer.net.%d.fn." % (c, 2 * i)
        y = F.layer_norm(h, (H,), P[a + "norm.weight"], P[a + "norm.bias"], 1e-5)
        q, k, v = F.linear(y, P[a + "fn.to_qkv.weight"]).view(B, n, 3, heads, H // heads).permute(2, 0, 3, 1, 4)
        att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (H ** -0.5), dim=-1)
        o = torch.matmul(att, v).transpose(1, 2).reshape(B, n, H)
        h = h + F.linear(o, P[a + "fn.to_out.weight"], P[a + "fn.to_out.bias"])
        m = "%sdecoder_transformer.net.%d.fn." % (c, 2 * i + 1)
        y = F.layer_norm(h, (H,), P[m + "norm.weight"], P[m + "norm.bias"], 1e-5)
        y = _gelu_tanh(F.linear(y, P[m + "fn.l1.weight"], P[m + "fn.l1.bias"]))
        h = h + F.linear(y, P[m + "fn.l2.weight"], P[m + "fn.l2.bias"])
    return F.linear(h, P[c + "vertice_map_reverse.weight"])


def legacy_loss(P, dims, vq_dims, x_speaker, z_l, v_listener, mask, pe, speaker_ids=None, listener_ids=None):
    """Differentiable ``ListenerGenerator.forward`` (reference code/seq2seq.py:235-278 with Transformer.forward :46-67).
    x_speaker [B,T,1024] and z_l [B,T] (-100 on padding) come from the frozen VQ-VAEs (HIP engine, no graph); ``pe`` is
    the listener decoder's positional buffer.  ``listener_ids`` (the call of code/x_engine.py:24) prepends
    fc_listener(relu(listener_embeddings[id])) to the encoder output, a True to the context mask and a -100 to the
    targets; ``speaker_ids`` prepends fc_speaker(relu(speaker_embeddings[id])) to the encoder INPUT.
    Returns (loss = cross entropy + continuous loss, pred_cont_seq [B,T-1,56], logits [B,T-1,512])."""
    B = x_speaker.shape[0]
    one = torch.ones(B, 1, dtype=torch.bool, device=mask.device)
    cmask = mask
    if speaker_ids is not None:
        sid = F.linear(F.relu(P["speaker_embeddings.weight"][speaker_ids]), P["fc_speaker.weight"], P["fc_speaker.bias"])
        x_speaker = torch.cat([sid[:, None], x_speaker], dim=1)
        cmask = torch.cat([one, cmask], dim=1)
    enc = xt_encoder(P, "generator.encoder.", x_speaker, cmask, False, dims.depth, dims.heads)
    tgt = z_l
    if listener_ids is not None:
        lid = F.linear(F.relu(P["listener_embeddings.weight"][listener_ids]), P["fc_listener.weight"], P["fc_listener.bias"])
        enc = torch.cat([lid[:, None], enc], dim=1)
        cmask = torch.cat([one, cmask], dim=1)
        tgt = torch.cat([torch.full_like(z_l[:, :1], -100), z_l], dim=1)
    inp, target = tgt[:, :-1].clamp(min=0), tgt[:, 1:]
    logits = xt_decoder_logits(P, "generator.decoder.net.", inp, enc, cmask, None, dims.depth, dims.heads, pos_emb=True)
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), target.reshape(-1), ignore_index=-100)
    if listener_ids is not None:
        logits = logits[:, 1:]
    zq = P["listener_vq.quantize.embedding.weight"][logits.argmax(-1)]
    pred = vq_decoder(P, "listener_vq.", zq, vq_dims.heads, vq_dims.layers, pe)
    m = mask[:, 1:].reshape(-1)
    p = pred.reshape(-1, pred.shape[-1])[m]
    t = v_listener[:, 1:].reshape(-1, pred.shape[-1])[m]
    loss_cont = F.pairwise_distance(p[:, 6:], t[:, 6:]).mean() + F.pairwise_distance(p[:, :6], t[:, :6]).mean()
    return loss + loss_cont, pred, logits


# ---------------------------------------------------------------------------------------------------------------------
# SLM pre-training (reference code/seq2seq_pretrain.py:72-323) -- what code/train_s2s_pretrain.py:41-64 trains with
# x_engine_pt.train_epoch
# ---------------------------------------------------------------------------------------------------------------------
SLM_FROZEN_PREFIXES = ("speaker_vq.encoder.", "speaker_vq.quantize.", "listener_vq.encoder.", "listener_vq.quantize.")


def slm_trainable_parameters(model):
    """reference :98-113: both VQ-VAEs' encoders and codebooks are frozen, their DECODERS and everything else train."""
    return [(n, p) for n, p in model.named_parameters() if not n.startswith(SLM_FROZEN_PREFIXES)]


def set_slm_trainable(model, flag=True):
    for _, p in slm_trainable_parameters(model):
        p.requires_grad_(flag)
    return model


def _masked_pairwise(pred, target, sel):
    m = sel[:, 1:].reshape(-1)
    p = pred.reshape(-1, pred.shape[-1])[m]
    t = target[:, 1:].reshape(-1, pred.shape[-1])[m]
    return F.pairwise_distance(p[:, 6:], t[:, 6:]).mean() + F.pairwise_distance(p[:, :6], t[:, :6]).mean()


def slm_loss(P, dims, vq_dims, v_speaker, v_listener, v_audio, mask, mask_speaker, mask_listener, z_s, z_l, pe_s, pe_l):
    """Differentiable ``SLM.forward`` (reference :300-323): masked speaker / listener streams through encoder_s / encoder_l
    (bidirectional, key padding), the joint encoder over the 2T concatenation and over each stream alone (:200-221), InfoNCE
    between the clip means (:270-289), the two cross-predicting decoders with absolute positional embedding (z_s from the
    listener half of x_joint, z_l from the speaker half, :223-243), and the continuous losses of the decoded arg-max codes,
    which train the two VQ-VAE DECODERS.  z_s / z_l [B,T]: codes from the frozen VQ encoders (HIP engine), unmasked;
    mask_speaker / mask_listener: True = masked frame (its code is a target).  Returns (total, dict)."""
    depth, heads = dims.enc_depth, dims.heads
    vs = (v_speaker + P["patch_embed_s"]).masked_fill(mask_speaker[..., None], 0.0)
    vl = (v_listener + P["patch_embed_l"]).masked_fill(mask_listener[..., None], 0.0)
    x_s = xt_encoder(P, "encoder_s.", vs, mask, False, depth, heads)
    x_l = xt_encoder(P, "encoder_l.", vl, mask, False, depth, heads)
    x_joint = xt_encoder(P, "encoder_joint.", torch.cat([x_s, x_l], dim=1), torch.cat([mask, mask], dim=-1), False, depth, heads)
    x_l = xt_encoder(P, "encoder_joint.", x_l, mask, False, depth, heads)
    x_s = xt_encoder(P, "encoder_joint.", x_s, mask, False, depth, heads)
    x_s, x_l = _ln(x_s, P["norm_s.weight"], P["norm_s.bias"]), _ln(x_l, P["norm_l.weight"], P["norm_l.bias"])
    x_joint = _ln(x_joint, P["norm.weight"], P["norm.bias"])
    valid = mask[..., None].to(x_s.dtype)
    n = valid.sum(1)
    s = F.normalize((x_s * valid).sum(1) / n, dim=-1)
    l_ = F.normalize((x_l * valid).sum(1) / n, dim=-1)
    total = s @ l_.t() / 0.05
    nce = -torch.mean(torch.diag(F.log_softmax(total, dim=0)))
    c_acc = (F.softmax(total.detach(), dim=0).argmax(0) == torch.arange(total.shape[0], device=total.device)).sum() / total.shape[0]
    T = mask.shape[1]
    out = {}
    for tag, z, msk, xj, patch, vq, pe, tgt in (("s", z_s, mask_speaker, x_joint[:, T:], "patch_embed_dec_l", "speaker_vq.", pe_s, v_speaker),
                                                ("l", z_l, mask_listener, x_joint[:, :T], "patch_embed_dec_s", "listener_vq.", pe_l, v_listener)):
        z = torch.where(msk, z, torch.full_like(z, -100))
        ctx = torch.cat([xj + P[patch], v_audio], dim=-1)
        logits = xt_decoder_logits(P, "decoder_joint.net.", z[:, :-1].clamp(min=0), ctx, mask, None, dims.dec_depth, heads, pos_emb=True)
        out["l_ce_" + tag] = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), z[:, 1:].reshape(-1), ignore_index=-100)
        pred = vq_decoder(P, vq, P[vq + "quantize.embedding.weight"][logits.argmax(-1)], vq_dims.heads, vq_dims.layers, pe)
        out["l_cont_" + tag] = _masked_pairwise(pred, tgt, msk)
    out["nce"], out["c_acc"] = nce, c_acc
    return out["l_ce_s"] + out["l_ce_l"] + out["l_cont_s"] + out["l_cont_l"] + nce, out


# ---------------------------------------------------------------------------------------------------------------------
# gradient synchronisation
# ---------------------------------------------------------------------------------------------------------------------
def all_reduce_grads(params, bucket_bytes=64 << 20):
    """Average .grad over the ranks of the default process group in flat buckets (sum all-reduce, then / world).
    xGMI is point-to-point, ring collectives are per-link bound: few large buckets beat one collective per tensor."""
    world = ddist.world_size()
    if world == 1:
        return 0
    import torch.distributed as dist
    # every trainable parameter takes part on every rank, in the same order: a parameter without a gradient on this rank
    # (unused in this batch) contributes zeros -- buckets built from "has a grad" could differ between ranks and hang
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    n_coll, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (size == 0 or size + grads[j].numel() * grads[j].element_size() <= bucket_bytes):
            size += grads[j].numel() * grads[j].element_size()
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat)
        flat.div_(world)
        o = 0
        for g in grads[i:j]:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        n_coll += 1
        i = j
    return n_coll


def broadcast_parameters(params, src=0):
    """Make every rank start from rank `src`'s parameters (flat buckets, like the gradients); without it the job relies on
    bit-identical construction on every rank."""
    if ddist.world_size() == 1:
        return 0
    import torch.distributed as dist
    n = 0
    with torch.no_grad():
        i = 0
        while i < len(params):
            j, size = i, 0
            while j < len(params) and (size == 0 or size + params[j].numel() * params[j].element_size() <= (64 << 20)):
                size += params[j].numel() * params[j].element_size()
                j += 1
            flat = torch.cat([p.detach().reshape(-1) for p in params[i:j]])
            dist.broadcast(flat, src)
            o = 0
            for p in params[i:j]:
                p.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            n += 1
            i = j
    return n


assert_same_batch_count = ddist.assert_same_batch_count   # lives in dist.py: the HIP training loops use it without importing this module


def train_step(model, optimizer, v_speaker, v_listener, v_audio, mask, clip=1.0, kv_mask=None, scheduler=None):
    """One optimisation step exactly as the reference's loop body (zero_grad, forward(mode='train'), backward, clip,
    step), with the gradient all-reduce in between for N > 1."""
    optimizer.zero_grad()
    loss, d, pred = model(v_speaker, v_listener, v_audio, mask, mode="train", kv_mask=kv_mask)
    loss.mean().backward()
    params = [p for _, p in trainable_parameters(model)]
    all_reduce_grads(params)
    if clip > 0:
        torch.nn.utils.clip_grad_norm_(params, clip)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), d, pred


def make_optimizer(model, lr=1e-5):
    """reference code/finetune_s2s_pretrain.py:119: AdamW over the trainable parameters."""
    set_trainable(model, True)
    return torch.optim.AdamW([p for _, p in trainable_parameters(model)], lr=lr)
