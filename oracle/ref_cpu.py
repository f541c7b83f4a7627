"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU float32 restatement of the arithmetic of the DIM-Listener hot
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product path (``dimx``) never does and
fails loudly when its HIP library is missing.

Every function works on a flat ``state_dict`` (name -> tensor) with the reference's
key names, and cites the reference lines it follows (paths relative to
``/root/reference``).

Parity status
-------------
* VQ-VAE half (encode / quantise / decode, ``forward_vq`` re-enactment): PINNED.
  ``tests/golden/make_golden.py`` imports the reference classes in the build
  container, loads the same regenerated weights, and the committed fixtures
  (``tests/golden/vq_*.npz``) are checked against this file by
  ``tests/test_oracle_golden.py``.
* x-transformers half (encoders, cross-attending decoder, generate): PARITY UNPINNED.
  ``x-transformers==1.30.16`` (``code/requirements.txt:99``) is neither vendored in the
  reference nor installable here, and no checkpoint or golden vector exists for it.
  The arithmetic below restates the library's published algorithm (SURVEY.md
  Appendix A.2); it is pinned only by self-consistency properties (cached ==
  uncached == teacher-forced, masking invariance) and can be compared with the real
  wheel via ``tools/verify_against_xtransformers.py`` wherever that wheel exists.
* Legacy ``ListenerGenerator`` (``code/seq2seq.py:138-306``): the speaker VQ-VAE encoder ->
  quantiser -> ``x_speaker`` construction is PINNED by ``tests/golden/legacy_speaker_features.npz`` (reference
  ``VQSpeakerAutoEncoder`` imported by ``tests/golden/make_golden.py --legacy``; checked by
  ``tests/test_oracle_legacy.py``).  Its x-transformers encoder/decoder (dim 512, absolute positional embedding)
  shares the PARITY UNPINNED status and the self-consistency pins of the stage above.
* ``SLM`` pre-training forward (``code/seq2seq_pretrain.py:58-323``): VQ-VAE halves pinned as above,
  the x-transformers stage PARITY UNPINNED (``tests/test_oracle_slm.py``: structural properties only).
"""
import math

import torch
import torch.nn.functional as F

NEG_SLOPE = 0.2
LN_EPS = 1e-5


# ----------------------------------------------------------------------------
# VQ-VAE  (code/models/stage1_BIWI.py, code/models/lib/base_models.py, quantizer.py)
# ----------------------------------------------------------------------------

def gelu_tanh(x):
    """code/utils/base_model_util.py:81-94 (tanh approximation, float32)."""
    c = math.sqrt(2.0 / math.pi)
    return x * (0.5 * (1.0 + torch.tanh(c * (x + 0.044715 * torch.pow(x, 3)))))


def _lin(x, sd, name, bias=True):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"] if bias else None)


def vq_conv_block(h, w, b):
    """Conv1d(k=5, replicate pad 2) -> LeakyReLU(0.2) -> InstanceNorm1d(no affine)
    over the time axis.  code/models/stage1_BIWI.py:263-267 / :330-334.  h: [B,L,C]."""
    x = h.permute(0, 2, 1)
    x = F.conv1d(F.pad(x, (2, 2), mode="replicate"), w, b)
    x = F.leaky_relu(x, NEG_SLOPE)
    x = F.instance_norm(x, eps=1e-5)
    return x.permute(0, 2, 1)


def vq_attention(y, sd, p, heads, hidden):
    """code/models/lib/base_models.py:125-146: packed qkv (qkv,h,d), scale = hidden^-0.5,
    no mask on this path."""
    B, L, _ = y.shape
    qkv = F.linear(y, sd[p + "to_qkv.weight"])
    qkv = qkv.view(B, L, 3, heads, hidden // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    dots = torch.einsum("bhid,bhjd->bhij", q, k) * (hidden ** -0.5)
    attn = F.softmax(dots, dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(B, L, hidden)
    return F.linear(out, sd[p + "to_out.weight"], sd[p + "to_out.bias"])


def vq_stack(h, sd, prefix, layers, heads):
    """6 x pre-LN {Attention, MLP} residual blocks, no final norm.
    code/models/lib/base_models.py:149-199."""
    hidden = h.shape[-1]
    for i in range(layers):
        a = "{}net.{}.fn.".format(prefix, 2 * i)
        y = F.layer_norm(h, (hidden,), sd[a + "norm.weight"], sd[a + "norm.bias"], LN_EPS)
        h = h + vq_attention(y, sd, a + "fn.", heads, hidden)
        m = "{}net.{}.fn.".format(prefix, 2 * i + 1)
        y = F.layer_norm(h, (hidden,), sd[m + "norm.weight"], sd[m + "norm.bias"], LN_EPS)
        h = h + _lin(gelu_tanh(_lin(y, sd, m + "fn.l1")), sd, m + "fn.l2")
    return h


def _pe_rows(sd, key, B, row_offset=0, batch_rows=True):
    """PositionalEncoding adds pe[:B] ([B,1,H]) to x[B,L,H]: the sinusoid of the BATCH
    ROW index, the same for every time step (code/models/lib/base_models.py:271-273)."""
    pe = sd[key]                      # [5000,1,H]
    if batch_rows:
        return pe[row_offset:row_offset + B]
    return pe[0:1].expand(B, -1, -1)


def vq_encode_features(sd, x, heads=8, layers=6, prefix="", batch_rows=True, row_offset=0):
    """TransformerEncoder.forward, code/models/stage1_BIWI.py:307-317.  x: [B,L,56] -> [B,L,128]."""
    e = prefix + "encoder."
    h = F.leaky_relu(_lin(x, sd, e + "vertice_mapping.0"), NEG_SLOPE)
    h = vq_conv_block(h, sd[e + "squasher.0.0.weight"], sd[e + "squasher.0.0.bias"])
    h = _lin(h, sd, e + "encoder_linear_embedding.net")
    h = h + _pe_rows(sd, e + "encoder_pos_embedding.pe", x.shape[0], row_offset, batch_rows)
    h = vq_stack(h, sd, e + "encoder_transformer.", layers, heads)
    return _lin(h, sd, e + "encoder_linear_embedding_post.net")


def vq_distances(z, E):
    """code/models/lib/quantizer.py:38-40, same association: (sum z^2 + sum e^2) - 2 z.E^T."""
    return torch.sum(z ** 2, dim=1, keepdim=True) + torch.sum(E ** 2, dim=1) - 2 * torch.matmul(z, E.t())


def vq_quantize(z, E):
    """argmin over the codebook, first index on ties (quantizer.py:45).  z: [N,128]."""
    d = vq_distances(z, E)
    idx = torch.argmin(d, dim=1)
    return idx, d


def vq_margins(d):
    """second-best minus best distance per row (diagnostic for fixture capture)."""
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    return top2[:, 1] - top2[:, 0]


def vq_encode(sd, x, prefix="", batch_rows=True, row_offset=0, return_all=False, heads=8, layers=6):
    """VQAutoEncoder.encode, code/models/stage1_BIWI.py:22-27.  Returns idx [B,L] int64."""
    B, L, _ = x.shape
    z = vq_encode_features(sd, x, heads, layers, prefix, batch_rows, row_offset)
    idx, d = vq_quantize(z.reshape(B * L, -1), sd[prefix + "quantize.embedding.weight"])
    if return_all:
        return idx.view(B, L), z, d
    return idx.view(B, L)


def vq_decode(sd, idx, prefix="", row_offset=0, heads=8, layers=6):
    """Codebook lookup (the one-hot matmul of code/seq2seq_pretrain.py:458-460 is a
    gather) + VQAutoEncoder.decode / TransformerDecoder.forward
    (code/models/stage1_BIWI.py:29-37, :376-393).  idx: [B,L] -> [B,L,56].  The
    positional row is the batch row of THIS call (+ row_offset)."""
    c = prefix + "decoder."
    zq = sd[prefix + "quantize.embedding.weight"][idx]           # [B,L,128]
    h = _lin(zq, sd, c + "decoder_linear_embedding_pre.net")
    h = vq_conv_block(h, sd[c + "expander.0.0.weight"], sd[c + "expander.0.0.bias"])
    h = _lin(h, sd, c + "decoder_linear_embedding.net")
    h = h + _pe_rows(sd, c + "decoder_pos_embedding.pe", idx.shape[0], row_offset, True)
    h = vq_stack(h, sd, c + "decoder_transformer.", layers, heads)
    return F.linear(h, sd[c + "vertice_map_reverse.weight"])


def forward_vq(sd, v_speaker, v_listener, mask, speaker_prefix="speaker_vq.",
               listener_prefix="listener_vq.", with_speaker=True):
    """SLMFT.forward_vq, code/seq2seq_pretrain.py:480-494: per-sample batch-1 encodes of the
    valid frames, speaker codes padded with 0, listener codes padded with -100."""
    B, T, _ = v_speaker.shape
    zs, zl = [], []
    for i in range(B):
        if with_speaker:
            s = vq_encode(sd, v_speaker[i][mask[i]].unsqueeze(0), speaker_prefix)[0]
            zs.append(F.pad(s, (0, T - s.shape[-1]), value=0))
        l = vq_encode(sd, v_listener[i][mask[i]].unsqueeze(0), listener_prefix)[0]
        zl.append(F.pad(l, (0, T - l.shape[-1]), value=-100))
    z_l = torch.stack(zl, 0)
    z_s = torch.stack(zs, 0) if with_speaker else None
    return z_s, z_l


# ----------------------------------------------------------------------------
# x-transformers 1.30.16 stage (restated; see module docstring: parity unpinned)
# ----------------------------------------------------------------------------

def _neg_max(t):
    return -torch.finfo(t.dtype).max


def xt_attention(x, context, sd, p, heads, key_mask=None, attn_mask=None, query_mask=None,
                 kv=None, zero_masked_queries=True):
    """x-transformers Attention + Attend(flash=False): q/k/v/out Linear without bias,
    dim_head 64, scale 64^-0.5, masked positions filled with -finfo.max, softmax fp32.
    key_mask [B,J] True=keep; attn_mask [I,J] or [B,1,I,J] True=attend; query_mask [B,I].
    ``kv`` optionally supplies precomputed (k, v) [B,H,J,64]."""
    B, I, _ = x.shape
    dh = 64
    q = F.linear(x, sd[p + "to_q.weight"]).view(B, I, heads, dh).permute(0, 2, 1, 3)
    if kv is None:
        src = x if context is None else context
        J = src.shape[1]
        k = F.linear(src, sd[p + "to_k.weight"]).view(B, J, heads, dh).permute(0, 2, 1, 3)
        v = F.linear(src, sd[p + "to_v.weight"]).view(B, J, heads, dh).permute(0, 2, 1, 3)
    else:
        k, v = kv
    dots = torch.einsum("bhid,bhjd->bhij", q, k) * (dh ** -0.5)
    keep = None
    if key_mask is not None:
        keep = key_mask[:, None, None, :]
    if attn_mask is not None:
        am = attn_mask if attn_mask.dim() == 4 else attn_mask[None, None]
        keep = am if keep is None else (keep & am)
    if keep is not None:
        dots = dots.masked_fill(~keep, _neg_max(dots))
    attn = F.softmax(dots, dim=-1, dtype=torch.float32).to(dots.dtype)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(B, I, heads * dh)
    out = F.linear(out, sd[p + "to_out.weight"])
    if query_mask is not None and zero_masked_queries:
        out = out.masked_fill(~query_mask[..., None], 0.0)
    return out, (k, v)


def xt_ff(y, sd, p):
    """FeedForward: Linear(bias) -> exact erf GELU -> Linear(bias)."""
    return F.linear(F.gelu(F.linear(y, sd[p + "ff.0.0.weight"], sd[p + "ff.0.0.bias"])),
                    sd[p + "ff.2.weight"], sd[p + "ff.2.bias"])


def _xt_norm(x, sd, key):
    """x-transformers LayerNorm = nn.LayerNorm(dim, bias=False)."""
    return F.layer_norm(x, (x.shape[-1],), sd[key], None, LN_EPS)


def xt_encoder(sd, prefix, x, mask, causal=True, depth=4, heads=12, zero_masked_queries=True):
    """ContinuousTransformerWrapper.forward(x, mask, attn_mask, return_embeddings=True):
    project_in (bias-free in x-transformers 1.30.16 as restated in SURVEY A.2; that detail is marked [XT?] there, so a
    ``project_in.bias`` / ``to_logits.bias`` tensor in the state dict is applied when present) + learned abs pos emb *
    dim^-0.5 + depth x {attn, ff} pre-norm
    + final norm; project_out is skipped.  Call sites code/seq2seq_pretrain.py:439-440."""
    B, T, _ = x.shape
    h = F.linear(x, sd[prefix + "project_in.weight"], sd.get(prefix + "project_in.bias"))
    dim = h.shape[-1]
    h = h + sd[prefix + "pos_emb.emb.weight"][:T] * (dim ** -0.5)
    attn_mask = None
    if causal:
        attn_mask = ~torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1)
    for i in range(depth):
        pa = "{}attn_layers.layers.{}.".format(prefix, 2 * i)
        y = _xt_norm(h, sd, pa + "0.0.weight")
        o, _ = xt_attention(y, None, sd, pa + "1.", heads, key_mask=mask, attn_mask=attn_mask,
                            query_mask=mask, zero_masked_queries=zero_masked_queries)
        h = h + o
        pf = "{}attn_layers.layers.{}.".format(prefix, 2 * i + 1)
        h = h + xt_ff(_xt_norm(h, sd, pf + "0.0.weight"), sd, pf + "1.")
    return _xt_norm(h, sd, prefix + "attn_layers.final_norm.weight")


def slmft_forward_encoder(sd, v_speaker, mask, zero_masked_queries=True):
    """SLMFT.forward_encoder, code/seq2seq_pretrain.py:431-442."""
    x = v_speaker + sd["patch_embed_s"]
    x = xt_encoder(sd, "encoder_s.", x, mask, True, zero_masked_queries=zero_masked_queries)
    x = xt_encoder(sd, "encoder_joint.", x, mask, True, zero_masked_queries=zero_masked_queries)
    return F.layer_norm(x, (x.shape[-1],), sd["norm_s.weight"], sd["norm_s.bias"], LN_EPS)


def slmft_context(sd, x_s, v_audio):
    """code/seq2seq_pretrain.py:445-446: context = cat(x_s + patch_embed_dec_s, audio)."""
    return torch.cat([x_s + sd["patch_embed_dec_s"], v_audio], dim=-1)


def xt_decoder_layers(sd, prefix, h, context, context_mask, self_attn_mask, self_kv_mask,
                      depth=4, heads=12, cache=None):
    """Decoder(cross_attend=True): depth x {causal self-attn, cross-attn, ff}, pre-norm,
    then final norm.  ``cache`` (list of dicts) enables incremental decoding: self K/V are
    appended, cross K/V computed once."""
    for i in range(depth):
        ps = "{}attn_layers.layers.{}.".format(prefix, 3 * i)
        y = _xt_norm(h, sd, ps + "0.0.weight")
        if cache is None:
            o, _ = xt_attention(y, None, sd, ps + "1.", heads, key_mask=self_kv_mask, attn_mask=self_attn_mask)
        else:
            c = cache[i]
            B, n, _ = y.shape
            k_new = F.linear(y, sd[ps + "1.to_k.weight"]).view(B, n, heads, 64).permute(0, 2, 1, 3)
            v_new = F.linear(y, sd[ps + "1.to_v.weight"]).view(B, n, heads, 64).permute(0, 2, 1, 3)
            c["k"] = k_new if c.get("k") is None else torch.cat([c["k"], k_new], 2)
            c["v"] = v_new if c.get("v") is None else torch.cat([c["v"], v_new], 2)
            o, _ = xt_attention(y, None, sd, ps + "1.", heads, kv=(c["k"], c["v"]))
        h = h + o
        pc = "{}attn_layers.layers.{}.".format(prefix, 3 * i + 1)
        y = _xt_norm(h, sd, pc + "0.0.weight")
        if cache is None:
            o, _ = xt_attention(y, context, sd, pc + "1.", heads, key_mask=context_mask)
        else:
            c = cache[i]
            if c.get("ck") is None:
                J = context.shape[1]
                B = context.shape[0]
                c["ck"] = F.linear(context, sd[pc + "1.to_k.weight"]).view(B, J, heads, 64).permute(0, 2, 1, 3)
                c["cv"] = F.linear(context, sd[pc + "1.to_v.weight"]).view(B, J, heads, 64).permute(0, 2, 1, 3)
            o, _ = xt_attention(y, None, sd, pc + "1.", heads, key_mask=context_mask, kv=(c["ck"], c["cv"]))
        h = h + o
        pf = "{}attn_layers.layers.{}.".format(prefix, 3 * i + 2)
        h = h + xt_ff(_xt_norm(h, sd, pf + "0.0.weight"), sd, pf + "1.")
    return _xt_norm(h, sd, prefix + "attn_layers.final_norm.weight")


def xt_decoder_logits(sd, tokens, context, context_mask, self_kv_mask=None,
                      prefix="decoder_joint.net.", depth=4, heads=12):
    """TransformerWrapper.forward for SLMFT: token embedding, NO positional embedding
    (use_abs_pos_emb=False, code/seq2seq_pretrain.py:386), decoder layers, to_logits."""
    B, n = tokens.shape
    h = sd[prefix + "token_emb.emb.weight"][tokens]
    causal = ~torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
    h = xt_decoder_layers(sd, prefix, h, context, context_mask, causal, self_kv_mask, depth, heads)
    return F.linear(h, sd[prefix + "to_logits.weight"], sd.get(prefix + "to_logits.bias"))


def ar_kv_mask(B, T, mask_prob=0.15, generator=None):
    """AutoregressiveWrapper.forward key masking: int(T*mask_prob) random key positions
    (never position 0) are hidden from self-attention.  Returns keep-mask [B,T-1]."""
    n = T - 1
    rand = torch.randn(B, n, generator=generator)
    rand[:, 0] = -torch.finfo(rand.dtype).max
    num_mask = min(int(T * mask_prob), T - 1)
    idx = rand.topk(num_mask, dim=-1).indices
    return ~torch.zeros(B, n).scatter(1, idx, 1.0).bool()


def ar_forward(sd, z, context, context_mask, kv_mask=None, ignore_index=-100, pad_value=0):
    """AutoregressiveWrapper.forward(z, context, context_mask, return_outputs=True):
    inp = z[:, :-1] with ignore_index -> pad_value, CE against z[:, 1:].
    Returns (loss, logits[B,T-1,512])."""
    inp, target = z[:, :-1], z[:, 1:]
    inp = torch.where(inp == ignore_index, torch.full_like(inp, pad_value), inp)
    logits = xt_decoder_logits(sd, inp, context, context_mask, kv_mask)
    loss = F.cross_entropy(logits.permute(0, 2, 1), target, ignore_index=ignore_index)
    return loss, logits


def top_k_filter(logits, k=52):
    """x-transformers top_k with frac_num_tokens 0.1: k = ceil(0.1*512) = 52."""
    val, ind = torch.topk(logits, k, dim=-1)
    out = torch.full_like(logits, float("-inf"))
    return out.scatter(-1, ind, val)


def sample_tokens(logits, noise=None, temperature=1.0, k=52):
    """One sampling step.  noise=None -> greedy argmax of the raw logits; otherwise
    argmax(softmax(top_k(logits)/T) / noise) with noise ~ Exp(1), which is what
    torch.multinomial(probs, 1) computes on CPU (fixture sampler_multinomial.npz)."""
    if noise is None:
        return logits.argmax(dim=-1)
    probs = F.softmax(top_k_filter(logits, k) / temperature, dim=-1)
    return (probs / noise).argmax(dim=-1)


def ar_generate(sd, start, seq_len, context, context_mask, noise=None, temperature=1.0, k=52,
                cached=True, prefix="decoder_joint.net.", depth=4, heads=12, return_logits=False):
    """AutoregressiveWrapper.generate(prompts=start[:,None], seq_len, context, context_mask)
    with KV cache.  noise: [seq_len,B,512] Exp(1) or None for greedy.  Returns the seq_len
    generated tokens [B,seq_len] (the prompt is stripped, as the library does)."""
    B = start.shape[0]
    out = start.view(B, 1)
    cache = [dict() for _ in range(depth)] if cached else None
    all_logits = []
    for t in range(seq_len):
        if cached:
            h = sd[prefix + "token_emb.emb.weight"][out[:, -1:]]
            h = xt_decoder_layers(sd, prefix, h, context, context_mask, None, None, depth, heads, cache)
            logits = F.linear(h[:, -1], sd[prefix + "to_logits.weight"], sd.get(prefix + "to_logits.bias"))
        else:
            logits = xt_decoder_logits(sd, out, context, context_mask, None, prefix, depth, heads)[:, -1]
        if return_logits:
            all_logits.append(logits)
        tok = sample_tokens(logits, None if noise is None else noise[t], temperature, k)
        out = torch.cat([out, tok.view(B, 1)], dim=1)
    toks = out[:, 1:]
    if return_logits:
        return toks, torch.stack(all_logits, 1)
    return toks


def continuous_loss(pred, target, mask):
    """SLMFT.forward_continuous_loss, code/seq2seq_pretrain.py:466-478."""
    target = target[:, 1:, :]
    m = mask[:, 1:].reshape(-1)
    p = pred.reshape(-1, pred.shape[-1])[m]
    t = target.reshape(-1, target.shape[-1])[m]
    d_pose = F.pairwise_distance(p[:, 0:6], t[:, 0:6])
    d_exp = F.pairwise_distance(p[:, 6:], t[:, 6:])
    return torch.mean(d_exp) + torch.mean(d_pose)


def slmft_forward(sd, v_speaker, v_listener, v_audio, mask, mode="train", noise=None,
                  kv_mask=None, temperature=1.0, return_aux=False):
    """SLMFT.forward, code/seq2seq_pretrain.py:496-514.  Necessary work only: one listener
    VQ encode per clip (the reference's duplicated forward_vq and unused speaker codes do
    not change the result).  mode 'train': teacher-forced logits -> argmax; mode 'val':
    generate from the ground-truth first listener code.  Randomness is injected:
    ``kv_mask`` [B,T-1] keep-mask for 'train' (None = no key masking), ``noise``
    [T-1,B,512] Exp(1) for 'val' (None = greedy)."""
    _, z_l = forward_vq(sd, v_speaker, v_listener, mask, with_speaker=False)
    x_s = slmft_forward_encoder(sd, v_speaker, mask)
    ctx = slmft_context(sd, x_s, v_audio)
    logits = None
    if mode == "train":
        l_ce, logits = ar_forward(sd, z_l, ctx, mask, kv_mask)
        tokens = logits.argmax(dim=-1)
    else:
        tokens = ar_generate(sd, z_l[:, 0], z_l.shape[1] - 1, ctx, mask, noise, temperature)
        l_ce = torch.zeros(())
    pred = vq_decode(sd, tokens, "listener_vq.")
    l_cont = continuous_loss(pred, v_listener, mask)
    total = l_ce + l_cont
    d = {"l_ce_s": 0, "l_ce_l": l_ce, "l_cont_s": 0, "l_cont_l": l_cont, "nce": 0, "c_acc": 0}
    if return_aux:
        return total, d, pred, {"z_l": z_l, "x_s": x_s, "ctx": ctx, "logits": logits, "tokens": tokens}
    return total, d, pred


# ----------------------------------------------------------------------------
# legacy ListenerGenerator (code/seq2seq.py) -- SURVEY.md section 8(f1); x-transformers half unpinned as above
# ----------------------------------------------------------------------------

def speaker_vq_encode_quant(sd, x, prefix="speaker_vq.", heads=8, layers=6, fqn=8, zdim=128):
    """VQSpeakerAutoEncoder.encode(x)[0] (code/models/stage1_BIWI.py:152-157): encoder features [B,L,fqn*zdim]
    viewed as [B, L*fqn, zdim], quantised, returned as the codebook vectors permuted to [B, zdim, L*fqn]
    (quantizer.py:65).  Also returns the indices [B, L*fqn]."""
    B, L, _ = x.shape
    h = vq_encode_features(sd, x, heads, layers, prefix, True, 0)      # [B, L, fqn*zdim]
    z = h.view(B, L * fqn, zdim)
    E = sd[prefix + "quantize.embedding.weight"]
    idx, d = vq_quantize(z.reshape(B * L * fqn, zdim), E)
    quant = E[idx].view(B, L * fqn, zdim).permute(0, 2, 1).contiguous()
    return quant, idx.view(B, L * fqn), d


def legacy_speaker_features(sd, v_speaker, mask, fqn=8, zdim=128):
    """The x_speaker construction of ListenerGenerator.forward / .generate (code/seq2seq.py:224-241): per-sample
    batch-1 encode of the valid frames, zero-pad of the [1,128,len*8] code-vector tensor along its LAST axis to
    T*8, then a raw .view(B,-1,8,128).view(B,-1,1024) of that channel-major memory (a reinterpretation, not a
    transpose -- reproduced literally)."""
    B, T, _ = v_speaker.shape
    xs = []
    for i in range(B):
        q, _, _ = speaker_vq_encode_quant(sd, v_speaker[i][mask[i]].unsqueeze(0), "speaker_vq.", fqn=fqn, zdim=zdim)
        xs.append(F.pad(q, (0, T * fqn - q.shape[-1]), value=0))
    x = torch.cat(xs, dim=0)                                          # [B, 128, T*8] contiguous
    x = x.view(B, -1, fqn, zdim).contiguous()
    return x.view(B, -1, fqn * zdim).contiguous()                     # [B, T, 1024]


def legacy_decoder_embed(sd, tokens, prefix="generator.decoder.net."):
    """TransformerWrapper with use_abs_pos_emb=True (code/seq2seq.py:39): token emb + pos_emb[0:n] * dim^-0.5."""
    n = tokens.shape[1]
    h = sd[prefix + "token_emb.emb.weight"][tokens]
    return h + sd[prefix + "pos_emb.emb.weight"][:n] * (h.shape[-1] ** -0.5)


def legacy_decoder_logits(sd, tokens, context, context_mask, prefix="generator.decoder.net.", depth=6, heads=8):
    n = tokens.shape[1]
    h = legacy_decoder_embed(sd, tokens, prefix)
    causal = ~torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
    h = xt_decoder_layers(sd, prefix, h, context, context_mask, causal, None, depth, heads)
    return F.linear(h, sd[prefix + "to_logits.weight"], sd.get(prefix + "to_logits.bias"))


def legacy_generate(sd, start, seq_len, context, context_mask, noise=None, temperature=1.0, k=52,
                    prefix="generator.decoder.net.", depth=6, heads=8):
    """AutoregressiveWrapper.generate for the legacy decoder (positional embedding of the absolute position);
    uncached recomputation per step would be O(T^2) -- this keeps the KV cache like the library."""
    B = start.shape[0]
    out = start.view(B, 1)
    cache = [dict() for _ in range(depth)]
    pos = sd[prefix + "pos_emb.emb.weight"]
    for t in range(seq_len):
        h = sd[prefix + "token_emb.emb.weight"][out[:, -1:]]
        h = h + pos[t:t + 1] * (h.shape[-1] ** -0.5)
        h = xt_decoder_layers(sd, prefix, h, context, context_mask, None, None, depth, heads, cache)
        logits = F.linear(h[:, -1], sd[prefix + "to_logits.weight"], sd.get(prefix + "to_logits.bias"))
        tok = sample_tokens(logits, None if noise is None else noise[t], temperature, k)
        out = torch.cat([out, tok.view(B, 1)], dim=1)
    return out[:, 1:]


def listener_generator_forward(sd, v_speaker, v_listener, mask, speaker_ids=None, listener_ids=None):
    """ListenerGenerator.forward(v_speaker[B,T,824], v_listener[B,T,56], mask, speaker_ids, listener_ids)
    -> (loss, pred_cont_seq [B,T-1,56]).  code/seq2seq.py:220-278 with Transformer.forward :46-67.
    x_engine.evaluate_epoch calls it with both ids None (code/x_engine.py:76); x_engine.train_epoch with
    speaker_ids=None and listener_ids given (code/x_engine.py:24): fc_listener(relu(listener_embeddings[id])) is put in
    front of the encoder output, the context mask gets a leading True and the targets a leading -100 (:50-57), and the
    extra first logit row is dropped (:64-65).  speaker_ids puts fc_speaker(relu(speaker_embeddings[id])) in front of
    the encoder INPUT (:238-243)."""
    B = v_speaker.shape[0]
    x_speaker = legacy_speaker_features(sd, v_speaker, mask)
    _, z_l = forward_vq(sd, v_speaker, v_listener, mask, with_speaker=False)
    cmask = mask
    one = torch.ones(B, 1, dtype=torch.bool)
    if speaker_ids is not None:
        sid = _lin(F.relu(sd["speaker_embeddings.weight"][speaker_ids]), sd, "fc_speaker")
        x_speaker = torch.cat([sid.unsqueeze(1), x_speaker], dim=1)
        cmask = torch.cat([one, cmask], dim=1)
    enc = xt_encoder(sd, "generator.encoder.", x_speaker, cmask, causal=False, depth=6, heads=8)
    tgt = z_l
    if listener_ids is not None:
        lid = _lin(F.relu(sd["listener_embeddings.weight"][listener_ids]), sd, "fc_listener")
        enc = torch.cat([lid.unsqueeze(1), enc], dim=1)
        cmask = torch.cat([one, cmask], dim=1)
        tgt = torch.cat([torch.full_like(z_l[:, :1], -100), z_l], dim=1)
    inp, target = tgt[:, :-1], tgt[:, 1:]
    inp = torch.where(inp == -100, torch.zeros_like(inp), inp)
    logits = legacy_decoder_logits(sd, inp, enc, cmask)
    loss = F.cross_entropy(logits.permute(0, 2, 1), target, ignore_index=-100)
    if listener_ids is not None:
        logits = logits[:, 1:, :]
    pred_seq = logits.argmax(dim=-1)
    pred = vq_decode(sd, pred_seq, "listener_vq.")
    loss_cont = continuous_loss(pred, v_listener, mask)
    return loss + loss_cont, pred, {"x_speaker": x_speaker, "enc": enc, "logits": logits, "z_l": z_l}


def listener_generator_generate(sd, v_speaker, v_listener, mask, noise=None):
    """ListenerGenerator.generate (code/seq2seq.py:280-306): seq_len = T generated tokens from the ground-truth
    first listener code -> (z_pred [B,T], z_listener [B,T])."""
    x_speaker = legacy_speaker_features(sd, v_speaker, mask)
    _, z_l = forward_vq(sd, v_speaker, v_listener, mask, with_speaker=False)
    enc = xt_encoder(sd, "generator.encoder.", x_speaker, mask, causal=False, depth=6, heads=8)
    z_pred = legacy_generate(sd, z_l[:, 0], z_l.shape[1], enc, mask, noise)
    return z_pred, z_l


# ----------------------------------------------------------------------------
# SLM pre-training forward (code/seq2seq_pretrain.py:58-323) -- SURVEY.md section 8(f2); x-transformers half
# PARITY UNPINNED like the stages above, pinned by the same self-consistency properties
# ----------------------------------------------------------------------------

def slm_random_masks(mask, mask_ratio=0.15, generator=None):
    """SLM.random_masking_unstructured (code/seq2seq_pretrain.py:170-183): per clip, int(len*ratio) distinct frames
    among the valid ones.  True = masked (input zeroed, token predicted)."""
    B, T = mask.shape
    out = torch.zeros(B, T, dtype=torch.bool)
    for i in range(B):
        n = int(mask[i].sum())
        idx = torch.randperm(n, generator=generator)[:int(n * mask_ratio)]
        out[i, :n][idx] = True
    return out


def slm_forward_encoder(sd, v_speaker, v_listener, mask, mask_speaker, mask_listener):
    """SLM.forward_encoder (code/seq2seq_pretrain.py:200-221) with the two random masks injected.  The encoders
    get ``mask=mask`` only (no attn_mask): bidirectional attention with key padding."""
    vs = v_speaker + sd["patch_embed_s"]
    vl = v_listener + sd["patch_embed_l"]
    vs = torch.where(mask_speaker[..., None], torch.zeros_like(vs), vs)
    vl = torch.where(mask_listener[..., None], torch.zeros_like(vl), vl)
    x_s = xt_encoder(sd, "encoder_s.", vs, mask, causal=False)
    x_l = xt_encoder(sd, "encoder_l.", vl, mask, causal=False)
    x_joint = xt_encoder(sd, "encoder_joint.", torch.cat([x_s, x_l], dim=1), torch.cat([mask, mask], dim=-1),
                         causal=False)
    x_l = xt_encoder(sd, "encoder_joint.", x_l, mask, causal=False)
    x_s = xt_encoder(sd, "encoder_joint.", x_s, mask, causal=False)
    ln = lambda x, n: F.layer_norm(x, (x.shape[-1],), sd[n + ".weight"], sd[n + ".bias"], LN_EPS)
    return ln(x_s, "norm_s"), ln(x_l, "norm_l"), ln(x_joint, "norm")


def slm_contrastive(s_rep, l_rep, mask):
    """SLM.forward_contrastive, single direction (code/seq2seq_pretrain.py:270-289)."""
    lens = mask.sum(1)
    s = torch.stack([s_rep[i, :lens[i]].mean(0) for i in range(len(lens))])
    l = torch.stack([l_rep[i, :lens[i]].mean(0) for i in range(len(lens))])
    s, l = F.normalize(s, dim=-1), F.normalize(l, dim=-1)
    total = s @ l.t() / 0.05
    nce = -torch.mean(torch.diag(F.log_softmax(total, dim=0)))
    c_acc = (F.softmax(total, dim=0).argmax(0) == torch.arange(total.shape[0])).sum() / total.shape[0]
    return nce, c_acc


def slm_decoder_tf(sd, z, context, context_mask, prefix="decoder_joint.net.", depth=4, heads=12):
    """decoder_joint(z, context, context_mask, return_outputs=True) of SLM: AutoregressiveWrapper (mask_prob 0)
    around a TransformerWrapper WITH absolute positional embedding (code/seq2seq_pretrain.py:131,160-165)."""
    inp, target = z[:, :-1], z[:, 1:]
    inp = torch.where(inp == -100, torch.zeros_like(inp), inp)
    n = inp.shape[1]
    h = sd[prefix + "token_emb.emb.weight"][inp]
    h = h + sd[prefix + "pos_emb.emb.weight"][:n] * (h.shape[-1] ** -0.5)
    causal = ~torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
    h = xt_decoder_layers(sd, prefix, h, context, context_mask, causal, None, depth, heads)
    logits = F.linear(h, sd[prefix + "to_logits.weight"], sd.get(prefix + "to_logits.bias"))
    loss = F.cross_entropy(logits.permute(0, 2, 1), target, ignore_index=-100)
    return loss, logits


def slm_forward(sd, v_speaker, v_listener, v_audio, mask, mask_speaker, mask_listener, return_aux=False):
    """SLM.forward (code/seq2seq_pretrain.py:300-323) -> (total_loss, dict, None)."""
    z_s, z_l = forward_vq(sd, v_speaker, v_listener, mask)
    x_s, x_l, x_joint = slm_forward_encoder(sd, v_speaker, v_listener, mask, mask_speaker, mask_listener)
    nce, c_acc = slm_contrastive(x_s, x_l, mask)
    T = x_s.shape[1]
    xj_s, xj_l = x_joint[:, :T], x_joint[:, T:]
    z_s = torch.where(mask_speaker, z_s, torch.full_like(z_s, -100))
    z_l = torch.where(mask_listener, z_l, torch.full_like(z_l, -100))
    ctx_s = torch.cat([xj_s + sd["patch_embed_dec_s"], v_audio], dim=-1)
    ctx_l = torch.cat([xj_l + sd["patch_embed_dec_l"], v_audio], dim=-1)
    l_ce_s, px_s = slm_decoder_tf(sd, z_s, ctx_l, mask)
    l_ce_l, px_l = slm_decoder_tf(sd, z_l, ctx_s, mask)
    pred_s = vq_decode(sd, px_s.argmax(-1), "speaker_vq.")
    pred_l = vq_decode(sd, px_l.argmax(-1), "listener_vq.")
    l_cont_s = continuous_loss(pred_s, v_speaker, mask_speaker)
    l_cont_l = continuous_loss(pred_l, v_listener, mask_listener)
    total = l_ce_s + l_ce_l + l_cont_s + l_cont_l + nce
    d = {"l_ce_s": l_ce_s, "l_ce_l": l_ce_l, "l_cont_s": l_cont_s, "l_cont_l": l_cont_l, "nce": nce, "c_acc": c_acc}
    if return_aux:
        return total, d, None, {"x_s": x_s, "x_l": x_l, "x_joint": x_joint, "px_s": px_s, "px_l": px_l,
                                "z_s": z_s, "z_l": z_l, "pred_s": pred_s, "pred_l": pred_l}
    return total, d, None
