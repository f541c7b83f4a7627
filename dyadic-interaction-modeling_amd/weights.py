"""State-dict surface of the DIM-Listener hot path + deterministic synthetic weights.

The key names and shapes are the ones a real reference checkpoint carries, so that
``best_vico_causal.pt`` / ``model.pth.tar`` files drop in unchanged:

* VQ-VAE keys: reference ``code/models/stage1_BIWI.py:254-393`` (module tree) and
  ``code/models/lib/base_models.py`` (Norm/Residual/Attention/MLP nesting);
  verified key-for-key against the imported reference by
  ``tests/golden/make_golden.py``.
* x-transformers 1.30.16 keys (``code/requirements.txt:99``; ctor sites
  ``code/seq2seq_pretrain.py:388-418``): restated from the library's module layout
   -- the library itself is not available here.
"""
from collections import OrderedDict
from dataclasses import dataclass
import math

import numpy as np
import torch

from . import prng


@dataclass(frozen=True)
class VQDims:
    in_dim: int = 56
    hidden: int = 384
    layers: int = 6
    heads: int = 8
    inter: int = 1536
    n_embed: int = 512
    zdim: int = 128
    neg: float = 0.2
    pe_len: int = 5000

    @staticmethod
    def from_cfg(cfg):
        assert cfg.quant_factor == 0 and cfg.face_quan_num == 1 and not cfg.INaffine, \
            "only the DIM-Listener VQ configuration (quant_factor 0, face_quan_num 1) is built"
        return VQDims(cfg.in_dim, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                      cfg.intermediate_size, cfg.n_embed, cfg.zquant_dim, float(cfg.neg))


@dataclass(frozen=True)
class S2SDims:
    """Hard-coded constructor values of SLMFT (reference code/seq2seq_pretrain.py:369-418)."""
    dim_in: int = 56
    dim: int = 384
    dim_a: int = 768
    enc_depth: int = 4
    dec_depth: int = 4
    heads: int = 12
    dim_head: int = 64
    num_tokens: int = 512
    max_seq_len: int = 2048
    ff_mult: int = 4
    mask_prob: float = 0.15

    @property
    def dec_dim(self):
        return self.dim + self.dim_a

    @property
    def inner(self):
        return self.heads * self.dim_head


# ----------------------------------------------------------------------------
# specs: ordered (name, shape, kind, fan_in)
# ----------------------------------------------------------------------------

def _vq_stack_spec(prefix, tname, d: VQDims):
    out = []
    H, I = d.hidden, d.inter
    for i in range(d.layers):
        a = "{}{}.net.{}.fn.".format(prefix, tname, 2 * i)
        out += [(a + "norm.weight", (H,), "ln_w", 0), (a + "norm.bias", (H,), "ln_b", 0),
                (a + "fn.to_qkv.weight", (3 * H, H), "w", H),
                (a + "fn.to_out.weight", (H, H), "w", H), (a + "fn.to_out.bias", (H,), "b", H)]
        m = "{}{}.net.{}.fn.".format(prefix, tname, 2 * i + 1)
        out += [(m + "norm.weight", (H,), "ln_w", 0), (m + "norm.bias", (H,), "ln_b", 0),
                (m + "fn.l1.weight", (I, H), "w", H), (m + "fn.l1.bias", (I,), "b", H),
                (m + "fn.l2.weight", (H, I), "w", I), (m + "fn.l2.bias", (H,), "b", I)]
    return out


def vq_spec(d: VQDims = VQDims(), prefix=""):
    H = d.hidden
    e = prefix + "encoder."
    s = [(e + "vertice_mapping.0.weight", (H, d.in_dim), "w", d.in_dim),
         (e + "vertice_mapping.0.bias", (H,), "b", d.in_dim),
         (e + "squasher.0.0.weight", (H, H, 5), "w", H * 5),
         (e + "squasher.0.0.bias", (H,), "b", H * 5)]
    s += _vq_stack_spec(e, "encoder_transformer", d)
    s += [(e + "encoder_pos_embedding.pe", (d.pe_len, 1, H), "pe", 0),
          (e + "encoder_linear_embedding.net.weight", (H, H), "w", H),
          (e + "encoder_linear_embedding.net.bias", (H,), "b", H),
          (e + "encoder_linear_embedding_post.net.weight", (d.zdim, H), "w", H),
          (e + "encoder_linear_embedding_post.net.bias", (d.zdim,), "b", H)]
    c = prefix + "decoder."
    s += [(c + "expander.0.0.weight", (H, H, 5), "w", H * 5),
          (c + "expander.0.0.bias", (H,), "b", H * 5)]
    s += _vq_stack_spec(c, "decoder_transformer", d)
    s += [(c + "decoder_pos_embedding.pe", (d.pe_len, 1, H), "pe", 0),
          (c + "decoder_linear_embedding.net.weight", (H, H), "w", H),
          (c + "decoder_linear_embedding.net.bias", (H,), "b", H),
          (c + "decoder_linear_embedding_pre.net.weight", (H, d.zdim), "w", d.zdim),
          (c + "decoder_linear_embedding_pre.net.bias", (H,), "b", d.zdim),
          (c + "vertice_map_reverse.weight", (d.in_dim, H), "w", H)]
    s += [(prefix + "quantize.embedding.weight", (d.n_embed, d.zdim), "codebook", 0)]
    return s


@dataclass(frozen=True)
class LegacyDims:
    """Hard-coded constructor values of the legacy ListenerGenerator (reference code/seq2seq.py:183-198) and its
    speaker VQ-VAE (reference code/config_speaker_old.yaml:15-30, arch stage1_BIWI_speaker)."""
    spk_in_dim: int = 824
    spk_hidden: int = 768
    spk_layers: int = 6
    spk_heads: int = 8
    spk_inter: int = 1536
    spk_face_quan_num: int = 8
    zdim: int = 128
    n_embed: int = 512
    dim: int = 512
    depth: int = 6
    heads: int = 8
    dim_head: int = 64
    num_tokens: int = 512
    max_seq_len: int = 1024
    ff_mult: int = 4

    @property
    def dim_in(self):
        return self.spk_face_quan_num * self.zdim

    @property
    def inner(self):
        return self.heads * self.dim_head


def legacy_speaker_vq_spec(d: LegacyDims = LegacyDims(), prefix="speaker_vq."):
    """Encoder + codebook of VQSpeakerAutoEncoder (reference code/models/stage1_BIWI.py:140-162); the two
    decoders (decoder_v / decoder_a) are not on the ListenerGenerator path and are accepted but ignored."""
    H = d.spk_hidden
    e = prefix + "encoder."
    vd = VQDims(in_dim=d.spk_in_dim, hidden=H, layers=d.spk_layers, heads=d.spk_heads, inter=d.spk_inter,
                n_embed=d.n_embed, zdim=d.zdim)
    s = [(e + "vertice_mapping.0.weight", (H, d.spk_in_dim), "w", d.spk_in_dim),
         (e + "vertice_mapping.0.bias", (H,), "b", d.spk_in_dim),
         (e + "squasher.0.0.weight", (H, H, 5), "w", H * 5),
         (e + "squasher.0.0.bias", (H,), "b", H * 5)]
    s += _vq_stack_spec(e, "encoder_transformer", vd)
    s += [(e + "encoder_pos_embedding.pe", (vd.pe_len, 1, H), "pe", 0),
          (e + "encoder_linear_embedding.net.weight", (H, H), "w", H),
          (e + "encoder_linear_embedding.net.bias", (H,), "b", H),
          (e + "encoder_linear_embedding_post.net.weight", (d.dim_in, H), "w", H),
          (e + "encoder_linear_embedding_post.net.bias", (d.dim_in,), "b", H)]
    s += [(prefix + "quantize.embedding.weight", (d.n_embed, d.zdim), "codebook", 0)]
    return s


def legacy_generator_spec(d: LegacyDims = LegacyDims(), prefix="generator."):
    """seq2seq.Transformer (reference code/seq2seq.py:13-52): x-tf encoder (dim_in 1024 -> 512, depth 6, heads 8)
    and cross-attending decoder WITH absolute positional embedding."""
    e = prefix + "encoder."
    s = [(e + "project_in.weight", (d.dim, d.dim_in), "w", d.dim_in),
         (e + "pos_emb.emb.weight", (d.max_seq_len, d.dim), "pos_emb", 0)]
    for i in range(d.depth):
        s += _xt_attn(e, 2 * i, d.dim, d.inner)
        s += _xt_ff(e, 2 * i + 1, d.dim, d.ff_mult)
    s += [(e + "attn_layers.final_norm.weight", (d.dim,), "ln_w", 0),
          (e + "project_out.weight", (d.dim, d.dim), "w", d.dim)]
    c = prefix + "decoder.net."
    s += [(c + "token_emb.emb.weight", (d.num_tokens, d.dim), "tok_emb", 0),
          (c + "pos_emb.emb.weight", (d.max_seq_len, d.dim), "pos_emb", 0)]
    for i in range(d.depth):
        s += _xt_attn(c, 3 * i, d.dim, d.inner)
        s += _xt_attn(c, 3 * i + 1, d.dim, d.inner)
        s += _xt_ff(c, 3 * i + 2, d.dim, d.ff_mult)
    s += [(c + "attn_layers.final_norm.weight", (d.dim,), "ln_w", 0),
          (c + "to_logits.weight", (d.num_tokens, d.dim), "w", d.dim)]
    return s


def listener_generator_spec(vq: VQDims = VQDims(), d: LegacyDims = LegacyDims()):
    """Every tensor of ListenerGenerator().state_dict() that the hot path uses plus the id-embedding layers
    (reference code/seq2seq.py:139-202); speaker_vq decoders are omitted (not built)."""
    s = legacy_speaker_vq_spec(d, "speaker_vq.") + vq_spec(vq, "listener_vq.") + legacy_generator_spec(d, "generator.")
    s += [("speaker_embeddings.weight", (100, 256), "tok_emb", 0), ("listener_embeddings.weight", (100, 256), "tok_emb", 0),
          ("fc_speaker.weight", (1024, 256), "w", 256), ("fc_speaker.bias", (1024,), "b", 256),
          ("fc_listener.weight", (512, 256), "w", 256), ("fc_listener.bias", (512,), "b", 256)]
    return s


def _xt_attn(prefix, li, dim, inner):
    p = "{}attn_layers.layers.{}.".format(prefix, li)
    return [(p + "0.0.weight", (dim,), "ln_w", 0),
            (p + "1.to_q.weight", (inner, dim), "w", dim),
            (p + "1.to_k.weight", (inner, dim), "w", dim),
            (p + "1.to_v.weight", (inner, dim), "w", dim),
            (p + "1.to_out.weight", (dim, inner), "w", inner)]


def _xt_ff(prefix, li, dim, mult):
    p = "{}attn_layers.layers.{}.".format(prefix, li)
    I = dim * mult
    return [(p + "0.0.weight", (dim,), "ln_w", 0),
            (p + "1.ff.0.0.weight", (I, dim), "w", dim), (p + "1.ff.0.0.bias", (I,), "b", dim),
            (p + "1.ff.2.weight", (dim, I), "w", I), (p + "1.ff.2.bias", (dim,), "b", I)]


def xt_encoder_spec(prefix, dim_in, d: S2SDims = S2SDims()):
    s = [(prefix + "project_in.weight", (d.dim, dim_in), "w", dim_in),
         (prefix + "pos_emb.emb.weight", (d.max_seq_len, d.dim), "pos_emb", 0)]
    for i in range(d.enc_depth):
        s += _xt_attn(prefix, 2 * i, d.dim, d.inner)
        s += _xt_ff(prefix, 2 * i + 1, d.dim, d.ff_mult)
    s += [(prefix + "attn_layers.final_norm.weight", (d.dim,), "ln_w", 0),
          (prefix + "project_out.weight", (d.dim, d.dim), "w", d.dim)]
    return s


def xt_decoder_spec(prefix="decoder_joint.net.", d: S2SDims = S2SDims()):
    D = d.dec_dim
    s = [(prefix + "token_emb.emb.weight", (d.num_tokens, D), "tok_emb", 0)]
    for i in range(d.dec_depth):
        s += _xt_attn(prefix, 3 * i, D, d.inner)
        s += _xt_attn(prefix, 3 * i + 1, D, d.inner)
        s += _xt_ff(prefix, 3 * i + 2, D, d.ff_mult)
    s += [(prefix + "attn_layers.final_norm.weight", (D,), "ln_w", 0),
          (prefix + "to_logits.weight", (d.num_tokens, D), "w", D)]
    return s


def slmft_spec(vq: VQDims = VQDims(), d: S2SDims = S2SDims()):
    """Every tensor of ``SLMFT().state_dict()`` in registration order
    (reference code/seq2seq_pretrain.py:348-418)."""
    s = vq_spec(vq, "speaker_vq.") + vq_spec(vq, "listener_vq.")
    s += xt_encoder_spec("encoder_s.", d.dim_in, d)
    s += xt_encoder_spec("encoder_l.", d.dim_in, d)
    s += xt_encoder_spec("encoder_joint.", d.dim, d)
    s += [("patch_embed_s", (1, 1, d.dim_in), "patch", 0), ("patch_embed_l", (1, 1, d.dim_in), "patch", 0),
          ("patch_embed_dec_s", (1, 1, d.dim), "patch", 0), ("patch_embed_dec_l", (1, 1, d.dim), "patch", 0)]
    for n in ("norm_s", "norm_l", "norm"):
        s += [(n + ".weight", (d.dim,), "ln_w", 0), (n + ".bias", (d.dim,), "ln_b", 0)]
    s += xt_decoder_spec("decoder_joint.net.", d)
    return s


def slm_spec(vq: VQDims = VQDims(), d: S2SDims = S2SDims()):
    """Every tensor of ``SLM().state_dict()`` (reference code/seq2seq_pretrain.py:58-165): the SLMFT set plus the
    decoder's absolute positional embedding (``use_abs_pos_emb`` defaults to True there, :131)."""
    return slmft_spec(vq, d) + [("decoder_joint.net.pos_emb.emb.weight", (d.max_seq_len, d.dec_dim), "pos_emb", 0)]


# ----------------------------------------------------------------------------
# synthetic initialisation
# ----------------------------------------------------------------------------

def sinusoid_pe(max_len: int, dim: int) -> torch.Tensor:
    """The `pe` buffer of the VQ-VAE PositionalEncoding, shape [max_len,1,dim]
    (reference code/models/lib/base_models.py:258-269), computed with the same
    torch float32 op sequence so that it is bit-identical to the reference buffer."""
    pe = torch.zeros(max_len, dim)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2).float() * (-math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).transpose(0, 1).contiguous()


def synth_tensor(seed, name, shape, kind, fan_in, plain=False):
    """One synthetic tensor.  ``plain=True`` gives the SURVEY Appendix-C flavour
    (zero biases, unit norm gains); the default exercises every bias / gain path."""
    if kind == "w":
        b = 1.0 / math.sqrt(fan_in)
        return torch.from_numpy(prng.uniform(seed, name, shape, -b, b))
    if kind == "b":
        if plain:
            return torch.zeros(shape)
        b = 1.0 / math.sqrt(fan_in)
        return torch.from_numpy(prng.uniform(seed, name, shape, -b, b))
    if kind == "ln_w":
        return torch.ones(shape) if plain else torch.from_numpy(prng.uniform(seed, name, shape, 0.8, 1.2))
    if kind == "ln_b":
        return torch.zeros(shape) if plain else torch.from_numpy(prng.uniform(seed, name, shape, -0.1, 0.1))
    if kind == "codebook":
        return torch.from_numpy(prng.uniform(seed, name, shape, -1.0, 1.0))
    if kind == "tok_emb":
        return torch.from_numpy(prng.normal(seed, name, shape))
    if kind == "pos_emb":
        return torch.from_numpy(prng.uniform(seed, name, shape, -1.0, 1.0))
    if kind == "patch":
        return torch.from_numpy(prng.uniform(seed, name, shape, -0.02, 0.02))
    if kind == "pe":
        return sinusoid_pe(shape[0], shape[2])
    raise ValueError(kind)


def synth_state_dict(spec, seed=20260928, plain=False, strip_prefix=""):
    """OrderedDict name -> float32 tensor for every entry of ``spec``.  The stream of
    a tensor is keyed by its *full* name, so ``listener_vq.`` and ``speaker_vq.`` get
    different weights; ``strip_prefix`` removes a prefix from the returned keys only."""
    sd = OrderedDict()
    pe_cache = {}
    for name, shape, kind, fan_in in spec:
        if kind == "pe":
            k = tuple(shape)
            if k not in pe_cache:
                pe_cache[k] = sinusoid_pe(shape[0], shape[2])
            t = pe_cache[k]
        else:
            t = synth_tensor(seed, name, shape, kind, fan_in, plain)
        key = name[len(strip_prefix):] if strip_prefix and name.startswith(strip_prefix) else name
        sd[key] = t
    return sd


def synth_clips(seed, B, T, vico_like=False):
    """Synthetic dyad clips: speaker motion, listener motion,
    speaker audio features, all float32."""
    v_s = torch.from_numpy(prng.normal(seed, "clip.v_speaker", (B, T, 56)))
    if vico_like:      # ViCo loader replaces speaker video by ones (code/dataset/data_loader.py:147)
        v_s = torch.ones(B, T, 56)
    v_l = torch.from_numpy(prng.normal(seed, "clip.v_listener", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(seed, "clip.v_audio", (B, T, 768)))
    return v_s, v_l, v_a
