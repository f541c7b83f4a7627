"""Drop-in ``SLMFT`` (DIM-Listener): same constructor surface, parameter names, method names and
``forward`` signature/return as the reference ``code/seq2seq_pretrain.py:325-514``, computing on the
HIP library (``include/dimx.h``) through ``dimx.engine.Engine``.

Differences that are deliberate and documented:
  * the no-arg constructor still works, but since the reference's checkpoint files do not exist here the
    VQ-VAEs start from deterministic synthetic weights; ``vq_speaker_ckpt`` / ``vq_listener_ckpt`` accept
    the reference's ``model.pth.tar`` files (``{'state_dict': ...}``) when they do exist;
  * randomness is injectable: ``noise`` (Exp(1) sampling noise, [T-1,B,512]), ``kv_mask`` (the
    AutoregressiveWrapper key mask, [B,T-1] bool), ``greedy`` and ``seed``; the defaults draw fresh
    randomness like the reference;
  * ``forward_vq`` encodes each stream once, batched over ragged clips on the GPU (the reference encodes
    batch-1 clips in a Python loop, twice, code/seq2seq_pretrain.py:497-500); results are identical.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import config as _config
from . import lib as L
from . import weights as W
from .models import _EngineOwner, build_param_tree, get_model


def _masked_pairwise_loss(pred, target, m):
    """mean_{valid} ||p[6:] - t[6:] + eps|| + mean_{valid} ||p[:6] - t[:6] + eps||  (F.pairwise_distance, eps 1e-6),
    without a device -> host synchronisation; an all-False mask gives NaN like the mean of an empty selection."""
    m = m.bool()
    d = F.pairwise_distance(pred[..., 6:], target[..., 6:]) + F.pairwise_distance(pred[..., 0:6], target[..., 0:6])
    zero = torch.zeros((), dtype=d.dtype, device=d.device)
    return torch.where(m, d, zero).sum() / m.sum()


def mark_prefix(mask):
    """Declare that every row of ``mask`` is True on a prefix [0, len) and False after it (what the engine protocol builds
    from ``src_len``, reference code/x_engine_pt.py:203-206).  compact_by_mask then has nothing to move and skips its
    stable argsort + gather over [B,T,56]; the claim is the caller's (a Python attribute on the tensor, lost by any op
    that creates a new tensor -- the safe direction)."""
    mask.dimx_prefix = True
    return mask


def compact_by_mask(x, mask):
    """Left-align the valid frames of every clip (``v[i][mask[i]]`` of the reference, batched).
    Returns (x_compact, lens int32).  A prefix mask (the engine protocol) keeps its valid frames where they are; frames past
    a clip's length are zero unless the mask was declared a prefix mask (mark_prefix) -- the engine never reads them."""
    lens = mask.sum(1).to(torch.int32)
    if getattr(mask, "dimx_prefix", False):
        return x, lens
    T = mask.shape[1]
    prefix = torch.arange(T, device=mask.device)[None, :] < lens[:, None]
    if not x.is_cuda and torch.equal(prefix, mask):
        return x, lens
    # On the GPU the same arithmetic runs for every mask (for a prefix mask the stable argsort is the identity): testing
    # `mask == prefix` on the host would synchronise with the device at the head of every forward call, and the GPU then
    # idles (1.3 ms per 256-clip batch in the round-2 trace) while the host queues the first kernels of the next batch.
    order = torch.argsort((~mask).to(torch.int8), dim=1, stable=True)
    xc = torch.gather(x, 1, order[..., None].expand(-1, -1, x.shape[-1]))
    return torch.where(prefix[..., None], xc, torch.zeros((), dtype=x.dtype, device=x.device)), lens


class SLMFT(_EngineOwner):
    def __init__(self, config_path=None, vq_speaker_ckpt=None, vq_listener_ckpt=None,
                 synthetic_seed=20260928, numeric_mode=L.MODE_PARITY_F32):
        super().__init__(numeric_mode)
        config_path = config_path or ("./config.yaml" if os.path.isfile("./config.yaml") else _config.DEFAULT_CONFIG)
        cfg_s = _config.load_cfg_from_cfg_file(config_path)
        cfg_l = _config.load_cfg_from_cfg_file(config_path)
        self.speaker_vq = get_model(cfg_s, synthetic_seed=synthetic_seed, weight_prefix="speaker_vq.", which=0,
                                    numeric_mode=numeric_mode)
        self.listener_vq = get_model(cfg_l, synthetic_seed=synthetic_seed, weight_prefix="listener_vq.", which=1,
                                     numeric_mode=numeric_mode)
        for m, ck in ((self.speaker_vq, vq_speaker_ckpt), (self.listener_vq, vq_listener_ckpt)):
            if ck is not None:
                m.load_state_dict(torch.load(ck, map_location="cpu")["state_dict"])
            m.eval()
        self.speaker_face_quan_num = cfg_s.face_quan_num
        self.speaker_zquant_dim = cfg_s.zquant_dim
        self.s2s = W.S2SDims()
        spec = [e for e in W.slmft_spec(self.speaker_vq.dims, self.s2s)
                if not (e[0].startswith("speaker_vq.") or e[0].startswith("listener_vq."))]
        build_param_tree(self, spec, W.synth_state_dict(spec, synthetic_seed))
        self.mask_prob = self.s2s.mask_prob

    # ------------------------------------------------------------------ engine plumbing
    def _engine_state_dict(self):
        return self.state_dict()

    def _mask8(self, mask):
        return mask.to(torch.uint8).contiguous()

    # ------------------------------------------------------------------ reference sub-APIs
    @torch.no_grad()
    def forward_vq(self, v_speaker, v_listener, mask, with_speaker=True):
        """reference :480-494 -> (z_speaker [B,T] padded with 0, z_listener [B,T] padded with -100), int64."""
        eng = self.engine(v_speaker.device)
        xl, lens = compact_by_mask(v_listener, mask)
        z_l = eng.vq_encode(1, xl, lens, pe_mode=0, pad_value=-100).long()
        z_s = None
        if with_speaker:
            xs, _ = compact_by_mask(v_speaker, mask)
            z_s = eng.vq_encode(0, xs, lens, pe_mode=0, pad_value=0).long()
        return z_s, z_l

    @torch.no_grad()
    def forward_encoder(self, v_speaker, mask):
        """reference :431-442 -> x_s [B,T,384] (rows of padded frames are unspecified)."""
        return self.engine(v_speaker.device).encode_speaker(v_speaker, self._mask8(mask.bool()))

    def _build_context(self, eng, x_s, v_speaker, x_a, m8, for_generate, n_samples=1):
        # x_s given (the reference's call shape): context straight from it; x_s None + v_speaker: the fused stage that
        # keeps the encoder output inside the workspace (what forward() uses)
        if x_s is not None:
            eng.set_context(x_s, x_a, which_patch=0, for_generate=for_generate, n_samples=n_samples)
        else:
            assert v_speaker is not None, "forward_decoder needs x_s (reference call) or v_speaker= (fused path)"
            eng.encode_ctx(v_speaker, x_a, m8, for_generate, n_samples=n_samples)

    @staticmethod
    def _user_seed(seed):
        """seed -> non-zero generator key (the C-ABI reserves 0 for 'greedy when no noise is given')."""
        if seed is None:
            return int(torch.randint(1, 2 ** 62, (1,)).item())
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        return seed if seed != 0 else 0x9E3779B97F4A7C15

    @torch.no_grad()
    def forward_decoder(self, x_s, z_l, x_a, mask, mode, v_speaker=None, noise=None, kv_mask=None, greedy=False,
                        seed=None, temperature=1.0, n_samples=1):
        """reference :444-452, same positional call: ``forward_decoder(x_s, z_l, x_a, mask, mode)`` with the ``x_s``
        that ``forward_encoder`` returned.  ``forward()`` passes ``x_s=None, v_speaker=...`` instead, which keeps the
        encoder output inside the engine workspace (no round trip through a tensor)."""
        dev = (x_s if x_s is not None else v_speaker).device
        eng = self.engine(dev)
        m8 = self._mask8(mask.bool())
        B, T = z_l.shape
        if mode == "train":
            self._build_context(eng, x_s, v_speaker, x_a, m8, False)
            if kv_mask is None:
                kv_mask = self.draw_kv_mask(B, T, dev)
            elif kv_mask is False:
                kv_mask = None
            logits, row_loss, _ = eng.decode_tf(z_l, m8, self._mask8(kv_mask) if kv_mask is not None else None)
            n_valid = (z_l[:, 1:] != -100).sum().clamp(min=1)
            return row_loss.sum() / n_valid, logits
        self._build_context(eng, x_s, v_speaker, x_a, m8, True, n_samples=n_samples)
        if greedy:
            temperature, seed_v = 0.0, 0
        else:
            seed_v = 0 if noise is not None else self._user_seed(seed)
        tokens = eng.generate(z_l[:, 0], m8, T, temperature, 52, noise, seed_v, n_samples=n_samples)
        return 0.0, tokens.long()

    def draw_kv_mask(self, B, T, device, generator=None):
        """AutoregressiveWrapper(mask_prob=0.15) key mask (reference ctor :419): keep-mask [B,T-1]."""
        n = T - 1
        rand = torch.randn(B, n, device=device, generator=generator)
        rand[:, 0] = -torch.finfo(rand.dtype).max
        num_mask = min(int(T * self.mask_prob), T - 1)
        idx = rand.topk(num_mask, dim=-1).indices
        return ~torch.zeros(B, n, device=device).scatter(1, idx, 1.0).bool()

    @torch.no_grad()
    def forward_vq_decoder(self, logits_l, mode="train", batch_row_offset=0, rows_per_clip=1):
        """reference :454-464: argmax (train) / tokens (val) -> codebook lookup -> listener_vq.decode."""
        pred_seq_l = torch.argmax(logits_l, dim=-1) if mode == "train" else logits_l
        return self.engine(pred_seq_l.device).vq_decode(1, pred_seq_l, batch_row_offset, rows_per_clip)

    def forward_continuous_loss(self, pred, target, mask):
        """reference :466-478: mean pairwise distance over the valid frames, expression part + pose part.  Written as
        masked sums / count instead of boolean indexing: ``x[m]`` synchronises with the device (nonzero) and left the GPU
        idle for ~0.6 ms per call inside the timed forward."""
        return _masked_pairwise_loss(pred, target[:, 1:, :], mask[:, 1:])

    # ------------------------------------------------------------------ training
    def dimx_trainable_parameters(self):
        from . import train as T
        return T.trainable_parameters(self)

    def _wants_grad(self, mode):
        return (mode == "train" and self.training and torch.is_grad_enabled()
                and any(p.requires_grad for p in self.parameters()))

    def train(self, mode=True):
        """nn.Module.train, with the reference's frozen parts kept in eval (code/seq2seq_pretrain.py:348-366)."""
        super().train(mode)
        self.speaker_vq.eval()
        self.listener_vq.eval()
        return self

    def _forward_autograd(self, v_speaker, v_listener, v_audio, mask, kv_mask=None, z_l=None, return_tokens=False):
        """mode='train' with a graph: differentiable cross entropy (dimx.train.slmft_loss on this module's own
        parameters); listener codes and decoded motion come from the HIP engine without a graph."""
        from . import train as T
        mask = mask.bool()
        B, Tn = mask.shape
        on_gpu = v_speaker.is_cuda
        if z_l is None:
            with torch.no_grad():
                _, z_l = self.forward_vq(v_speaker, v_listener, mask, with_speaker=False)
        if kv_mask is None:
            kv_mask = self.draw_kv_mask(B, Tn, v_speaker.device)
        elif kv_mask is False:
            kv_mask = None
        P = dict(self.named_parameters())
        l_ce_l, logits = T.slmft_loss(P, self.s2s, v_speaker.float(), v_audio.float(), mask, z_l.long(),
                                      kv_mask.bool() if kv_mask is not None else None)
        pred, l_cont_l = None, torch.zeros((), device=v_speaker.device)
        if on_gpu:     # the continuous loss has no gradient path in the reference either (argmax -> one-hot, :454-464)
            with torch.no_grad():
                pred = self.forward_vq_decoder(logits.detach(), mode="train")
                l_cont_l = self.forward_continuous_loss(pred, v_listener, mask)
        total_loss = l_ce_l + l_cont_l
        d = {"l_ce_s": 0, "l_ce_l": l_ce_l.detach(), "l_cont_s": 0, "l_cont_l": l_cont_l, "nce": 0, "c_acc": 0}
        if return_tokens:
            return total_loss, d, pred, logits.detach().argmax(-1)
        return total_loss, d, pred

    # ------------------------------------------------------------------ forward
    def forward(self, v_speaker, v_listener, v_audio, mask, mode="train", speaker_ids=None, listener_ids=None,
                noise=None, kv_mask=None, greedy=False, seed=None, temperature=1.0, batch_row_offset=0,
                return_tokens=False, n_samples=1, shard=None, z_l=None):
        """reference :496-514.  In training (``model.train()``, grad enabled, parameters requiring grad) the
        teacher-forced pass returns a loss with an autograd graph; everything else is the HIP inference path."""
        if self._wants_grad(mode):
            return self._forward_autograd(v_speaker, v_listener, v_audio, mask, kv_mask=kv_mask, z_l=z_l,
                                          return_tokens=return_tokens)
        with self.engine_pinned():     # one weights check per forward, not one per stage
            return self._forward_nograd(v_speaker, v_listener, v_audio, mask, mode=mode, noise=noise, kv_mask=kv_mask,
                                        greedy=greedy, seed=seed, temperature=temperature,
                                        batch_row_offset=batch_row_offset, return_tokens=return_tokens,
                                        n_samples=n_samples, shard=shard)

    @torch.no_grad()
    def _forward_nograd(self, v_speaker, v_listener, v_audio, mask, mode="train", speaker_ids=None, listener_ids=None,
                        noise=None, kv_mask=None, greedy=False, seed=None, temperature=1.0, batch_row_offset=0,
                        return_tokens=False, n_samples=1, shard=None):
        """reference :496-514 -> (total_loss, dict, pred_cont_seq_l [B,T-1,56]).

        ``n_samples`` S > 1 (mode 'val' only): S independent generations per clip in ONE pass -- what the
        reference's evaluation loop obtains from S separate forward calls (code/x_engine_pt.py:257) -- sharing the
        VQ encode, the encoder stack and the context K/V stream; pred is then [B,S,T-1,56], tokens [B,S,T-1]."""
        mask = mask.bool()
        S = int(n_samples)
        assert S == 1 or mode != "train", "n_samples applies to mode='val'"
        _, z_l = self.forward_vq(v_speaker, v_listener, mask, with_speaker=False)
        # shard = (first clip row, clips in the whole batch) when this call is one rank's slice of a batch: together
        # with batch_row_offset it makes the slice reproduce the same rows of a single-process call
        eng = self.engine(v_speaker.device)
        eng.set_shard(*(shard if shard is not None else (0, 0)))
        try:
            l_ce_l, px_l = self.forward_decoder(None, z_l, v_audio, mask, mode, v_speaker=v_speaker, noise=noise,
                                                kv_mask=kv_mask, greedy=greedy, seed=seed, temperature=temperature,
                                                n_samples=S)
        finally:
            eng.set_shard(0, 0)
        pred = self.forward_vq_decoder(px_l, mode=mode, batch_row_offset=batch_row_offset, rows_per_clip=S)
        if S > 1:
            B, T = mask.shape
            pred = pred.view(B, S, T - 1, -1)
            l_cont_l = torch.stack([self.forward_continuous_loss(pred[:, i], v_listener, mask) for i in range(S)]).mean()
        else:
            l_cont_l = self.forward_continuous_loss(pred, v_listener, mask)
        total_loss = l_ce_l + l_cont_l
        d = {"l_ce_s": 0, "l_ce_l": l_ce_l, "l_cont_s": 0, "l_cont_l": l_cont_l, "nce": 0, "c_acc": 0}
        if return_tokens:
            tokens = px_l if mode != "train" else torch.argmax(px_l, dim=-1)
            if S > 1:
                tokens = tokens.view(mask.shape[0], S, -1)
            return total_loss, d, pred, tokens
        return total_loss, d, pred


class SLM(_EngineOwner):
    """Drop-in ``SLM`` (the pre-training model, reference ``code/seq2seq_pretrain.py:58-323``), forward pass only:
    ``forward(v_speaker, v_listener, v_audio, mask, ...) -> (total_loss, dict, None)`` with the reference's six
    dict entries.  The random frame masks of ``random_masking_unstructured`` (:170-183) are injectable
    (``mask_speaker`` / ``mask_listener`` bool [B,T], True = masked); by default they are drawn like the reference.
    The InfoNCE term (:270-289) is a handful of [B,384] torch ops on the engine's encoder outputs.
    In training (``model.train()``, grad enabled, parameters requiring grad -- what ``x_engine_pt.train_epoch`` sets up for
    code/train_s2s_pretrain.py:41-64) the loss carries an autograd graph (``dimx.train.slm_loss``; the frozen VQ encoders
    run on the HIP engine); everything else is the HIP inference path."""
    engine_variant = "slm"

    def __init__(self, config_path=None, vq_speaker_ckpt=None, vq_listener_ckpt=None, synthetic_seed=20260928,
                 numeric_mode=L.MODE_PARITY_F32):
        super().__init__(numeric_mode)
        config_path = config_path or ("./config.yaml" if os.path.isfile("./config.yaml") else _config.DEFAULT_CONFIG)
        cfg = _config.load_cfg_from_cfg_file(config_path)
        self.vq_dims = W.VQDims.from_cfg(cfg)
        self.s2s = W.S2SDims()
        self.speaker_face_quan_num = cfg.face_quan_num
        self.speaker_zquant_dim = cfg.zquant_dim
        spec = W.slm_spec(self.vq_dims, self.s2s)
        build_param_tree(self, spec, W.synth_state_dict(spec, synthetic_seed))
        for pre, ck in (("speaker_vq.", vq_speaker_ckpt), ("listener_vq.", vq_listener_ckpt)):
            if ck is not None:
                sd = torch.load(ck, map_location="cpu")["state_dict"]
                own = self.state_dict()
                self.load_state_dict({pre + k.replace("module.", "", 1): v for k, v in sd.items()
                                      if pre + k.replace("module.", "", 1) in own}, strict=False)
        self.eval()

    def _engine_state_dict(self):
        return self.state_dict()

    @staticmethod
    def random_masking_unstructured(x, mask, mask_ratio, generator=None):
        """reference :170-183 -> bool [N,L], True = masked."""
        N, L_ = mask.shape
        out = torch.zeros(N, L_, dtype=torch.bool)
        lens = mask.sum(1).tolist()
        for i, n in enumerate(lens):
            idx = torch.randperm(int(n), generator=generator)[:int(n * mask_ratio)]
            out[i, :int(n)][idx] = True
        return out.to(mask.device)

    @torch.no_grad()
    def forward_vq(self, v_speaker, v_listener, mask):
        eng = self.engine(v_speaker.device)
        xl, lens = compact_by_mask(v_listener, mask)
        xs, _ = compact_by_mask(v_speaker, mask)
        z_l = eng.vq_encode(1, xl.contiguous(), lens, pe_mode=0, pad_value=-100).long()
        z_s = eng.vq_encode(0, xs.contiguous(), lens, pe_mode=0, pad_value=0).long()
        return z_s, z_l

    @torch.no_grad()
    def forward_encoder(self, v_speaker, v_listener, mask, mask_ratio=0.15, mask_speaker=None, mask_listener=None):
        if mask_speaker is None:
            mask_speaker = self.random_masking_unstructured(v_speaker, mask, mask_ratio)
        if mask_listener is None:
            mask_listener = self.random_masking_unstructured(v_listener, mask, mask_ratio)
        eng = self.engine(v_speaker.device)
        u8 = lambda m: m.to(torch.uint8).contiguous()
        x_s, x_l, x_joint = eng.slm_encode(v_speaker, v_listener, u8(mask), u8(mask_speaker), u8(mask_listener))
        return x_s, x_l, x_joint, mask_speaker, mask_listener

    @staticmethod
    def forward_contrastive(s_rep, l_rep, mask, bidirect_contrast=False):
        """reference :270-298."""
        valid = mask[..., None].to(s_rep.dtype)
        n = valid.sum(1)
        s = F.normalize((s_rep * valid).sum(1) / n, dim=-1)
        l_ = F.normalize((l_rep * valid).sum(1) / n, dim=-1)
        total = s @ l_.t() / 0.05
        ar = torch.arange(total.shape[0], device=total.device)

        def one(t):
            return (-torch.mean(torch.diag(F.log_softmax(t, dim=0))),
                    (F.softmax(t, dim=0).argmax(0) == ar).sum() / t.shape[0])
        nce, acc = one(total)
        if bidirect_contrast:
            n2, a2 = one(total.t())
            nce, acc = (nce + n2) / 2, (acc + a2) / 2
        return nce, acc

    @torch.no_grad()
    def _decode_tf(self, eng, x_joint, half, patch, z, v_audio, m8):
        T = z.shape[1]
        eng.set_context(x_joint[:, half * T:], v_audio, which_patch=patch, T=T)
        logits, row_loss, amax = eng.decode_tf(z, m8, None)
        n_valid = (z[:, 1:] != -100).sum().clamp(min=1)
        return row_loss.sum() / n_valid, logits, amax

    def forward_continuous_loss(self, pred, target, mask):
        return _masked_pairwise_loss(pred, target[:, 1:, :], mask[:, 1:])

    # ------------------------------------------------------------------ training
    def dimx_trainable_parameters(self):
        from . import train as T
        return T.slm_trainable_parameters(self)

    def train(self, mode=True):
        """nn.Module.train with both VQ-VAEs kept in eval (reference :96, :105)."""
        super().train(mode)
        self.speaker_vq.eval()
        self.listener_vq.eval()
        return self

    def _forward_autograd(self, v_speaker, v_listener, v_audio, mask, mask_ratio=0.15, mask_speaker=None, mask_listener=None,
                          z_s=None, z_l=None):
        from . import train as T
        mask = mask.bool()
        if z_s is None or z_l is None:
            z_s, z_l = self.forward_vq(v_speaker, v_listener, mask)
        if mask_speaker is None:
            mask_speaker = self.random_masking_unstructured(v_speaker, mask, mask_ratio)
        if mask_listener is None:
            mask_listener = self.random_masking_unstructured(v_listener, mask, mask_ratio)
        total, d = T.slm_loss(dict(self.named_parameters()), self.s2s, self.vq_dims, v_speaker.float(), v_listener.float(),
                              v_audio.float(), mask, mask_speaker.bool(), mask_listener.bool(), z_s.long(), z_l.long(),
                              self.speaker_vq.decoder.decoder_pos_embedding.pe, self.listener_vq.decoder.decoder_pos_embedding.pe)
        return total, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()}, None

    def forward(self, v_speaker, v_listener, v_audio, mask, speaker_ids=None, listener_ids=None, mode="train",
                mask_speaker=None, mask_listener=None, return_aux=False, z_s=None, z_l=None):
        """reference :300-323 -> (total_loss, d, None)."""
        if (self.training and torch.is_grad_enabled() and not return_aux
                and any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(v_speaker, v_listener, v_audio, mask, mask_speaker=mask_speaker,
                                          mask_listener=mask_listener, z_s=z_s, z_l=z_l)
        with torch.no_grad():
            return self._forward_nograd(v_speaker, v_listener, v_audio, mask, mask_speaker, mask_listener, return_aux)

    def _forward_nograd(self, v_speaker, v_listener, v_audio, mask, mask_speaker=None, mask_listener=None, return_aux=False):
        mask = mask.bool()
        eng = self.engine(v_speaker.device)
        z_s, z_l = self.forward_vq(v_speaker, v_listener, mask)
        x_s, x_l, x_joint, mask_speaker, mask_listener = self.forward_encoder(
            v_speaker, v_listener, mask, mask_speaker=mask_speaker, mask_listener=mask_listener)
        nce, c_acc = self.forward_contrastive(x_s, x_l, mask)
        z_s = torch.where(mask_speaker, z_s, torch.full_like(z_s, -100))
        z_l = torch.where(mask_listener, z_l, torch.full_like(z_l, -100))
        m8 = mask.to(torch.uint8).contiguous()
        # z_s is predicted from the listener half of x_joint, z_l from the speaker half (:227-228)
        l_ce_s, px_s, am_s = self._decode_tf(eng, x_joint, 1, 1, z_s, v_audio, m8)
        l_ce_l, px_l, am_l = self._decode_tf(eng, x_joint, 0, 0, z_l, v_audio, m8)
        pred_s = eng.vq_decode(0, am_s, 0)
        pred_l = eng.vq_decode(1, am_l, 0)
        l_cont_s = self.forward_continuous_loss(pred_s, v_speaker, mask_speaker)
        l_cont_l = self.forward_continuous_loss(pred_l, v_listener, mask_listener)
        total_loss = l_ce_s + l_ce_l + l_cont_s + l_cont_l + nce
        d = {"l_ce_s": l_ce_s, "l_ce_l": l_ce_l, "l_cont_s": l_cont_s, "l_cont_l": l_cont_l, "nce": nce, "c_acc": c_acc}
        if return_aux:
            return total_loss, d, None, {"x_s": x_s, "x_l": x_l, "x_joint": x_joint, "px_s": px_s, "px_l": px_l,
                                         "pred_s": pred_s, "pred_l": pred_l}
        return total_loss, d, None


class SpeakerSLMFT(nn.Module):
    """Import-compatibility placeholder for reference ``code/seq2seq_pretrain.py:516-757`` (``test_s2s_pretrain.py:7``
    imports it next to SLMFT).  The speaker-generation model depends on the EMOCA->FLAME converter and BIWI templates
    and is outside the DIM-Listener path (out of scope): constructing it fails loudly."""

    def __init__(self, *a, **kw):
        super().__init__()
        raise NotImplementedError("SpeakerSLMFT is not built: only the DIM-Listener path (SLMFT, SLM, the legacy "
                                  "ListenerGenerator and the VQ-VAEs) runs on the HIP library")
