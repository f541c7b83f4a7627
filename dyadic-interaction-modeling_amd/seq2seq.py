"""Drop-in ``ListenerGenerator`` (the legacy listener generator driven by ``x_engine.py``): constructor
surface, parameter names, ``forward`` / ``generate`` signatures and returns of the reference
``code/seq2seq.py:138-306``, computing on the HIP library (variant 1 of ``include/dimx.h``).

Path (reference lines in brackets):
  speaker VQ-VAE encode of every clip's valid frames, batch-1 [228] -> codebook vectors, zero pad [229] ->
  the raw ``.view`` re-reading of the channel-major tensor as [B,T,1024] [236-237] -> 6-layer bidirectional
  x-transformers encoder [54] -> 6-layer cross-attending decoder with absolute positional embedding, teacher
  forced [66] or sampled (top-k 52 of 512 logits, ``seq_len = T`` tokens) [300] -> listener VQ-VAE decode [263].

Deliberate, documented differences:
  * the no-arg constructor works without the reference's ``./runs/*/model.pth.tar`` files (deterministic
    synthetic weights); ``vq_speaker_ckpt`` / ``vq_listener_ckpt`` load them when they exist;
  * ``speaker_ids`` / ``listener_ids`` conditioning (train_epoch only, code/x_engine.py:23) exists on the training
    branch (``model.train()`` + grad enabled: ``dimx.train.legacy_loss`` on autograd, the VQ-VAE halves from the HIP
    engine); the no-grad evaluation path raises NotImplementedError for them, x_engine.evaluate_epoch never passes them;
  * sampling randomness is injectable (``noise`` [T,B,512] Exp(1) variates, ``seed``, ``greedy``);
  * the per-clip Python loops [227-233] are one batched ragged pass on the GPU; results are identical.
"""
import os

import torch
import torch.nn.functional as F

from . import config as _config
from . import lib as L
from . import weights as W
from .models import _EngineOwner, build_param_tree
from .seq2seq_pretrain import compact_by_mask


class ListenerGenerator(_EngineOwner):
    engine_variant = "legacy"

    def __init__(self, config_listener_pth=None, vq_speaker_ckpt=None, vq_listener_ckpt=None,
                 synthetic_seed=20260928, numeric_mode=L.MODE_PARITY_F32):
        super().__init__(numeric_mode)
        config_listener_pth = config_listener_pth or (
            "./config.yaml" if os.path.isfile("./config.yaml") else _config.DEFAULT_CONFIG)
        cfg_l = _config.load_cfg_from_cfg_file(config_listener_pth)
        self.dims = W.LegacyDims()
        self.vq_dims = W.VQDims.from_cfg(cfg_l)
        self.speaker_face_quan_num = self.dims.spk_face_quan_num      # config_speaker_old.yaml: face_quan_num 8
        self.speaker_zquant_dim = self.dims.zdim                      # zquant_dim 128
        spec = W.listener_generator_spec(self.vq_dims, self.dims)
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

    # ------------------------------------------------------------------ pieces
    def _prepare(self, v_speaker, v_listener, mask):
        """-> engine, compacted speaker/listener streams, uint8 mask whose popcount is each clip's length.
        The encoder / cross-attention key mask is the caller's mask (positions), the VQ-VAEs see the
        compacted valid frames (``v[i][mask[i]]``)."""
        mask = mask.bool()
        eng = self.engine(v_speaker.device)
        xs, lens = compact_by_mask(v_speaker, mask)
        xl, _ = compact_by_mask(v_listener, mask)
        return eng, xs.contiguous(), xl.contiguous(), lens, mask.to(torch.uint8).contiguous()

    @torch.no_grad()
    def speaker_features(self, v_speaker, mask):
        """x_speaker [B,T,1024] exactly as ListenerGenerator.forward builds it (reference :224-237)."""
        eng, xs, _, _, m8 = self._prepare(v_speaker, v_speaker[..., :1], mask)
        return eng.legacy_speaker_features(xs, m8)

    @torch.no_grad()
    def listener_codes(self, v_listener, mask):
        """z_listener [B,T] int64 padded with -100 (reference :230-232)."""
        eng = self.engine(v_listener.device)
        xl, lens = compact_by_mask(v_listener, mask.bool())
        return eng.vq_encode(1, xl.contiguous(), lens, pe_mode=0, pad_value=-100).long()

    def get_3dmm_loss(self, pred, gt):
        b, t, c = pred.shape
        loss = torch.mean(F.pairwise_distance(pred.reshape(b * t, c), gt.reshape(b * t, c)))
        return loss, self.get_spiky_loss(pred, gt)

    def get_spiky_loss(self, pred, gt):
        b, t, c = pred.shape
        ps = (pred[:, 1:, :] - pred[:, :-1, :]).reshape(b * (t - 1), c)
        gs = (gt[:, 1:, :] - gt[:, :-1, :]).reshape(b * (t - 1), c)
        return torch.mean(F.pairwise_distance(ps, gs))

    # ------------------------------------------------------------------ training (reference code/x_engine.py:8-36)
    def train(self, mode=True):
        """nn.Module.train with both VQ-VAEs kept in eval (reference :166, :170)."""
        super().train(mode)
        self.speaker_vq.eval()
        self.listener_vq.eval()
        return self

    def dimx_trainable_parameters(self):
        from . import train as T
        return T.legacy_trainable_parameters(self)

    def _wants_grad(self):
        return self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _forward_autograd(self, v_speaker, v_listener, mask, speaker_ids=None, listener_ids=None, return_logits=False):
        """The loop body's ``model(src, tgt, mask, speaker_ids=None, listener_ids=listener_ids)`` with a graph: frozen
        halves (speaker features, listener codes) from the HIP engine, the generator + the listener VQ-VAE's decoder
        + the id embeddings through dimx.train.legacy_loss."""
        from . import train as T
        mask = mask.bool()
        with torch.no_grad():
            eng, xs, xl, lens, m8 = self._prepare(v_speaker, v_listener, mask)
            z_l = eng.vq_encode(1, xl, lens, pe_mode=0, pad_value=-100).long()
            x_speaker = eng.legacy_speaker_features(xs, m8).clone()
        P = dict(self.named_parameters())
        pe = self.listener_vq.decoder.decoder_pos_embedding.pe
        loss, pred, logits = T.legacy_loss(P, self.dims, self.vq_dims, x_speaker, z_l, v_listener.float(), mask, pe,
                                           speaker_ids=speaker_ids, listener_ids=listener_ids)
        self.last_logits = logits.detach()
        if return_logits:
            return loss, pred, logits
        return loss, pred

    # ------------------------------------------------------------------ forward / generate
    def forward(self, v_speaker, v_listener, mask, speaker_ids=None, listener_ids=None, return_logits=False):
        """reference :220-278 -> (loss, pred_cont_seq [B,T-1,56]); ``self.last_logits`` keeps the [B,T-1,512]
        teacher-forced logits (the reference discards them; code/x_engine.py:78 wants them for perplexity).
        In training (``model.train()``, grad enabled, parameters requiring grad) the loss carries an autograd graph."""
        if self._wants_grad():
            return self._forward_autograd(v_speaker, v_listener, mask, speaker_ids, listener_ids, return_logits)
        with torch.no_grad():
            return self._forward_nograd(v_speaker, v_listener, mask, speaker_ids, listener_ids, return_logits)

    def _forward_nograd(self, v_speaker, v_listener, mask, speaker_ids=None, listener_ids=None, return_logits=False):
        if speaker_ids is not None or listener_ids is not None:
            raise NotImplementedError("speaker_ids / listener_ids conditioning is a training-only path "
                                      "(code/x_engine.py:23); evaluation calls model(src, tgt, mask)")
        eng, xs, xl, lens, m8 = self._prepare(v_speaker, v_listener, mask)
        mask = mask.bool()
        z_l = eng.vq_encode(1, xl, lens, pe_mode=0, pad_value=-100).long()
        eng.encode_ctx(xs, None, m8, False)
        logits, row_loss, amax = eng.decode_tf(z_l, m8, None)
        n_valid = (z_l[:, 1:] != -100).sum().clamp(min=1)
        loss = row_loss.sum() / n_valid
        pred = eng.vq_decode(1, amax, 0)
        B, T = mask.shape
        m = mask[:, 1:].reshape(B * (T - 1))
        p = pred.reshape(B * (T - 1), -1)[m]
        t = v_listener[:, 1:, :].reshape(B * (T - 1), -1)[m]
        loss_cont = torch.mean(F.pairwise_distance(p[:, 6:], t[:, 6:])) + torch.mean(F.pairwise_distance(p[:, 0:6], t[:, 0:6]))
        self.last_logits = logits
        if return_logits:
            return loss + loss_cont, pred, logits
        return loss + loss_cont, pred

    @torch.no_grad()
    def generate(self, v_speaker, v_listener, mask, noise=None, greedy=False, seed=None, temperature=1.0,
                 n_samples=1):
        """reference :280-306 -> (z_listener_pred [B,T] (or [B,S,T]), z_listener [B,T])."""
        eng, xs, xl, lens, m8 = self._prepare(v_speaker, v_listener, mask)
        z_l = eng.vq_encode(1, xl, lens, pe_mode=0, pad_value=-100).long()
        eng.encode_ctx(xs, None, m8, True, n_samples=n_samples)
        if greedy:
            temperature, seed_v = 0.0, 0
        else:
            # seed 0 is reserved by the C-ABI for "greedy when no noise is given": remap it
            from .seq2seq_pretrain import SLMFT as _S
            seed_v = 0 if noise is not None else _S._user_seed(seed)
        T = z_l.shape[1]
        tok = eng.generate(z_l[:, 0], m8, T, temperature, 52, noise, seed_v, n_samples=n_samples).long()
        if n_samples > 1:
            tok = tok.view(z_l.shape[0], n_samples, T)
        return tok, z_l
