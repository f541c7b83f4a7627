"""Training step of the DIM-Listener fine-tuning model on hand-written HIP kernels.

Reference: ``train_epoch`` (code/x_engine_pt.py:9-60) as driven by code/finetune_s2s_pretrain.py:105-143 -- AdamW
(lr 1e-5), gradient clipping at 1.0, both VQ-VAEs frozen.  The reference differentiates ``SLMFT.forward(mode='train')``
with autograd; here the forward AND the backward pass of the teacher-forced stack run in libdimx_hip.so
(csrc/train.hip, csrc/train_kernels.hip): every Linear and both of its adjoints on the library's MFMA GEMM, attention /
LayerNorm / GELU / cross-entropy adjoints as HIP kernels, fused clip + AdamW.  ``dimx.train`` (the PyTorch-autograd
restatement of round 2) stays as the CHECKER of this path (tests/test_gpu_train_hip.py) and as the CPU fallback-free
reference of the gradient math; it is no longer what ``x_engine_pt.train_epoch`` runs on a GPU.

State: parameters, gradients and the two AdamW moments are four flat f32 device tensors laid out by the library
(``dimx_train_param_info``); the module's ``nn.Parameter``s are written back by ``sync_to_model()`` (end of an epoch,
before evaluation or ``state_dict()``).  Multi-GPU: one process per GPU, replicated weights, per-rank batch shard, and
ONE all-reduce of the flat gradient tensor per step over RCCL (xGMI rings are per-link bound: one 0.4 GB collective
instead of one per tensor)."""
import ctypes

import torch

from . import dist as ddist
from . import lib as L


class HipTrainer:
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip=1.0, device=None):
        """AdamW defaults are torch.optim.AdamW's (the reference passes only lr, code/finetune_s2s_pretrain.py:119)."""
        self.model = model
        self.device = torch.device(device if device is not None else next(model.parameters()).device)
        if self.device.type != "cuda":
            raise L.DimxError("HipTrainer needs the module on a ROCm GPU (model.to('cuda:0')); there is no CPU path")
        self.lr, self.betas, self.eps, self.weight_decay, self.clip = lr, betas, eps, weight_decay, clip
        self.lib = L.load()
        self.eng = model.engine(self.device)
        h = self.eng.h
        n = self.lib.dimx_train_num_params(h)
        if n <= 0:
            L.check(n, "dimx_train_num_params")
        self.total = int(self.lib.dimx_train_total(h))
        self.layout = []
        for i in range(n):
            name, off, numel = ctypes.c_char_p(), ctypes.c_int64(), ctypes.c_int64()
            L.check(self.lib.dimx_train_param_info(h, i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(numel)), "train_param_info")
            self.layout.append((name.value.decode(), int(off.value), int(numel.value)))
        kw = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.total, **kw)
        self.grads = torch.zeros(self.total, **kw)
        self.exp_avg = torch.zeros(self.total, **kw)
        self.exp_avg_sq = torch.zeros(self.total, **kw)
        self._scratch = torch.zeros(1026, **kw)
        self._loss = torch.zeros(2, **kw)
        self._ws, self._ws_bytes = None, 0
        self._stage = {}
        self.step_count = 0
        self.load_from_model()
        if ddist.world_size() > 1:                      # every rank starts from rank 0's parameters
            import torch.distributed as dist
            dist.broadcast(self.params, 0)

    # ------------------------------------------------------------------ arena <-> module
    def _named(self):
        return dict(self.model.named_parameters())

    def _versions(self):
        named = self._named()
        return tuple(int(named[name]._version) for name, _, _ in self.layout)

    def load_from_model(self):
        named = self._named()
        with torch.no_grad():
            for name, off, numel in self.layout:
                self.params[off:off + numel].copy_(named[name].detach().reshape(-1).to(self.device, torch.float32))
        self._synced_versions = self._versions()

    def refresh_from_model_if_changed(self):
        """the arenas are the master copy between ``sync_to_model()`` calls; if somebody wrote the module's parameters in the
        meantime (``load_state_dict`` of a checkpoint, a manual re-initialisation), adopt them instead of overwriting them at the
        end of the next epoch.  Detected through the tensors' version counters.  Returns True when the arena was reloaded."""
        if getattr(self, "_synced_versions", None) == self._versions():
            return False
        self.load_from_model()
        return True

    def sync_to_model(self):
        """write the trained parameters back into the nn.Module (its engine re-packs them before its next launch)."""
        named = self._named()
        with torch.no_grad():
            for name, off, numel in self.layout:
                p = named[name]
                p.copy_(self.params[off:off + numel].view_as(p))
        self._synced_versions = self._versions()

    # ------------------------------------------------------------------ torch.optim.AdamW <-> arenas
    def import_optimizer_state(self, optimizer):
        """adopt the moments / step count a torch.optim.AdamW already holds for the trained parameters (resuming from an
        ``optimizer.load_state_dict``); parameters without state start from zero moments like a fresh AdamW."""
        named = self._named()
        step = 0
        with torch.no_grad():
            for name, off, numel in self.layout:
                st = optimizer.state.get(named[name])
                if not st:
                    # no state = a fresh AdamW for this parameter: ZERO moments, not whatever the arena held (a cleared
                    # optimizer.state or a reload after an interrupted epoch resumed on stale moments)
                    self.exp_avg[off:off + numel].zero_()
                    self.exp_avg_sq[off:off + numel].zero_()
                    continue
                self.exp_avg[off:off + numel].copy_(st["exp_avg"].reshape(-1).to(self.device, torch.float32))
                self.exp_avg_sq[off:off + numel].copy_(st["exp_avg_sq"].reshape(-1).to(self.device, torch.float32))
                step = max(step, int(st["step"]))
        # the optimizer is the authority at this point (a freshly loaded state_dict may carry a SMALLER step count than the
        # arenas' last one): take its count, not the maximum
        self.step_count = step
        self._state_sig = self._optimizer_state_signature(optimizer)

    def _optimizer_state_signature(self, optimizer):
        """identity + version of the moment tensors torch keeps for the trained parameters: changes when optimizer.load_state_dict
        replaced them or an autograd epoch stepped them in place since the last import / export"""
        named = self._named()
        sig = []
        for name, _, _ in self.layout:
            st = optimizer.state.get(named[name])
            if st and "exp_avg" in st:
                sig.append((id(st["exp_avg"]), st["exp_avg"]._version, id(st["exp_avg_sq"]), st["exp_avg_sq"]._version))
            else:
                sig.append(None)
        return sig

    def optimizer_state_changed(self, optimizer):
        return getattr(self, "_state_sig", None) != self._optimizer_state_signature(optimizer)

    def export_optimizer_state(self, optimizer):
        """write the moments and the step count into ``optimizer.state`` (the entries torch.optim.AdamW keeps per parameter), so
        ``optimizer.state_dict()`` / a later autograd epoch continue from what the HIP steps left."""
        named = self._named()
        with torch.no_grad():
            for name, off, numel in self.layout:
                p = named[name]
                st = optimizer.state[p]
                for key, arena in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                    src = arena[off:off + numel].view_as(p)
                    if key in st and st[key].shape == p.shape and st[key].device == p.device:
                        st[key].copy_(src)
                    else:
                        st[key] = src.to(p.device, copy=True)
                st["step"] = torch.tensor(float(self.step_count))
        self._state_sig = self._optimizer_state_signature(optimizer)

    def view(self, arena, name):
        for n, off, numel in self.layout:
            if n == name:
                return arena[off:off + numel].view(self._named()[name].shape)
        raise KeyError(name)

    def grad(self, name):
        return self.view(self.grads, name)

    # ------------------------------------------------------------------ one step
    def _workspace(self, B, T):
        need = int(self.lib.dimx_train_workspace_bytes(self.eng.h, B, T))
        if need == 0:
            raise L.DimxError("dimx_train_workspace_bytes(B=%d, T=%d) = 0: %s" % (B, T, (self.lib.dimx_last_error() or b"").decode()))
        if need > self._ws_bytes:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        base = self._ws.data_ptr()
        return ctypes.c_void_p((base + 255) // 256 * 256), self._ws.numel() - 256

    def _staging(self, B, T, d_s, d_a):
        st = self._stage.get((B, T, d_s, d_a))
        if st is None:
            kw = dict(device=self.device)
            st = {"v_s": torch.empty(B, T, d_s, dtype=torch.float32, **kw), "v_a": torch.empty(B, T, d_a, dtype=torch.float32, **kw),
                  "mask": torch.empty(B, T, dtype=torch.uint8, **kw), "z": torch.empty(B, T, dtype=torch.int32, **kw),
                  "kv": torch.empty(B, T - 1, dtype=torch.uint8, **kw)}
            box = {}

            def logits():
                if "t" not in box:
                    box["t"] = torch.empty(B, T - 1, 512, dtype=torch.float32, **kw)
                return box["t"]
            st["logits"] = logits
            if len(self._stage) >= 8:          # a handful of (B, T) buckets; ragged epochs do not grow it without bound
                self._stage.pop(next(iter(self._stage)))
            self._stage[(B, T, d_s, d_a)] = st
        return st

    def graph_stats(self):
        """(steps replayed from the captured hipGraph, steps launched kernel by kernel, nodes of the captured graph)"""
        out = (ctypes.c_int64 * 3)()
        L.check(self.lib.dimx_train_graph_stats(self.eng.h, ctypes.cast(out, ctypes.c_void_p)), "dimx_train_graph_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def forward_backward(self, v_speaker, v_listener, v_audio, mask, kv_mask=None, z_l=None, return_logits=False, _alias_logits=False):
        """loss (0-dim device tensor, mean cross entropy over the valid listener codes) and the gradients in ``self.grads``.
        kv_mask: keep-mask [B,T-1] of the mask_prob draw; None draws one like the reference, False disables it.
        return_logits: a tensor of the caller's own (the step writes into a staging buffer that the next call with the same (B, T)
        overwrites -- possibly from a hipGraph replay; only train_step, which consumes the logits at once, takes the alias)."""
        mask = mask.bool()
        B, T = mask.shape
        if z_l is None:
            with torch.no_grad():
                _, z_l = self.model.forward_vq(v_speaker, v_listener, mask, with_speaker=False)
        if kv_mask is None:
            kv_mask = self.model.draw_kv_mask(B, T, self.device)
        elif kv_mask is False:
            kv_mask = None
        # the batch is staged in buffers that persist per (B, T): a call whose pointers equal the previous call's is replayed
        # from the captured hipGraph of the step (dimx_train_forward_backward), fresh tensors every step would keep it on the
        # kernel-by-kernel path
        st = self._staging(B, T, int(v_speaker.shape[-1]), int(v_audio.shape[-1]))
        v_s, v_a, m8, z32 = st["v_s"], st["v_a"], st["mask"], st["z"]
        v_s.copy_(v_speaker, non_blocking=True)
        v_a.copy_(v_audio, non_blocking=True)
        m8.copy_(mask, non_blocking=True)
        z32.copy_(z_l, non_blocking=True)
        kv8 = None
        if kv_mask is not None:
            kv8 = st["kv"]
            kv8.copy_(kv_mask, non_blocking=True)
        logits = st["logits"]() if return_logits else None
        ws, wsb = self._workspace(B, T)
        L.check(self.lib.dimx_train_forward_backward(self.eng.h, L.ptr(self.params), L.ptr(self.grads), L.ptr(v_s), L.ptr(v_a),
                                                     L.ptr(m8), L.ptr(z32), L.ptr(kv8), B, T, L.ptr(self._loss), L.ptr(logits), ws,
                                                     wsb, L.stream_ptr(self.device)), "dimx_train_forward_backward")
        loss = self._loss[0].clone()
        if return_logits and not _alias_logits:
            logits = logits.clone()
        return (loss, logits) if return_logits else loss

    def all_reduce_grads(self):
        world = ddist.world_size()
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.grads)
            self.grads.div_(world)

    def step(self):
        """clip (torch.nn.utils.clip_grad_norm_ semantics) + AdamW; returns the gradient norm before clipping (device scalar)."""
        self.step_count += 1
        L.check(self.lib.dimx_train_adamw(L.ptr(self.params), L.ptr(self.grads), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                          self.total, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                          float(self.weight_decay), self.step_count, float(self.clip or 0.0), L.ptr(self._scratch),
                                          L.stream_ptr(self.device)), "dimx_train_adamw")
        return self._scratch[1024]

    def train_step(self, v_speaker, v_listener, v_audio, mask, kv_mask=None, z_l=None, with_cont_loss=False):
        """one optimisation step; returns l_ce_l (device scalar), or (total, dict) like SLMFT.forward(mode='train') when
        with_cont_loss: the reference's total also carries the continuous loss of the decoded arg-max codes, which has no
        gradient path (code/seq2seq_pretrain.py:454-478) -- it is evaluated from this step's logits on the HIP engine."""
        if not with_cont_loss:
            loss = self.forward_backward(v_speaker, v_listener, v_audio, mask, kv_mask=kv_mask, z_l=z_l)
            self.all_reduce_grads()
            self.step()
            return loss
        l_ce, logits = self.forward_backward(v_speaker, v_listener, v_audio, mask, kv_mask=kv_mask, z_l=z_l, return_logits=True,
                                             _alias_logits=True)
        self.all_reduce_grads()
        self.step()
        with torch.no_grad():
            pred = self.model.forward_vq_decoder(logits, mode="train")
            l_cont = self.model.forward_continuous_loss(pred, v_listener.to(self.device), mask.bool().to(self.device))
        return l_ce + l_cont, {"l_ce_s": 0, "l_ce_l": l_ce, "l_cont_s": 0, "l_cont_l": l_cont, "nce": 0, "c_acc": 0}


class LegacyHipTrainer(HipTrainer):
    """The legacy ListenerGenerator's training step on the HIP kernels (reference loop code/x_engine.py:8-36):
    the generator (bidirectional encoder, decoder with absolute positions), the listener-id conditioning and the listener
    VQ-VAE's DECODER train; the frozen halves (speaker features, listener codes) come from the engine without a graph.  Same
    flat arenas, clip + AdamW and gradient all-reduce as HipTrainer; ``dimx.train.legacy_loss`` (PyTorch autograd) is its checker
    (tests/test_gpu_train_legacy.py)."""

    def forward_backward(self, v_speaker, v_listener, mask, listener_ids=None, return_logits=False):
        """-> (total loss = cross entropy + continuous loss, dict, pred [B,T-1,56][, logits]); gradients in ``self.grads``."""
        m = self.model
        mask = mask.bool()
        B, T = mask.shape
        with torch.no_grad():
            eng, xs, xl, lens, m8 = m._prepare(v_speaker, v_listener, mask)
            z_l = eng.vq_encode(1, xl, lens, pe_mode=0, pad_value=-100).to(torch.int32).contiguous()
            x_speaker = eng.legacy_speaker_features(xs, m8).float().contiguous().clone()
        v_l = v_listener.to(self.device, torch.float32).contiguous()
        ids = listener_ids.to(self.device, torch.int32).contiguous() if listener_ids is not None else None
        book = m.listener_vq.quantize.embedding.weight.detach().to(self.device, torch.float32).contiguous()
        pe = m.listener_vq.decoder.decoder_pos_embedding.pe.detach().to(self.device, torch.float32).contiguous()
        nd = T if ids is not None else T - 1
        pred = torch.empty(B, T - 1, 56, dtype=torch.float32, device=self.device)
        logits = torch.empty(B, nd, 512, dtype=torch.float32, device=self.device) if return_logits else None
        need = int(self.lib.dimx_train_legacy_workspace_bytes(self.eng.h, B, T))
        if need == 0:
            raise L.DimxError("dimx_train_legacy_workspace_bytes(B=%d, T=%d) = 0: %s" % (B, T, (self.lib.dimx_last_error() or b"").decode()))
        if need > self._ws_bytes:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        ws = ctypes.c_void_p((self._ws.data_ptr() + 255) // 256 * 256)
        if self._loss.numel() < 4:
            self._loss = torch.zeros(4, dtype=torch.float32, device=self.device)
        L.check(self.lib.dimx_train_legacy_forward_backward(
            self.eng.h, L.ptr(self.params), L.ptr(self.grads), L.ptr(x_speaker), L.ptr(z_l), L.ptr(v_l), L.ptr(m8), L.ptr(ids), L.ptr(book),
            L.ptr(pe), B, T, L.ptr(self._loss), L.ptr(pred), L.ptr(logits), ws, self._ws.numel() - 256, L.stream_ptr(self.device)),
            "dimx_train_legacy_forward_backward")
        l_ce, l_cont = self._loss[0].clone(), self._loss[2].clone()
        out = (l_ce + l_cont, {"l_ce": l_ce, "l_cont": l_cont}, pred)
        return out + (logits,) if return_logits else out

    def train_step(self, v_speaker, v_listener, mask, listener_ids=None):
        """one optimisation step of the reference loop's body; returns (loss, pred) like ``model(src, tgt, mask, listener_ids=...)``."""
        loss, _, pred = self.forward_backward(v_speaker, v_listener, mask, listener_ids=listener_ids)
        self.all_reduce_grads()
        self.step()
        return loss, pred


class SlmHipTrainer(HipTrainer):
    """The SLM pre-training step on the HIP kernels (reference loop code/train_s2s_pretrain.py:41-64 ->
    x_engine_pt.train_epoch over SLM.forward, code/seq2seq_pretrain.py:300-323): the three encoders (the joint one over the 2T
    concatenation and over each stream, one pass), InfoNCE, the decoder for both streams (one pass over 2B sequences), both cross
    entropies and the continuous losses through BOTH VQ-VAE decoders, which train.  The frozen VQ encoders run on the inference
    engine.  Same flat arenas, clip + AdamW and gradient all-reduce as HipTrainer; ``dimx.train.slm_loss`` (PyTorch autograd) is
    its checker (tests/test_gpu_train_slm.py)."""

    KEYS = ("l_ce_s", "l_ce_l", "l_cont_s", "l_cont_l", "nce", "c_acc")

    def _frozen(self):
        m = self.model
        c = getattr(self, "_frozen_cache", None)
        if c is None:
            f = lambda t: t.detach().to(self.device, torch.float32).contiguous()
            c = (f(m.speaker_vq.quantize.embedding.weight), f(m.listener_vq.quantize.embedding.weight),
                 f(m.speaker_vq.decoder.decoder_pos_embedding.pe), f(m.listener_vq.decoder.decoder_pos_embedding.pe))
            self._frozen_cache = c
        return c

    def forward_backward(self, v_speaker, v_listener, v_audio, mask, mask_speaker=None, mask_listener=None, z_s=None, z_l=None,
                         mask_ratio=0.15):
        """-> (total loss, dict of the reference's six entries) as device scalars; gradients in ``self.grads``.  mask_speaker /
        mask_listener (True = frame masked out) are drawn like the reference when not given."""
        m = self.model
        mask = mask.bool()
        B, T = mask.shape
        with torch.no_grad():
            if z_s is None or z_l is None:
                z_s, z_l = m.forward_vq(v_speaker, v_listener, mask)
            if mask_speaker is None:
                mask_speaker = m.random_masking_unstructured(v_speaker, mask, mask_ratio)
            if mask_listener is None:
                mask_listener = m.random_masking_unstructured(v_listener, mask, mask_ratio)
        f = lambda t: t.to(self.device, torch.float32).contiguous()
        u8 = lambda t: t.to(self.device).to(torch.uint8).contiguous()
        i32 = lambda t: t.to(self.device).to(torch.int32).contiguous()
        v_s, v_l, v_a = f(v_speaker), f(v_listener), f(v_audio)
        m8, ms8, ml8, zs, zl = u8(mask), u8(mask_speaker), u8(mask_listener), i32(z_s), i32(z_l)
        book_s, book_l, pe_s, pe_l = self._frozen()
        need = int(self.lib.dimx_train_slm_workspace_bytes(self.eng.h, B, T))
        if need == 0:
            raise L.DimxError("dimx_train_slm_workspace_bytes(B=%d, T=%d) = 0: %s" % (B, T, (self.lib.dimx_last_error() or b"").decode()))
        if need > self._ws_bytes:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        ws = ctypes.c_void_p((self._ws.data_ptr() + 255) // 256 * 256)
        if self._loss.numel() < 10:
            self._loss = torch.zeros(10, dtype=torch.float32, device=self.device)
        L.check(self.lib.dimx_train_slm_forward_backward(
            self.eng.h, L.ptr(self.params), L.ptr(self.grads), L.ptr(v_s), L.ptr(v_l), L.ptr(v_a), L.ptr(m8), L.ptr(ms8), L.ptr(ml8),
            L.ptr(zs), L.ptr(zl), L.ptr(book_s), L.ptr(book_l), L.ptr(pe_s), L.ptr(pe_l), B, T, L.ptr(self._loss), ws,
            self._ws.numel() - 256, L.stream_ptr(self.device)), "dimx_train_slm_forward_backward")
        out = self._loss.clone()
        d = {"l_ce_s": out[0], "l_ce_l": out[2], "l_cont_s": out[4], "l_cont_l": out[6], "nce": out[8], "c_acc": out[9]}
        return out[0] + out[2] + out[4] + out[6] + out[8], d

    def train_step(self, v_speaker, v_listener, v_audio, mask, with_cont_loss=True, **kw):
        """one optimisation step of the reference loop's body; returns (total, d) like ``SLM.forward`` (its third value is None)."""
        total, d = self.forward_backward(v_speaker, v_listener, v_audio, mask, **kw)
        self.all_reduce_grads()
        self.step()
        return total, d
