"""Engine: one libdimx_hip handle on one GPU + its device workspace.

This is the thin host layer above the C-ABI: it owns the handle, uploads a reference-shaped
state dict, sizes the workspace with ``dimx_workspace_bytes`` and forwards torch CUDA tensors
as raw pointers on the current HIP stream.  All arithmetic happens in the HIP library.
"""
import ctypes

import numpy as np
import torch

from . import lib as L


class Engine:
    def __init__(self, device=None, mode=L.MODE_PARITY_F32, variant="slmft"):
        """variant: "slmft" (DIM-Listener, code/seq2seq_pretrain.py) or "legacy" (ListenerGenerator,
        code/seq2seq.py)."""
        if not torch.cuda.is_available():
            raise L.DimxError("dimx needs a ROCm GPU (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = L.load()
        self.mode = mode
        assert variant in ("slmft", "legacy", "slm")
        self.variant = variant
        self.dims = {"slmft": L.default_dims, "legacy": L.legacy_dims, "slm": L.slm_dims}[variant]()
        h = ctypes.c_void_p()
        L.check(self.lib.dimx_create(ctypes.byref(h), self.device.index, ctypes.byref(self.dims), mode),
                "dimx_create")
        self.h = h
        self._ws = None
        self._ws_bytes = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.dimx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, new_checkpoint=True):
        """Upload every tensor of a reference-shaped state dict (extra reference keys that
        are not on the hot path are ignored by the library).  ``new_checkpoint`` (default): ``sd`` is a whole checkpoint, the
        optional tensors an earlier one brought (``project_in.bias`` / ``to_logits.bias``) are forgotten first
        (``dimx_begin_checkpoint``); pass False for the further chunks of a checkpoint loaded in pieces."""
        if new_checkpoint:
            L.check(self.lib.dimx_begin_checkpoint(self.h), "dimx_begin_checkpoint")
        keep, descs = [], []
        for name, t in sd.items():
            a = t.detach().to("cpu", torch.float32).contiguous().numpy()
            keep.append(a)
            d = L.WeightDesc()
            d.name = name.encode()
            d.data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            d.ndim = a.ndim
            for i, s in enumerate(a.shape):
                d.shape[i] = s
            descs.append(d)
        arr = (L.WeightDesc * len(descs))(*descs)
        L.check(self.lib.dimx_load_weights(self.h, arr, len(descs)), "dimx_load_weights")

    def missing_weights(self):
        return self.lib.dimx_missing_weights(self.h)

    # ------------------------------------------------------------------ workspace
    def workspace(self, B, T, n_samples=1):
        need = self.lib.dimx_workspace_bytes_samples(self.h, B, T, n_samples)
        if need == 0:
            raise L.DimxError("dimx_workspace_bytes(%d,%d) = 0" % (B, T))
        if self._ws is None or self._ws_bytes < need:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        return ctypes.c_void_p(aligned), self._ws_bytes - (aligned - base) + 256

    def _s(self):
        return L.stream_ptr(self.device)

    def _chk(self, *ts):
        for t in ts:
            if t is not None:
                assert t.device == self.device and t.is_contiguous(), "tensor must be contiguous on %s" % self.device

    # ------------------------------------------------------------------ stages
    def vq_encode(self, which, x, lens=None, pe_mode=0, row_offset=0, pad_value=-100, return_z=False):
        B, T, _ = x.shape
        x = x.to(torch.float32).contiguous()
        self._chk(x, lens)
        fqn = self.dims.spk_face_quan_num if (self.variant == "legacy" and which == 0) else 1
        idx = torch.empty(B, T * fqn, dtype=torch.int32, device=self.device)
        z = torch.empty(B, T, fqn * self.dims.vq_zdim, dtype=torch.float32, device=self.device) if return_z else None
        ws, wsb = self.workspace(B, T)
        L.check(self.lib.dimx_vq_encode(self.h, which, L.ptr(x), L.ptr(lens), B, T, pe_mode, row_offset,
                                        pad_value, L.ptr(idx), L.ptr(z), ws, wsb, self._s()), "dimx_vq_encode")
        return (idx, z) if return_z else idx

    def vq_argmin(self, which, z, with_stats=False):
        z = z.to(torch.float32).contiguous()
        N = z.shape[0]
        idx = torch.empty(N, dtype=torch.int32, device=self.device)
        bd = torch.empty(N, dtype=torch.float32, device=self.device) if with_stats else None
        mg = torch.empty(N, dtype=torch.float32, device=self.device) if with_stats else None
        L.check(self.lib.dimx_vq_argmin(self.h, which, L.ptr(z), N, L.ptr(idx), L.ptr(bd), L.ptr(mg), self._s()),
                "dimx_vq_argmin")
        return (idx, bd, mg) if with_stats else idx

    def vq_decode(self, which, idx, row_offset=0, rows_per_clip=1):
        B, Lq = idx.shape
        idx = idx.to(torch.int32).contiguous()
        self._chk(idx)
        out = torch.empty(B, Lq, self.dims.vq_in_dim, dtype=torch.float32, device=self.device)
        ws, wsb = self.workspace(B, Lq)
        L.check(self.lib.dimx_vq_decode(self.h, which, L.ptr(idx), B, Lq, row_offset, rows_per_clip, L.ptr(out), ws, wsb,
                                        self._s()), "dimx_vq_decode")
        return out

    def vq_decode_latent(self, which, z, row_offset=0):
        """z [B,L,128] f32 (time-major latents, quantised or not) -> [B,L,56]: VQAutoEncoder.decode on what it is given."""
        B, Lq, _ = z.shape
        z = z.to(torch.float32).contiguous()
        self._chk(z)
        out = torch.empty(B, Lq, self.dims.vq_in_dim, dtype=torch.float32, device=self.device)
        ws, wsb = self.workspace(B, Lq)
        L.check(self.lib.dimx_vq_decode_latent(self.h, which, L.ptr(z), B, Lq, row_offset, L.ptr(out), ws, wsb,
                                               self._s()), "dimx_vq_decode_latent")
        return out

    def encode_speaker(self, v_speaker, mask_u8):
        """SLMFT.forward_encoder alone -> x_s [B,T,384] f32 (no audio, no context, no K/V projection)."""
        B, T, _ = v_speaker.shape
        v_speaker = v_speaker.to(torch.float32).contiguous()
        self._chk(v_speaker, mask_u8)
        x_s = torch.empty(B, T, self.dims.dim, dtype=torch.float32, device=self.device)
        ws, wsb = self.workspace(B, T)
        L.check(self.lib.dimx_encode_speaker(self.h, L.ptr(v_speaker), L.ptr(mask_u8), B, T, L.ptr(x_s), ws, wsb,
                                             self._s()), "dimx_encode_speaker")
        return x_s

    def set_shard(self, row_offset=0, rows_total=0):
        """This engine generates clips [row_offset, row_offset+B) of a sharded batch of rows_total clips (only the
        sampler's counter-based generator depends on it); (0, 0) = unsharded."""
        L.check(self.lib.dimx_set_shard(self.h, int(row_offset), int(rows_total)), "dimx_set_shard")

    def encode_ctx(self, v_speaker, v_audio, mask_u8, for_generate, return_x_s=False, n_samples=1):
        B, T, _ = v_speaker.shape
        v_speaker = v_speaker.to(torch.float32).contiguous()
        v_audio = v_audio.to(torch.float32).contiguous() if v_audio is not None else None
        if self.variant == "slmft" and v_audio is None:
            raise L.DimxError("encode_ctx: the SLMFT variant needs v_audio")
        self._chk(v_speaker, v_audio, mask_u8)
        x_s = torch.empty(B, T, self.dims.dim, dtype=torch.float32, device=self.device) if return_x_s else None
        ws, wsb = self.workspace(B, T, n_samples)   # the generate call that follows must see the same workspace
        L.check(self.lib.dimx_encode_ctx(self.h, L.ptr(v_speaker), L.ptr(v_audio), L.ptr(mask_u8), B, T,
                                         1 if for_generate else 0, L.ptr(x_s), ws, wsb, self._s()),
                "dimx_encode_ctx")
        return x_s

    def slm_encode(self, v_speaker, v_listener, mask_u8, mask_speaker_u8=None, mask_listener_u8=None):
        """SLM.forward_encoder -> (x_s, x_l [B,T,384], x_joint [B,2T,384]) f32."""
        B, T, _ = v_speaker.shape
        v_speaker = v_speaker.to(torch.float32).contiguous()
        v_listener = v_listener.to(torch.float32).contiguous()
        self._chk(v_speaker, v_listener, mask_u8, mask_speaker_u8, mask_listener_u8)
        dim = self.dims.dim
        x_s = torch.empty(B, T, dim, dtype=torch.float32, device=self.device)
        x_l = torch.empty(B, T, dim, dtype=torch.float32, device=self.device)
        x_j = torch.empty(B, 2 * T, dim, dtype=torch.float32, device=self.device)
        ws, wsb = self.workspace(B, T)
        L.check(self.lib.dimx_slm_encode(self.h, L.ptr(v_speaker), L.ptr(v_listener), L.ptr(mask_u8),
                                         L.ptr(mask_speaker_u8), L.ptr(mask_listener_u8), B, T, L.ptr(x_s), L.ptr(x_l),
                                         L.ptr(x_j), ws, wsb, self._s()), "dimx_slm_encode")
        return x_s, x_l, x_j

    def set_context(self, x, v_audio, which_patch=0, for_generate=False, T=None, n_samples=1):
        """context = cat(x[:, :T] + patch_embed_dec_{s|l}, v_audio) + cross K/V; x [B, rows>=T, dim] f32."""
        B, rows, _ = x.shape
        T = T or rows
        x = x.to(torch.float32)
        v_audio = v_audio.to(torch.float32).contiguous()
        assert x.stride(2) == 1 and x.stride(1) == x.shape[2], "x rows must be dense"
        ldx_rows = x.stride(0) // x.shape[2]
        ws, wsb = self.workspace(B, T, n_samples)
        L.check(self.lib.dimx_set_context(self.h, ctypes.c_void_p(x.data_ptr()), ldx_rows, which_patch, L.ptr(v_audio),
                                          B, T, 1 if for_generate else 0, ws, wsb, self._s()), "dimx_set_context")

    def legacy_speaker_features(self, v_speaker, mask_u8, return_idx=False):
        """x_speaker [B,T,1024] of ListenerGenerator.forward (code/seq2seq.py:224-241); v_speaker must hold each
        clip's valid frames first."""
        B, T, _ = v_speaker.shape
        v_speaker = v_speaker.to(torch.float32).contiguous()
        self._chk(v_speaker, mask_u8)
        fqn = self.dims.spk_face_quan_num
        x = torch.empty(B, T, fqn * self.dims.vq_zdim, dtype=torch.float32, device=self.device)
        idx = torch.empty(B, T * fqn, dtype=torch.int32, device=self.device) if return_idx else None
        ws, wsb = self.workspace(B, T)
        L.check(self.lib.dimx_legacy_speaker_features(self.h, L.ptr(v_speaker), L.ptr(mask_u8), B, T, L.ptr(x),
                                                      L.ptr(idx), ws, wsb, self._s()),
                "dimx_legacy_speaker_features")
        return (x, idx) if return_idx else x

    def n_gen(self, T):
        """tokens generated per sequence: T-1 (SLMFT) / T (legacy, code/seq2seq.py:300)."""
        return T if self.variant == "legacy" else T - 1

    def decode_tf(self, z_l, mask_u8, kv_mask_u8=None):
        B, T = z_l.shape
        z_l = z_l.to(torch.int32).contiguous()
        self._chk(z_l, mask_u8, kv_mask_u8)
        n = T - 1
        logits = torch.empty(B, n, self.dims.num_tokens, dtype=torch.float32, device=self.device)
        row_loss = torch.empty(B, n, dtype=torch.float32, device=self.device)
        amax = torch.empty(B, n, dtype=torch.int32, device=self.device)
        ws, wsb = self.workspace(B, T)
        L.check(self.lib.dimx_decode_tf(self.h, L.ptr(z_l), L.ptr(mask_u8), L.ptr(kv_mask_u8), B, T, L.ptr(logits),
                                        L.ptr(row_loss), L.ptr(amax), ws, wsb, self._s()), "dimx_decode_tf")
        return logits, row_loss, amax

    def generate(self, start, mask_u8, T, temperature=1.0, top_k=52, noise=None, seed=0, return_logits=False,
                 n_samples=1):
        """n_samples S > 1: S sequences per clip in one pass (rows b*S+s), sharing the clip's context K/V."""
        B = start.shape[0]
        R = B * n_samples
        start = start.to(torch.int32).contiguous()
        if noise is not None:
            noise = noise.to(torch.float32).contiguous()
            assert tuple(noise.shape) == (self.n_gen(T), R, self.dims.num_tokens)
        self._chk(start, mask_u8, noise)
        n = self.n_gen(T)
        tokens = torch.empty(R, n, dtype=torch.int32, device=self.device)
        lg = torch.empty(R, n, self.dims.num_tokens, dtype=torch.float32, device=self.device) \
            if return_logits else None
        ws, wsb = self.workspace(B, T, n_samples)
        L.check(self.lib.dimx_generate(self.h, L.ptr(start), L.ptr(mask_u8), B, T, int(n_samples), float(temperature),
                                       int(top_k),
                                       L.ptr(noise), int(seed) & 0xFFFFFFFFFFFFFFFF, L.ptr(tokens), L.ptr(lg), ws,
                                       wsb, self._s()), "dimx_generate")
        return (tokens, lg) if return_logits else tokens

    def chain_faults(self):
        """generate() calls of this handle whose XCD-local chain kernels reported a placement / barrier fault; each was
        regenerated on the one-kernel-per-op step before generate() returned (0 = never happened)."""
        return int(self.lib.dimx_chain_faults(self.h))

    def debug_chain_fault(self, n_calls=1):
        """test hook: the next n_calls generate() calls run their chain kernels on a non-bijective placement."""
        L.check(self.lib.dimx_debug_chain_fault(self.h, int(n_calls)), "dimx_debug_chain_fault")


# ---------------------------------------------------------------------- kernel-level wrappers (tests)
def _pad_k(w, mult):
    N, K = w.shape
    Kp = (K + mult - 1) // mult * mult
    if Kp == K:
        return w.contiguous()
    out = torch.zeros(N, Kp, dtype=w.dtype, device=w.device)
    out[:, :K] = w
    return out


def tile_weight(w_, bk):
    """[N, Kp] -> blocks of 8 rows x bk elements (128 B), block (n // 8, k // bk) contiguous: the layout the decode-step
    GEMM reads with one contiguous KiB per LDS-DMA piece (GemmArgs.w_tiled)."""
    N, Kp = w_.shape
    assert N % 8 == 0 and Kp % bk == 0
    return w_.view(N // 8, 8, Kp // bk, bk).permute(0, 2, 1, 3).contiguous().view(N, Kp)


def op_split_x3(w):
    """f32 [..] -> the three bf16 planes [3, ..] whose sum is ``w`` exactly (csrc/gemm_x3.hip split_x3_kernel)."""
    lib = L.load()
    w_ = w.float().contiguous()
    planes = torch.empty((3,) + tuple(w_.shape), dtype=torch.bfloat16, device=w.device)
    L.check(lib.dimx_op_split_x3(L.ptr(w_), L.ptr(planes), w_.numel(), L.stream_ptr(w.device)), "dimx_op_split_x3")
    return planes


def op_gemm_x3(a, w, bias=None, act=0, residual=None, slabs=False, planes=None):
    """The f32 parity mode's split-bf16 decode GEMM (csrc/gemm_x3.hip): epilogue(a[M,K] @ w[N,K]^T), f32 in and out,
    K % 32 == 0.  ``slabs``: the split-K partial sums [splits, M, N] as the decode step's consumers get them (the kernel plans
    the count from (N, K)).  ``planes``: the pre-split weight (op_split_x3), as the model keeps it."""
    lib = L.load()
    a_ = a.float().contiguous()
    if planes is None:
        planes = op_split_x3(w)
    (M, K), N = a_.shape, planes.shape[1]
    flags = 4 if slabs else 0
    ns = lib.dimx_op_gemm_slabs(L.F32, M, N, K, 16 | 5) if slabs else 0
    out = torch.full((ns, M, N) if slabs else (M, N), float("nan"), dtype=torch.float32, device=a.device)
    L.check(lib.dimx_op_gemm_x3(L.ptr(a_), K, L.ptr(planes), L.ptr(out), N, M, N, K, L.ptr(bias), act, L.ptr(residual),
                                residual.shape[1] if residual is not None else 0, flags, L.stream_ptr(a.device)), "dimx_op_gemm_x3")
    return out


def op_gemm(a, w, bias=None, act=0, residual=None, bf16=False, out_bf16=False, conv_T=0, conv_lens=None,
            slabs=0, force_simple=False, cfg=0, w_tiled=False):
    """epilogue(a[M,K] @ w[N,K]^T); conv_T > 0: a is [B*conv_T, C], w is [N, C, 5]."""
    lib = L.load()
    dev = a.device
    if conv_T > 0:
        N, C, _ = w.shape
        wk = w.permute(0, 2, 1).reshape(N, 5 * C)
        K = 5 * C
    else:
        wk, K = w, w.shape[1]
        N = w.shape[0]
    dt = torch.bfloat16 if bf16 else torch.float32
    a_ = a.to(dt).contiguous()
    w_ = _pad_k(wk.to(dt), 64 if bf16 else 32)
    if w_tiled:
        w_ = tile_weight(w_, 64 if bf16 else 32)
    M = a_.shape[0]
    if slabs:   # split-K partial sums as `slabs` f32 slabs [slabs, M, N] (the decode-step projections)
        out = torch.full((slabs, M, N), float("nan"), dtype=torch.float32, device=dev)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dev)
    flags = (5 if slabs else 0) | (2 if force_simple else 0) | (8 if w_tiled else 0) | (cfg << 8) | (slabs << 16)
    L.check(lib.dimx_op_gemm(L.BF16 if bf16 else L.F32, L.BF16 if out_bf16 else L.F32, L.ptr(a_), a_.shape[1],
                             L.ptr(w_), w_.shape[1], L.ptr(out), N, M, N, K, L.ptr(bias), act, L.ptr(residual),
                             residual.shape[1] if residual is not None else 0, conv_T, L.ptr(conv_lens), flags,
                             L.stream_ptr(dev)), "dimx_op_gemm")
    return out


def op_layernorm(x, gamma, beta=None, out_bf16=False):
    lib = L.load()
    M, C = x.shape
    y = torch.empty(M, C, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    L.check(lib.dimx_op_layernorm(L.BF16 if out_bf16 else L.F32, L.ptr(x.contiguous()), L.ptr(y), L.ptr(gamma),
                                  L.ptr(beta), M, C, L.stream_ptr(x.device)), "dimx_op_layernorm")
    return y


def op_instnorm(x, lens=None, out_bf16=False):
    lib = L.load()
    B, T, C = x.shape
    y = torch.empty(B, T, C, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    L.check(lib.dimx_op_instnorm(L.BF16 if out_bf16 else L.F32, L.ptr(x.contiguous()), L.ptr(y), L.ptr(lens), B, T,
                                 C, L.stream_ptr(x.device)), "dimx_op_instnorm")
    return y


def op_attention(q, k, v, scale, causal=False, lens=None, kmask=None, bf16=False, row_v=False):
    """q [B,Lq,H,D], k/v [B,Lk,H,D] -> [B,Lq,H,D]; v is transposed to [B,H,D,Lk_pad] here, or (row_v, bf16) handed over
    row-major like k."""
    lib = L.load()
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    if row_v:
        q_, k_, v_ = (t.to(torch.bfloat16).reshape(B, -1, H * D).contiguous() for t in (q, k, v))
        out = torch.empty(B, Lq, H * D, dtype=torch.bfloat16, device=q.device)
        L.check(lib.dimx_op_attention_rowv(L.ptr(q_), L.ptr(k_), L.ptr(v_), L.ptr(out), B, H, Lq, Lk, D, H * D, H * D, H * D, H * D,
                                           float(scale), 1 if causal else 0, L.ptr(lens), L.ptr(kmask), L.stream_ptr(q.device)),
                "dimx_op_attention_rowv")
        return out.view(B, Lq, H, D)
    dt = torch.bfloat16 if bf16 else torch.float32
    Lp = (Lk + 7) // 8 * 8
    q_ = q.to(dt).reshape(B, Lq, H * D).contiguous()
    k_ = k.to(dt).reshape(B, Lk, H * D).contiguous()
    vt = torch.full((B, H, D, Lp), float("nan"), dtype=dt, device=q.device)   # padding must be ignored
    vt[..., :Lk] = v.to(dt).permute(0, 2, 3, 1)
    out = torch.empty(B, Lq, H * D, dtype=dt, device=q.device)
    L.check(lib.dimx_op_attention(L.BF16 if bf16 else L.F32, L.ptr(q_), L.ptr(k_), L.ptr(vt), L.ptr(out), B, H, Lq,
                                  Lk, D, H * D, H * D, Lp, H * D, float(scale), 1 if causal else 0, L.ptr(lens),
                                  L.ptr(kmask), L.stream_ptr(q.device)), "dimx_op_attention")
    return out.view(B, Lq, H, D)


def op_train_attention(q, k, v, scale, d_o=None, causal=False, kmask=None, kmask2=None, mfma=False):
    """The training step's attention operator alone: q [B,Lq,H*64], k / v [B,Lk,H*64] f32 -> (o, lse) and, with d_o,
    (o, lse, dq, dk, dv).  mfma: 0 / False = the f32 VALU kernels, 1 / True = the bf16 matrix-core kernels of the perf mode,
    2 = the exact-f32 matrix-core kernels of the parity mode."""
    lib = L.load()
    B, Lq, C = q.shape
    Lk, H = k.shape[1], C // 64
    f = lambda t: None if t is None else t.float().contiguous()
    u8 = lambda t: None if t is None else t.to(torch.uint8).contiguous()
    q, k, v, d_o, kmask, kmask2 = f(q), f(k), f(v), f(d_o), u8(kmask), u8(kmask2)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
    delta = dq = dk = dv = None
    if d_o is not None:
        delta, dq, dk, dv = torch.empty_like(lse), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    L.check(lib.dimx_op_train_attention(int(mfma), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(d_o), L.ptr(kmask), L.ptr(kmask2),
                                        B, H, Lq, Lk, 1 if causal else 0, float(scale), L.ptr(o), L.ptr(lse), L.ptr(delta),
                                        L.ptr(dq), L.ptr(dk), L.ptr(dv), L.stream_ptr(q.device)), "dimx_op_train_attention")
    return (o, lse) if d_o is None else (o, lse, dq, dk, dv)


def op_sample(logits, top_k=52, temperature=1.0, noise=None, seed=0, step=0):
    lib = L.load()
    R = logits.shape[0]
    tok = torch.empty(R, dtype=torch.int32, device=logits.device)
    L.check(lib.dimx_op_sample(L.ptr(logits.contiguous()), R, top_k, float(temperature), L.ptr(noise), int(seed),
                               int(step), L.ptr(tok), L.stream_ptr(logits.device)), "dimx_op_sample")
    return tok


def op_chain(x, gamma, a1=None, w1=None, slabs=None, w2=None):
    """One XCD-local chain launch (csrc/chain.hip): x [B,C] f32 is updated in place; returns (y bf16 [B,C],
    out2 f32 [B,N2] or None).  a1 [B,K1] / w1 [C,K1] / w2 [N2,C] are converted to bf16."""
    lib = L.load()
    dev = x.device
    B, C = x.shape
    bf = lambda t: None if t is None else t.to(torch.bfloat16).contiguous()
    a1, w1, w2 = bf(a1), bf(w1), bf(w2)
    y = torch.empty(B, C, dtype=torch.bfloat16, device=dev)
    out2 = torch.empty(B, w2.shape[0], dtype=torch.float32, device=dev) if w2 is not None else None
    scratch = torch.zeros(512 + B * C, dtype=torch.int32, device=dev)
    nslab = 0 if slabs is None else slabs.shape[0]
    L.check(lib.dimx_op_chain(L.ptr(a1), a1.shape[1] if a1 is not None else 0, L.ptr(w1), L.ptr(x),
                              L.ptr(slabs.contiguous()) if slabs is not None else None, nslab, L.ptr(gamma), L.ptr(y),
                              L.ptr(w2), w2.shape[0] if w2 is not None else 0, L.ptr(out2), B, C, L.ptr(scratch),
                              L.stream_ptr(dev)), "dimx_op_chain")
    torch.cuda.synchronize(dev)
    flags = int(scratch[129].item())
    if flags:
        raise L.DimxError("chain kernel error flags 0x%x (1 = (XCD, slot) claimed twice, 2 = group barrier timeout)" % flags)
    return y, out2


def op_chain_ln(x, a1, w1, w2s=None, colsum2=None, return_flags=False):
    """Deferred-LayerNorm chain launch: x [B,C] f32 += a1 . w1^T in place; returns (y = bf16(x) [B,C], stats [8,32,32,2],
    out2 [B,N2] or None) with out2 = LayerNorm(x) * gamma . W2^T when w2s = gamma o W2 (bf16) and colsum2 = w2s row sums.
    stats[g, c, r] = {sum, centred sum of squares} of CU c's column slice of row 32 g + r.  return_flags: also the kernel's
    flag word (bit 2 = precision guard: some row's |mean| > 8 std)."""
    lib = L.load()
    dev = x.device
    B, C = x.shape
    a1, w1 = a1.to(torch.bfloat16).contiguous(), w1.to(torch.bfloat16).contiguous()
    y = torch.empty(B, C, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(8, 32, 32, 2, dtype=torch.float32, device=dev)
    out2 = torch.empty(B, w2s.shape[0], dtype=torch.float32, device=dev) if w2s is not None else None
    scratch = torch.zeros(512 + B * C, dtype=torch.int32, device=dev)
    L.check(lib.dimx_op_chain_ln(L.ptr(a1), a1.shape[1], L.ptr(w1), L.ptr(x), L.ptr(y), L.ptr(stats), L.ptr(w2s),
                                 L.ptr(colsum2), w2s.shape[0] if w2s is not None else 0, L.ptr(out2), B, C, L.ptr(scratch),
                                 L.stream_ptr(dev)), "dimx_op_chain_ln")
    torch.cuda.synchronize(dev)
    flags = int(scratch[129].item())
    if flags & 3:
        raise L.DimxError("chain kernel error flags 0x%x (1 = (XCD, slot) claimed twice, 2 = group barrier timeout)" % flags)
    return (y, stats, out2, flags) if return_flags else (y, stats, out2)


def op_gemm_ln(a, ws, stats, colsum, bias=None, act=0, out_bf16=False):
    """act(LayerNorm-corrected a . ws^T + bias): a = bf16(x) un-normalised, ws = gamma o W (bf16), stats from op_chain_ln."""
    lib = L.load()
    M, K = a.shape
    N = ws.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
    L.check(lib.dimx_op_gemm_ln(L.BF16 if out_bf16 else L.F32, L.ptr(a), L.ptr(ws), L.ptr(out), M, N, K, L.ptr(bias), act,
                                L.ptr(stats), L.ptr(colsum), L.stream_ptr(a.device)), "dimx_op_gemm_ln")
    return out


def op_decode_attn(q, kcache, vcache, n_keys, scale, kmask=None, nsplit=0):
    """q [B,H*64]; kcache/vcache [B,H,Tmax,64] (f32 or bf16) -> [B,H*64]."""
    lib = L.load()
    B, H, Tmax, _ = kcache.shape
    bf = kcache.dtype == torch.bfloat16
    q_f32 = q.dtype == torch.float32 and bf      # f32 projection slab feeding a bf16 cache (generate's form)
    out = torch.empty(q.shape, dtype=kcache.dtype, device=q.device)
    L.check(lib.dimx_op_decode_attn(L.BF16 if bf else L.F32, L.ptr(q), L.ptr(kcache), L.ptr(vcache), L.ptr(out), B, H,
                                    Tmax, n_keys, float(scale), L.ptr(kmask), nsplit, 1 if q_f32 else 0,
                                    L.stream_ptr(q.device)),
            "dimx_op_decode_attn")
    return out
