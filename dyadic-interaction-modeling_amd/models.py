"""Drop-in VQ-VAE surface: ``get_model(cfg)`` and ``VQAutoEncoder`` with the reference's constructor,
parameter names and method signatures (reference ``code/models/__init__.py:1-17``,
``code/models/stage1_BIWI.py:10-137``), computing on the HIP library.

The module owns ordinary ``nn.Parameter`` / buffer objects under the reference's key names, so
``state_dict()``, ``load_state_dict()``, ``.to()``, ``.eval()`` and ``.parameters()`` behave as usual;
the packed device copy inside the engine is refreshed lazily whenever the parameters changed.
Inference only: outputs carry no autograd graph (training is out of scope).
"""
import torch
import torch.nn as nn

from . import lib as L
from . import weights as W
from .engine import Engine


def build_param_tree(root: nn.Module, spec, tensors, buffers=("pe",)):
    """Register every (dotted) key of ``spec`` under ``root`` creating container modules on the way, so
    that ``root.state_dict()`` has exactly the reference's keys."""
    for name, shape, kind, _ in spec:
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        t = tensors[name].clone()
        assert tuple(t.shape) == tuple(shape), name
        if kind in buffers:
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


class _EngineOwner(nn.Module):
    """Keeps one Engine per (device, numeric mode) and re-uploads weights when they changed."""

    engine_variant = "slmft"     # which handle geometry the module computes on ("slmft" / "legacy")

    def __init__(self, numeric_mode):
        super().__init__()
        self.numeric_mode = numeric_mode
        self._engine = None
        self._engine_version = None
        self._engine_pin = 0          # > 0: inside a forward whose first engine() call already checked the weights

    # ------------------------------------------------------------------ checkpoints of other x-transformers releases
    _OPTIONAL_LINEAR_BIAS = ("project_in.bias", "to_logits.bias")

    def _adopt_optional_tensors(self, state_dict):
        """SURVEY A.2 marks three details of x-transformers 1.30.16 [XT?]: bias-free ``project_in`` / ``to_logits`` and a
        LayerNorm without bias.  A checkpoint written with another release carries those tensors, and the reference loads its
        checkpoints with ``strict=False`` (code/finetune_s2s_pretrain.py:57), which would drop them without a word.  Here:
        a ``project_in.bias`` / ``to_logits.bias`` next to a weight this module owns becomes a parameter (the engine applies
        it); a LayerNorm ``bias`` inside ``attn_layers`` -- the zero ``beta`` buffer that loader renames -- is dropped when it
        is zero and refused otherwise.  Returns the state dict to hand to ``nn.Module.load_state_dict``."""
        own = self.state_dict(keep_vars=True)
        out = {}
        for k, v in state_dict.items():
            if k in own or not k.endswith(".bias") or (k[:-4] + "weight") not in own:
                out[k] = v
                continue
            w = own[k[:-4] + "weight"]
            if k.endswith(self._OPTIONAL_LINEAR_BIAS):
                if tuple(v.shape) != (w.shape[0],):
                    raise L.DimxError("%s: shape %s, expected (%d,)" % (k, tuple(v.shape), w.shape[0]))
                mod = self
                for part in k.split(".")[:-1]:
                    mod = mod._modules[part]
                mod.register_parameter("bias", torch.nn.Parameter(torch.zeros_like(w[:, 0]), requires_grad=False))
                out[k] = v
            elif ".attn_layers." in k and w.dim() == 1:
                if bool((v != 0).any()):
                    raise L.DimxError("%s: a non-zero LayerNorm bias -- this checkpoint was written by an x-transformers "
                                      "variant the path does not implement" % k)
            else:
                out[k] = v
        # a bias adopted from an earlier checkpoint does not survive a checkpoint without it
        for k in list(own):
            if k.endswith(self._OPTIONAL_LINEAR_BIAS) and k not in state_dict and (k[:-4] + "weight") in state_dict:
                mod = self
                for part in k.split(".")[:-1]:
                    mod = mod._modules[part]
                del mod._parameters["bias"]
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        return super().load_state_dict(self._adopt_optional_tensors(state_dict), strict=strict, **kw)

    def _weights_version(self):
        return tuple((k, v._version, v.data_ptr()) for k, v in self.state_dict(keep_vars=True).items())

    def _engine_state_dict(self):
        raise NotImplementedError

    def engine_pinned(self):
        """Context manager for a forward pass made of several engine() calls (forward_vq -> forward_decoder ->
        forward_vq_decoder): the weights are compared with the packed copy ONCE, at the first call inside the block -- the
        comparison walks the state dict (505 tensors, ~1.5 ms), and repeated behind the generate call's host synchronisation it
        ran with the GPU idle.  Parameters must not be modified inside the block."""
        owner = self

        class _Pin:
            def __enter__(self_inner):
                owner._engine_pin += 1
                owner._engine_pin_checked = owner._engine_pin > 1 and getattr(owner, "_engine_pin_checked", False)
                return owner

            def __exit__(self_inner, *exc):
                owner._engine_pin -= 1
                if owner._engine_pin == 0:
                    owner._engine_pin_checked = False
                return False
        return _Pin()

    def engine(self, device=None):
        if self._engine_pin > 0 and getattr(self, "_engine_pin_checked", False) and self._engine is not None and (
                device is None or self._engine.device == torch.device(device)):
            return self._engine
        eng = self._engine_checked(device)
        if self._engine_pin > 0:
            self._engine_pin_checked = True
        return eng

    def _engine_checked(self, device=None):
        if device is None:
            p = next(self.parameters())
            device = p.device
        device = torch.device(device)
        if device.type != "cuda":
            raise L.DimxError("dimx modules compute on a ROCm GPU only: move the module with .to('cuda:0') "
                              "(there is no CPU path; the CPU oracle under oracle/ is test infrastructure)")
        ver = self._weights_version()
        if self._engine is None or self._engine.device != device:
            self._engine = Engine(device, self.numeric_mode, self.engine_variant)
            self._engine_version = None
        if self._engine_version != ver:
            self._engine.load_state_dict(self._engine_state_dict())
            self._engine_version = ver
        return self._engine


class VectorQuantizerView(nn.Module):
    """``model.quantize``: holds ``embedding.weight`` [512,128] like the reference VectorQuantizer."""

    def __init__(self, weight):
        super().__init__()
        self.embedding = nn.Embedding(weight.shape[0], weight.shape[1])
        self.embedding.weight = nn.Parameter(weight.clone(), requires_grad=False)
        self.n_e, self.e_dim, self.beta = weight.shape[0], weight.shape[1], 0.25

    def get_codebook_entry(self, indices, shape=None):
        z_q = self.embedding.weight[indices.long()]
        return z_q.view(shape) if shape is not None else z_q


class VQAutoEncoder(_EngineOwner):
    """reference code/models/stage1_BIWI.py:10-137 (encode / decode / forward / get_quant /
    decode_to_img / entry_to_feature)."""

    def __init__(self, args, synthetic_seed=20260928, weight_prefix="listener_vq.",
                 numeric_mode=L.MODE_PARITY_F32, which=1):
        super().__init__(numeric_mode)
        self.args = args
        self.dims = W.VQDims.from_cfg(args)
        self.which = which
        spec = W.vq_spec(self.dims, prefix=weight_prefix)
        sd = W.synth_state_dict(spec, synthetic_seed, strip_prefix=weight_prefix)
        local = [(n[len(weight_prefix):], s, k, f) for n, s, k, f in spec]
        q = [e for e in local if e[0].startswith("quantize.")]
        build_param_tree(self, [e for e in local if not e[0].startswith("quantize.")], sd)
        self.quantize = VectorQuantizerView(sd[q[0][0]])

    def _engine_state_dict(self):
        # a standalone VQ-VAE occupies one slot of the handle (listener by default); the library packs
        # components lazily, so nothing else has to be loaded
        pre = "listener_vq." if self.which == 1 else "speaker_vq."
        return {pre + k: v for k, v in self.state_dict().items()}

    @torch.no_grad()
    def encode(self, x, x_a=None):
        """x [B,L,56] -> (quant [B,128,L], emb_loss, (perplexity, one_hot [B*L,512], idx [B*L,1]))."""
        eng = self.engine(x.device)
        B, Lq, _ = x.shape
        idx, z = eng.vq_encode(self.which, x, None, pe_mode=1, return_z=True)
        idx = idx.long().view(-1)
        E = self.quantize.embedding.weight.to(x.device)
        z_q = E[idx].view(B, Lq, -1)
        loss = (1.0 + self.quantize.beta) * torch.mean((z_q - z) ** 2)
        onehot = torch.zeros(idx.shape[0], E.shape[0], device=x.device)
        onehot.scatter_(1, idx[:, None], 1)
        e_mean = onehot.mean(0)
        perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
        return z_q.permute(0, 2, 1).contiguous(), loss, (perplexity, onehot, idx[:, None])

    @torch.no_grad()
    def decode_indices(self, idx, batch_row_offset=0):
        """idx [B,L] -> [B,L,56] (codebook lookup fused into the decoder)."""
        return self.engine(idx.device).vq_decode(self.which, idx, batch_row_offset)

    @torch.no_grad()
    def decode(self, quant, batch_row_offset=0):
        """quant [B,128,L] -> [B,L,56] (reference :29-37): the decoder runs on the latents it is GIVEN -- codebook
        rows as ``encode`` returns them, or any other [B,128,L] tensor -- with no re-quantisation."""
        return self.engine(quant.device).vq_decode_latent(self.which, quant.permute(0, 2, 1).contiguous(),
                                                          batch_row_offset)

    def forward(self, x):
        quant, emb_loss, info = self.encode(x)
        return self.decode(quant), emb_loss, info

    def get_quant(self, x, x_a=None):
        quant_z, _, info = self.encode(x, x_a)
        return quant_z, info[2]

    @torch.no_grad()
    def entry_to_feature(self, index, zshape):
        return self.quantize.get_codebook_entry(index.reshape(-1)).reshape(zshape)

    @torch.no_grad()
    def decode_to_img(self, index, zshape):
        return self.decode_indices(index.long().reshape(zshape[0], zshape[1]))


def get_model(cfg, **kw):
    """reference code/models/__init__.py:1-17; only the architecture on the DIM-Listener path is built."""
    if cfg.arch == "stage1_BIWI":
        return VQAutoEncoder(cfg, **kw)
    raise Exception("architecture not supported yet: {} (only stage1_BIWI is on the DIM-Listener path)".format(cfg.arch))
