"""A STAND-IN for the `x_transformers` module, built on oracle/ref_cpu.py -- NOT the library and no evidence about it.

Its only purpose: let tests/test_xt_pin_plumbing.py EXECUTE tools/verify_against_xtransformers.py (the offline pin
path of DESIGN section 2) once in a container without the wheel, so that the writer's plumbing is not dead code: the
load_state_dict key filtering, the torch.randn capture of AutoregressiveWrapper's key-mask draw, the torch.multinomial
replacement inside generate, the npz keys, and the two consumer tests that read them.  Every number it produces comes
from the oracle itself, so a fixture written through it proves nothing and is never committed (the test writes to a
temporary directory and checks that tests/golden/ stays untouched).  It mimics the call surface the script uses:
ContinuousTransformerWrapper / TransformerWrapper / AutoregressiveWrapper / Encoder / Decoder of x-transformers 1.30.16
(reference call sites code/seq2seq_pretrain.py:388-419,439-450, code/seq2seq.py:28-45,54)."""
import collections

import torch
import torch.nn.functional as F

from oracle import ref_cpu

__version__ = "fake-from-oracle"
_Keys = collections.namedtuple("_Keys", ["missing_keys", "unexpected_keys"])


class _Layers:
    def __init__(self, dim, depth, heads, cross_attend=False, **_ignored):
        self.dim, self.depth, self.heads, self.cross_attend = dim, depth, heads, cross_attend


class Encoder(_Layers):
    pass


class Decoder(_Layers):
    pass


class _Holder(torch.nn.Module):
    """keeps the tensors it is given under the library's key names; reports keys the way nn.Module does"""
    expected_missing = ()

    def load_state_dict(self, sd, strict=True):
        self._sd = {k: v for k, v in sd.items()}
        return _Keys(list(self.expected_missing), [])

    def _prefixed(self, prefix):
        return {prefix + k: v for k, v in self._sd.items()}


class ContinuousTransformerWrapper(_Holder):
    expected_missing = ("project_out.weight",)     # exists in the library, skipped by return_embeddings=True

    def __init__(self, dim_in, dim_out, max_seq_len, attn_layers):
        super().__init__()
        self.layers = attn_layers

    def forward(self, x, mask=None, attn_mask=None, return_embeddings=False):
        assert return_embeddings
        return ref_cpu.xt_encoder(self._prefixed("m."), "m.", x, mask, causal=attn_mask is not None,
                                  depth=self.layers.depth, heads=self.layers.heads)


class TransformerWrapper(_Holder):
    def __init__(self, num_tokens, max_seq_len, attn_layers, use_abs_pos_emb=True, emb_dropout=0):
        super().__init__()
        self.layers, self.abs_pos = attn_layers, use_abs_pos_emb


class AutoregressiveWrapper(_Holder):
    def __init__(self, net, ignore_index=-100, pad_value=0, mask_prob=0.0):
        super().__init__()
        self.net, self.ignore_index, self.pad_value, self.mask_prob = net, ignore_index, pad_value, mask_prob

    def _sd_for_oracle(self):
        pre = "generator.decoder." if self.net.abs_pos else "decoder_joint."
        return self._prefixed(pre)

    def forward(self, z, context=None, context_mask=None, return_outputs=False):
        sd = self._sd_for_oracle()
        B, T = z.shape
        kv = None
        if self.mask_prob > 0:      # the draw the script's _Capture records
            rand = torch.randn((B, T - 1))
            rand[:, 0] = -torch.finfo(rand.dtype).max
            num_mask = min(int(T * self.mask_prob), T - 1)
            kv = ~torch.zeros(B, T - 1).scatter(1, rand.topk(num_mask, dim=-1).indices, 1.0).bool()
        if self.net.abs_pos:
            inp, target = z[:, :-1].clamp(min=0), z[:, 1:]
            logits = ref_cpu.legacy_decoder_logits(sd, inp, context, context_mask, depth=self.net.layers.depth,
                                                   heads=self.net.layers.heads)
            loss = F.cross_entropy(logits.permute(0, 2, 1), target, ignore_index=self.ignore_index)
        else:
            loss, logits = ref_cpu.ar_forward(sd, z, context, context_mask, kv, self.ignore_index, self.pad_value)
        return (loss, (logits, None)) if return_outputs else loss

    @torch.no_grad()
    def generate(self, prompts, seq_len, temperature=1.0, context=None, context_mask=None):
        sd = self._sd_for_oracle()

        def sampler(logits, noise=None, temp=1.0, k=52):     # the call the script's _Capture replaces
            if temperature == 0.0:
                return logits.argmax(dim=-1)
            probs = F.softmax(ref_cpu.top_k_filter(logits, k) / temperature, dim=-1)
            return torch.multinomial(probs, 1)[:, 0]
        keep = ref_cpu.sample_tokens
        ref_cpu.sample_tokens = sampler
        try:
            if self.net.abs_pos:
                return ref_cpu.legacy_generate(sd, prompts[:, 0], seq_len, context, context_mask, None,
                                               depth=self.net.layers.depth, heads=self.net.layers.heads)
            return ref_cpu.ar_generate(sd, prompts[:, 0], seq_len, context, context_mask, None)
        finally:
            ref_cpu.sample_tokens = keep
