"""Offline pin of the x-transformers half of the CPU oracle (oracle/ref_cpu.py) against the REAL library, for anyone
who has the wheel the reference pins (`pip install x-transformers==1.30.16`, code/requirements.txt:99).  The library
is not installable in the build container nor on the GPU box, so this script could not be executed there; what it
writes is what would turn the "parity unpinned" note of oracle/ref_cpu.py into a pinned one:

    python tools/verify_against_xtransformers.py                 # print max |library - oracle| per stage
    python tools/verify_against_xtransformers.py --write-golden  # + tests/golden/xt_slmft.npz, xt_legacy.npz

The fixtures are consumed by tests/test_oracle_xt_golden.py (oracle vs library) and tests/test_gpu_xt_golden.py (HIP
path vs library); both skip while the files are absent.  Covered: the two SLMFT encoder stacks (causal attn_mask +
padding mask), AutoregressiveWrapper.forward incl. its mask_prob=0.15 self-attention key mask and loss, generate
(greedy and multinomial with injected Exp(1) noise), and the legacy ListenerGenerator encoder / decoder (absolute
positional embedding, no causal mask in the encoder).  Randomness is captured, not replayed: torch.randn (key mask) is
wrapped to record what the library drew, torch.multinomial is replaced by argmax(p / q) with q from dimx.prng
(tests/golden/sampler_multinomial.npz shows the two are the same function of (p, q)).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dimx  # noqa: E402,F401
from dimx import prng, weights  # noqa: E402
from oracle import ref_cpu  # noqa: E402

SEED = 20260928
GOLD = os.path.join(ROOT, "tests", "golden")


class _Capture:
    """Record torch.randn draws and replace torch.multinomial by argmax(p / q) with injected noise."""

    def __init__(self, noise=None):
        self.noise, self.step, self.randn = noise, 0, []

    def __enter__(self):
        self._randn, self._mult = torch.randn, torch.multinomial

        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.randn.append(t.clone())
            return t

        def multinomial(p, n, *a, **k):
            assert n == 1 and self.noise is not None
            q = self.noise[self.step]
            self.step += 1
            return (p / q).argmax(-1, keepdim=True)
        torch.randn, torch.multinomial = randn, multinomial
        return self

    def __exit__(self, *exc):
        torch.randn, torch.multinomial = self._randn, self._mult


def _load(mod, sd, pre):
    own = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    missing, unexpected = mod.load_state_dict(own, strict=False)
    # project_out exists in the library but is skipped by return_embeddings=True (never in the oracle's spec)
    assert not [k for k in missing if "project_out" not in k], (pre, missing)
    assert not unexpected, (pre, unexpected)
    return mod.eval()


def slmft_case(xt, write):
    sd = weights.synth_state_dict(weights.slmft_spec(), SEED)
    enc = lambda din: xt.ContinuousTransformerWrapper(dim_in=din, dim_out=384, max_seq_len=2048,
                                                      attn_layers=xt.Encoder(dim=384, depth=4, heads=12))
    enc_s, enc_j = _load(enc(56), sd, "encoder_s."), _load(enc(384), sd, "encoder_joint.")
    dec = xt.AutoregressiveWrapper(
        xt.TransformerWrapper(num_tokens=512, max_seq_len=2048, use_abs_pos_emb=False, emb_dropout=0,
                              attn_layers=xt.Decoder(dim=1152, depth=4, heads=12, cross_attend=True)),
        ignore_index=-100, pad_value=0, mask_prob=0.15)
    _load(dec, sd, "decoder_joint.")
    B, T, lens = 3, 40, (40, 33, 7)
    v_s = torch.from_numpy(prng.normal(1, "xt.vs", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(1, "xt.va", (B, T, 768)))
    z = torch.from_numpy(prng.integers(1, "xt.z", (B, T), 0, 512))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    z = torch.where(mask, z, torch.full_like(z, -100))
    attn_mask = torch.tril(torch.ones(T, T)).bool()                       # code/seq2seq_pretrain.py:435
    x = enc_s(v_s + sd["patch_embed_s"], mask=mask, attn_mask=attn_mask, return_embeddings=True)
    x = enc_j(x, mask=mask, attn_mask=attn_mask, return_embeddings=True)
    x_s = torch.nn.functional.layer_norm(x, (384,), sd["norm_s.weight"], sd["norm_s.bias"])
    o_xs = ref_cpu.slmft_forward_encoder(sd, v_s, mask)
    e_enc = max((x_s[b, :n] - o_xs[b, :n]).abs().max().item() for b, n in enumerate(lens))
    ctx = torch.cat([x_s + sd["patch_embed_dec_s"], v_a], -1)             # :445-446
    torch.manual_seed(11)
    with _Capture() as cap:
        loss, (logits, _) = dec(z, context=ctx, context_mask=mask, return_outputs=True)
    rand = cap.randn[0]
    rand[:, 0] = -torch.finfo(rand.dtype).max
    num_mask = min(int(T * 0.15), T - 1)
    kv_mask = ~torch.zeros(B, T - 1).scatter(1, rand.topk(num_mask, dim=-1).indices, 1.0).bool()
    o_loss, o_logits = ref_cpu.ar_forward(sd, z, ref_cpu.slmft_context(sd, o_xs, v_a), mask, kv_mask)
    valid = mask[:, 1:]
    e_tf = (logits - o_logits).abs()[valid].max().item()
    noise = torch.from_numpy(prng.exponential(2, "xt.noise", (T - 1, B, 512)))
    start = z[:, :1].clamp(min=0)
    greedy = dec.generate(start, T - 1, temperature=0.0, context=ctx, context_mask=mask)
    with _Capture(noise) as cap:
        sampled = dec.generate(start, T - 1, context=ctx, context_mask=mask)
    octx = ref_cpu.slmft_context(sd, o_xs, v_a)
    o_greedy = ref_cpu.ar_generate(sd, start[:, 0], T - 1, octx, mask, None)
    o_sampled = ref_cpu.ar_generate(sd, start[:, 0], T - 1, octx, mask, noise)
    print("SLMFT  encoder stacks      max |library - oracle| = %.3g" % e_enc)
    print("SLMFT  teacher-forced      max |logits diff| = %.3g, loss %.6f vs %.6f" % (e_tf, loss.item(), o_loss.item()))
    print("SLMFT  generate            greedy equal: %s, injected-noise equal: %s" % (
        torch.equal(greedy, o_greedy), torch.equal(sampled, o_sampled)))
    if write:
        np.savez_compressed(os.path.join(GOLD, "xt_slmft.npz"), B=B, T=T, lens=np.array(lens), x_s=x_s.numpy(),
                            kv_mask=kv_mask.numpy(), tf_logits=logits.numpy(), tf_loss=loss.item(),
                            gen_greedy=greedy.numpy(), gen_sampled=sampled.numpy(), xt_version=xt.__version__
                            if hasattr(xt, "__version__") else "1.30.16")


def legacy_case(xt, write):
    sd = weights.synth_state_dict(weights.legacy_generator_spec(), SEED)
    enc = _load(xt.ContinuousTransformerWrapper(dim_in=1024, dim_out=512, max_seq_len=1024,
                                                attn_layers=xt.Encoder(dim=512, depth=6, heads=8)), sd, "generator.encoder.")
    dec = xt.AutoregressiveWrapper(
        xt.TransformerWrapper(num_tokens=512, max_seq_len=1024, use_abs_pos_emb=True,
                              attn_layers=xt.Decoder(dim=512, depth=6, heads=8, cross_attend=True)),
        ignore_index=-100, pad_value=0)
    _load(dec, sd, "generator.decoder.")
    B, T, lens = 2, 32, (32, 21)
    xsp = torch.from_numpy(prng.normal(3, "xtl.x", (B, T, 1024)))
    z = torch.from_numpy(prng.integers(3, "xtl.z", (B, T), 0, 512))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    z = torch.where(mask, z, torch.full_like(z, -100))
    ctx = enc(xsp, mask=mask, return_embeddings=True)                     # code/seq2seq.py:54
    o_ctx = ref_cpu.xt_encoder(sd, "generator.encoder.", xsp, mask, causal=False, depth=6, heads=8)
    e_enc = max((ctx[b, :n] - o_ctx[b, :n]).abs().max().item() for b, n in enumerate(lens))
    loss, (logits, _) = dec(z, context=ctx, context_mask=mask, return_outputs=True)
    o_logits = ref_cpu.legacy_decoder_logits(sd, z[:, :-1].clamp(min=0), o_ctx, mask)
    e_tf = (logits - o_logits).abs()[mask[:, 1:]].max().item()
    noise = torch.from_numpy(prng.exponential(4, "xtl.noise", (T, B, 512)))
    start = z[:, :1].clamp(min=0)
    with _Capture(noise):
        sampled = dec.generate(start, T, context=ctx, context_mask=mask)   # code/seq2seq.py:300: seq_len = T
    o_sampled = ref_cpu.legacy_generate(sd, start[:, 0], T, o_ctx, mask, noise)
    print("legacy encoder             max |library - oracle| = %.3g" % e_enc)
    print("legacy teacher-forced      max |logits diff| = %.3g" % e_tf)
    print("legacy generate            injected-noise equal: %s" % torch.equal(sampled, o_sampled))
    if write:
        np.savez_compressed(os.path.join(GOLD, "xt_legacy.npz"), B=B, T=T, lens=np.array(lens), enc_out=ctx.numpy(),
                            tf_logits=logits.numpy(), tf_loss=loss.item(), gen_sampled=sampled.numpy())


def main():
    try:
        import x_transformers as xt
    except ImportError:
        print("x-transformers is not installed: nothing to verify (the oracle stays 'parity unpinned').")
        return 0
    torch.set_grad_enabled(False)
    write = "--write-golden" in sys.argv
    slmft_case(xt, write)
    legacy_case(xt, write)
    return 0


if __name__ == "__main__":
    sys.exit(main())
