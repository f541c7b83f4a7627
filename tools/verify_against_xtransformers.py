"""Optional, offline: compare the x-transformers half of the CPU oracle (oracle/ref_cpu.py) with the REAL
library, for anyone who has the wheel (`pip install x-transformers==1.30.16`).  Not used by tests, smoke or
bench (the library is unavailable in the build container and on the GPU box) -- this is the script that would
turn the "parity unpinned" note of oracle/ref_cpu.py into a pinned one.

    python tools/verify_against_xtransformers.py
"""
import sys

import torch

sys.path.insert(0, ".")
import dimx  # noqa: E402
from dimx import prng, weights  # noqa: E402
from oracle import ref_cpu  # noqa: E402


def main():
    try:
        from x_transformers import (AutoregressiveWrapper, ContinuousTransformerWrapper, Decoder, Encoder,
                                    TransformerWrapper)
    except ImportError:
        print("x-transformers is not installed: nothing to verify (the oracle stays 'parity unpinned').")
        return 0
    torch.set_grad_enabled(False)
    sd = weights.synth_state_dict(weights.slmft_spec(), 20260928)
    enc_s = ContinuousTransformerWrapper(dim_in=56, dim_out=384, max_seq_len=2048,
                                         attn_layers=Encoder(dim=384, depth=4, heads=12))
    enc_j = ContinuousTransformerWrapper(dim_in=384, dim_out=384, max_seq_len=2048,
                                         attn_layers=Encoder(dim=384, depth=4, heads=12))
    dec = AutoregressiveWrapper(TransformerWrapper(num_tokens=512, max_seq_len=2048, use_abs_pos_emb=False,
                                                   emb_dropout=0, attn_layers=Decoder(dim=1152, depth=4, heads=12,
                                                                                      cross_attend=True)),
                                ignore_index=-100, pad_value=0, mask_prob=0.15)
    for mod, pre in ((enc_s, "encoder_s."), (enc_j, "encoder_joint."), (dec, "decoder_joint.")):
        own = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        missing, unexpected = mod.load_state_dict(own, strict=False)
        print(pre, "missing", list(missing)[:4], "unexpected", list(unexpected)[:4])
    B, T = 2, 24
    v_s = torch.from_numpy(prng.normal(1, "xt.vs", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(1, "xt.va", (B, T, 768)))
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[1, 17:] = False
    attn_mask = ~torch.triu(torch.ones(T, T), diagonal=1).bool()
    x = enc_s.eval()(v_s + sd["patch_embed_s"], mask=mask, attn_mask=attn_mask, return_embeddings=True)
    x = enc_j.eval()(x, mask=mask, attn_mask=attn_mask, return_embeddings=True)
    x = torch.nn.functional.layer_norm(x, (384,), sd["norm_s.weight"], sd["norm_s.bias"])
    ref = ref_cpu.slmft_forward_encoder(sd, v_s, mask)
    e = max((x[b, :n] - ref[b, :n]).abs().max().item() for b, n in ((0, 24), (1, 17)))
    print("encoder stack: max |library - oracle| on valid rows = %.3g" % e)
    ctx = ref_cpu.slmft_context(sd, ref, v_a)
    z = torch.from_numpy(prng.integers(1, "xt.z", (B, T), 0, 512))
    logits_lib = dec.net.eval()(z[:, :-1], context=ctx, context_mask=mask)
    logits_or = ref_cpu.xt_decoder_logits(sd, z[:, :-1], ctx, mask, None)
    print("decoder logits: max |library - oracle| = %.3g" % (logits_lib - logits_or).abs().max().item())
    return 0


if __name__ == "__main__":
    sys.exit(main())
