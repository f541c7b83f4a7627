"""CPU: the x-transformers half of the oracle against fixtures written from the REAL library
(tools/verify_against_xtransformers.py --write-golden, needs x-transformers==1.30.16).  The wheel is not available
in the build container, so the fixtures may be absent: the tests skip, and the oracle's x-transformers half stays
"parity unpinned"  until someone with the wheel commits them."""
import os

import numpy as np
import pytest
import torch

torch.set_grad_enabled(False)


def _gold(golden_dir, name):
    p = os.path.join(golden_dir, name)
    if not os.path.exists(p):
        pytest.skip("%s not present (run tools/verify_against_xtransformers.py --write-golden where the wheel exists)" % name)
    return np.load(p)


def _slmft_inputs(g):
    from dimx import prng
    B, T, lens = int(g["B"]), int(g["T"]), [int(n) for n in g["lens"]]
    v_s = torch.from_numpy(prng.normal(1, "xt.vs", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(1, "xt.va", (B, T, 768)))
    z = torch.from_numpy(prng.integers(1, "xt.z", (B, T), 0, 512))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    z = torch.where(mask, z, torch.full_like(z, -100))
    noise = torch.from_numpy(prng.exponential(2, "xt.noise", (T - 1, B, 512)))
    return v_s, v_a, z, mask, noise, lens


def test_oracle_slmft_stages_match_the_library(golden_dir, full_sd):
    from oracle import ref_cpu
    g = _gold(golden_dir, "xt_slmft.npz")
    v_s, v_a, z, mask, noise, lens = _slmft_inputs(g)
    x_s = ref_cpu.slmft_forward_encoder(full_sd, v_s, mask)
    for b, n in enumerate(lens):
        assert np.abs(x_s[b, :n].numpy() - g["x_s"][b, :n]).max() < 1e-4
    ctx = ref_cpu.slmft_context(full_sd, x_s, v_a)
    loss, logits = ref_cpu.ar_forward(full_sd, z, ctx, mask, torch.from_numpy(g["kv_mask"]))
    valid = mask[:, 1:].numpy()
    assert np.abs(logits.numpy() - g["tf_logits"])[valid].max() < 1e-3
    assert abs(loss.item() - float(g["tf_loss"])) < 1e-4 * max(1.0, abs(float(g["tf_loss"])))
    start = z[:, 0].clamp(min=0)
    assert np.array_equal(ref_cpu.ar_generate(full_sd, start, z.shape[1] - 1, ctx, mask, None).numpy(), g["gen_greedy"])
    assert np.array_equal(ref_cpu.ar_generate(full_sd, start, z.shape[1] - 1, ctx, mask, noise).numpy(), g["gen_sampled"])


def test_oracle_legacy_stages_match_the_library(golden_dir):
    from dimx import prng, weights
    from oracle import ref_cpu
    g = _gold(golden_dir, "xt_legacy.npz")
    sd = weights.synth_state_dict(weights.legacy_generator_spec(), 20260928)
    B, T, lens = int(g["B"]), int(g["T"]), [int(n) for n in g["lens"]]
    xsp = torch.from_numpy(prng.normal(3, "xtl.x", (B, T, 1024)))
    z = torch.from_numpy(prng.integers(3, "xtl.z", (B, T), 0, 512))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    z = torch.where(mask, z, torch.full_like(z, -100))
    ctx = ref_cpu.xt_encoder(sd, "generator.encoder.", xsp, mask, causal=False, depth=6, heads=8)
    for b, n in enumerate(lens):
        assert np.abs(ctx[b, :n].numpy() - g["enc_out"][b, :n]).max() < 2e-4
    logits = ref_cpu.legacy_decoder_logits(sd, z[:, :-1].clamp(min=0), ctx, mask)
    assert np.abs(logits.numpy() - g["tf_logits"])[mask[:, 1:].numpy()].max() < 2e-3
    noise = torch.from_numpy(prng.exponential(4, "xtl.noise", (T, B, 512)))
    assert np.array_equal(ref_cpu.legacy_generate(sd, z[:, 0].clamp(min=0), T, ctx, mask, noise).numpy(), g["gen_sampled"])
