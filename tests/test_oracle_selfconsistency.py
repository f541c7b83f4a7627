"""CPU: self-consistency properties that pin the x-transformers half of the oracle (the library itself is
unavailable: see the 'parity unpinned' note in oracle/ref_cpu.py)."""
import pytest
import torch

from oracle import ref_cpu

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def setup(full_sd):
    from dimx import prng
    B, T = 2, 14
    v_s = torch.from_numpy(prng.normal(1, "sc.vs", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(1, "sc.va", (B, T, 768)))
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[1, 9:] = False
    x_s = ref_cpu.slmft_forward_encoder(full_sd, v_s, mask)
    ctx = ref_cpu.slmft_context(full_sd, x_s, v_a)
    return full_sd, v_s, v_a, mask, ctx


def test_cached_equals_uncached_equals_teacher_forced(setup):
    sd, v_s, v_a, mask, ctx = setup
    B, T = mask.shape
    start = torch.tensor([3, 400])
    tok_c, lg_c = ref_cpu.ar_generate(sd, start, T - 1, ctx, mask, None, cached=True, return_logits=True)
    tok_u, lg_u = ref_cpu.ar_generate(sd, start, T - 1, ctx, mask, None, cached=False, return_logits=True)
    assert torch.equal(tok_c, tok_u)
    assert (lg_c - lg_u).abs().max() < 2e-4
    seq = torch.cat([start[:, None], tok_c], 1)
    lg_tf = ref_cpu.xt_decoder_logits(sd, seq[:, :-1], ctx, mask, None)
    assert (lg_tf - lg_c).abs().max() < 2e-4


def test_injected_noise_is_deterministic_and_not_greedy(setup):
    sd, v_s, v_a, mask, ctx = setup
    B, T = mask.shape
    from dimx import prng
    noise = torch.from_numpy(prng.exponential(5, "sc.noise", (T - 1, B, 512)))
    start = torch.tensor([3, 400])
    a = ref_cpu.ar_generate(sd, start, T - 1, ctx, mask, noise)
    b = ref_cpu.ar_generate(sd, start, T - 1, ctx, mask, noise)
    g = ref_cpu.ar_generate(sd, start, T - 1, ctx, mask, None)
    assert torch.equal(a, b) and not torch.equal(a, g)


def test_padding_never_influences_valid_outputs(setup):
    sd, v_s, v_a, mask, ctx = setup
    v_s2, v_a2 = v_s.clone(), v_a.clone()
    v_s2[1, 9:] = 7.0          # scribble over the padded frames of clip 1
    v_a2[1, 9:] = -3.0
    x1 = ref_cpu.slmft_forward_encoder(sd, v_s, mask)
    x2 = ref_cpu.slmft_forward_encoder(sd, v_s2, mask)
    assert torch.equal(x1[0], x2[0]) and (x1[1, :9] - x2[1, :9]).abs().max() < 1e-5
    c2 = ref_cpu.slmft_context(sd, x2, v_a2)
    t1 = ref_cpu.ar_generate(sd, torch.tensor([1, 2]), 13, ctx, mask, None)
    t2 = ref_cpu.ar_generate(sd, torch.tensor([1, 2]), 13, c2, mask, None)
    assert torch.equal(t1, t2)
    # the zero-fill of padded query rows ([XT?] in SURVEY.md) is immaterial for valid rows
    x3 = ref_cpu.slmft_forward_encoder(sd, v_s, mask, zero_masked_queries=False)
    assert (x1[1, :9] - x3[1, :9]).abs().max() < 1e-5 and torch.equal(x1[0], x3[0])


def test_kv_mask_shape_and_first_key(setup):
    g = torch.Generator().manual_seed(0)
    m = ref_cpu.ar_kv_mask(4, 300, 0.15, g)
    assert m.shape == (4, 299) and m[:, 0].all() and ((~m).sum(1) == 45).all()


def test_slmft_forward_modes(setup):
    sd, v_s, v_a, mask, ctx = setup
    from dimx import prng
    v_l = torch.from_numpy(prng.normal(1, "sc.vl", v_s.shape))
    total, d, pred = ref_cpu.slmft_forward(sd, v_s, v_l, v_a, mask, "train")
    assert pred.shape == (2, 13, 56) and torch.isfinite(total)
    total2, d2, pred2 = ref_cpu.slmft_forward(sd, v_s, v_l, v_a, mask, "val")
    assert pred2.shape == (2, 13, 56) and float(d2["l_ce_l"]) == 0.0
