"""GPU: BASELINE.json's full single-GPU size (C3: B=256, T=300, autoregressive decode) through
size-independent properties, since the CPU oracle cannot finish that size in seconds:
  * batch invariance   -- a clip's result does not depend on which other clips share the batch;
  * shard invariance   -- two 128-clip shards with batch_row_offset reproduce the 256-clip batch (C4's layout);
  * determinism        -- same injected seed -> identical tokens; different seed -> different tokens;
  * cache consistency  -- KV-cached generation == teacher-forced logits at the generated prefix;
  * bf16 perf mode     -- finite, in-range, deterministic, and close to the f32 mode on the same inputs.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

B, T = 256, 300


@pytest.fixture(scope="module")
def clips():
    from dimx import prng
    dev = torch.device("cuda:0")
    v_s = torch.from_numpy(prng.normal(31, "full.vs", (B, T, 56))).to(dev)
    v_l = torch.from_numpy(prng.normal(31, "full.vl", (B, T, 56))).to(dev)
    v_a = torch.from_numpy(prng.normal(31, "full.va", (B, T, 768))).to(dev)
    lens = torch.from_numpy(prng.integers(31, "full.lens", (B,), 5, T + 1))
    lens[:64] = T
    mask = (torch.arange(T)[None, :] < lens[:, None]).to(dev)
    return v_s, v_l, v_a, mask, lens


@pytest.fixture(scope="module")
def model_f32():
    from dimx.seq2seq_pretrain import SLMFT
    return SLMFT().eval()


@pytest.fixture(scope="module")
def full_f32(model_f32, clips):
    v_s, v_l, v_a, mask, lens = clips
    tot, d, pred, tok = model_f32(v_s, v_l, v_a, mask, mode="val", seed=4242, return_tokens=True)
    return pred, tok


def test_c3_batch_and_shard_invariance(model_f32, clips, full_f32):
    v_s, v_l, v_a, mask, lens = clips
    pred, tok = full_f32
    assert pred.shape == (B, T - 1, 56) and tok.shape == (B, T - 1)
    assert torch.isfinite(pred).all() and int(tok.min()) >= 0 and int(tok.max()) < 512
    # NOTE the counter-based sampler noise is indexed by (step, clip row, code), so a sub-batch that keeps
    # its row positions draws the same noise: rows 0..7 alone == rows 0..7 of the full batch
    _, _, p8, t8 = model_f32(v_s[:8].contiguous(), v_l[:8].contiguous(), v_a[:8].contiguous(),
                             mask[:8].contiguous(), mode="val", seed=4242, return_tokens=True)
    # the noise stream depends on the batch size through (step*B + row): compare with injected noise instead
    from dimx import prng
    noise = torch.from_numpy(prng.exponential(5, "full.noise", (T - 1, 16, 512))).cuda()
    a = model_f32(v_s[:16].contiguous(), v_l[:16].contiguous(), v_a[:16].contiguous(), mask[:16].contiguous(),
                  mode="val", noise=noise, return_tokens=True)
    b0 = model_f32(v_s[:8].contiguous(), v_l[:8].contiguous(), v_a[:8].contiguous(), mask[:8].contiguous(),
                   mode="val", noise=noise[:, :8].contiguous(), return_tokens=True)
    b1 = model_f32(v_s[8:16].contiguous(), v_l[8:16].contiguous(), v_a[8:16].contiguous(), mask[8:16].contiguous(),
                   mode="val", noise=noise[:, 8:].contiguous(), return_tokens=True, batch_row_offset=8)
    assert torch.equal(a[3][:8], b0[3]) and torch.equal(a[3][8:], b1[3])
    assert (a[2][:8] - b0[2]).abs().max() < 1e-5 and (a[2][8:] - b1[2]).abs().max() < 1e-5


def test_c3_determinism_and_cache_consistency(model_f32, clips, full_f32):
    v_s, v_l, v_a, mask, lens = clips
    pred, tok = full_f32
    _, _, pred2, tok2 = model_f32(v_s, v_l, v_a, mask, mode="val", seed=4242, return_tokens=True)
    assert torch.equal(tok, tok2) and torch.equal(pred, pred2)
    _, _, _, tok3 = model_f32(v_s, v_l, v_a, mask, mode="val", seed=4243, return_tokens=True)
    assert not torch.equal(tok, tok3)
    # greedy generation must be the argmax chain of the teacher-forced pass over its own output
    eng = model_f32.engine(v_s.device)
    m8 = mask.to(torch.uint8).contiguous()
    _, z_l = model_f32.forward_vq(v_s, v_l, mask, with_speaker=False)
    eng.encode_ctx(v_s, v_a, m8, True)
    g_tok, g_logits = eng.generate(z_l[:, 0].contiguous(), m8, T, 0.0, 52, None, 0, return_logits=True)
    seq = torch.cat([z_l[:, :1].to(torch.int32), g_tok], 1).contiguous()
    eng.encode_ctx(v_s, v_a, m8, False)
    tf_logits, _, tf_arg = eng.decode_tf(seq, m8, None)
    assert (tf_logits - g_logits).abs().max() < 5e-3
    top2 = tf_logits.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-2
    assert torch.equal(tf_arg[safe], g_tok[safe])


def test_c3_bf16_perf_mode(clips, full_f32):
    from dimx import lib
    from dimx.seq2seq_pretrain import SLMFT
    v_s, v_l, v_a, mask, lens = clips
    m = SLMFT(numeric_mode=lib.MODE_PERF_BF16).eval()
    tot, d, pred, tok = m(v_s, v_l, v_a, mask, mode="val", seed=4242, return_tokens=True)
    tot2, _, pred2, tok2 = m(v_s, v_l, v_a, mask, mode="val", seed=4242, return_tokens=True)
    assert torch.isfinite(pred).all() and int(tok.min()) >= 0 and int(tok.max()) < 512
    # split-K partial sums are reduced in a fixed order by the consumer kernels: perf mode is bit-reproducible
    assert torch.equal(tok, tok2) and torch.equal(pred, pred2)
    # teacher-forced comparison with the f32 mode on identical inputs (no sampling feedback)
    eng, m8 = m.engine(v_s.device), mask.to(torch.uint8).contiguous()
    _, z_l = m.forward_vq(v_s, v_l, mask, with_speaker=False)
    eng.encode_ctx(v_s, v_a, m8, False)
    lb, _, ab = eng.decode_tf(z_l.to(torch.int32).contiguous(), m8, None)
    from dimx.seq2seq_pretrain import SLMFT as S2
    mf = S2().eval()
    ef = mf.engine(v_s.device)
    _, z_f = mf.forward_vq(v_s, v_l, mask, with_speaker=False)
    agree_idx = (z_f == z_l)[mask].float().mean().item()
    ef.encode_ctx(v_s, v_a, m8, False)
    lf, _, af = ef.decode_tf(z_l.to(torch.int32).contiguous(), m8, None)
    valid = mask[:, 1:]
    agree = (ab == af)[valid].float().mean().item()
    err = (lb - lf).abs()[valid].max().item()
    print("C3 bf16 vs f32: VQ index agreement %.4f, argmax agreement %.4f, max logit err %.4f" % (agree_idx, agree, err))
    # measured on MI355X (round 2): 0.9915 / 0.9952 / 0.0093 -- asserted with a small margin
    assert agree_idx >= 0.98 and agree >= 0.99 and err <= 2e-2


def test_c5_long_context_t1500(model_f32):
    """BASELINE config C5's sequence length (T=1500 > one LDS K/V tile, decode cache of 1500 keys): the
    KV-cached generation must reproduce the teacher-forced logits over its own output, and the VQ round trip
    at T=1500 stays on the reference golden (tests/test_gpu_vq.py::test_encode_indices_bit_exact[1500])."""
    from dimx import prng
    dev = torch.device("cuda:0")
    Bc, Tc = 2, 1500
    v_s = torch.from_numpy(prng.normal(41, "c5.vs", (Bc, Tc, 56))).to(dev)
    v_l = torch.from_numpy(prng.normal(41, "c5.vl", (Bc, Tc, 56))).to(dev)
    v_a = torch.from_numpy(prng.normal(41, "c5.va", (Bc, Tc, 768))).to(dev)
    mask = torch.ones(Bc, Tc, dtype=torch.bool, device=dev)
    mask[1, 1203:] = False
    eng = model_f32.engine(dev)
    m8 = mask.to(torch.uint8).contiguous()
    _, z_l = model_f32.forward_vq(v_s, v_l, mask, with_speaker=False)
    assert (z_l[1, 1203:] == -100).all() and int(z_l[0].min()) >= 0
    eng.encode_ctx(v_s, v_a, m8, True)
    g_tok, g_logits = eng.generate(z_l[:, 0].contiguous(), m8, Tc, 0.0, 52, None, 0, return_logits=True)
    seq = torch.cat([z_l[:, :1].to(torch.int32), g_tok], 1).contiguous()
    eng.encode_ctx(v_s, v_a, m8, False)
    tf_logits, _, tf_arg = eng.decode_tf(seq, m8, None)
    assert torch.isfinite(g_logits).all()
    assert (tf_logits - g_logits).abs().max() < 5e-3
    top2 = tf_logits.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-2
    assert torch.equal(tf_arg[safe], g_tok[safe])
    pred = eng.vq_decode(1, g_tok)
    assert pred.shape == (Bc, Tc - 1, 56) and torch.isfinite(pred).all()


def test_c3_bf16_mode_leaves_the_generated_motion_distribution_where_the_f32_mode_has_it(model_f32):
    """free-running generation diverges after the first token a bf16 rounding tie flips, so per-token
    agreement says nothing about the generated MOTION.  The reference's own statistics do (per-clip Frechet distance, MSE,
    variance on pose / exp: code/metrics/eval_utils.py:12-46, code/mymetrics.py:7-88): C3 shape, full clips, same sampler
    seed through both modes.  Measured on MI355X (round 5): FD between the modes = 0.26-0.27 x the FD between two sampler
    seeds of the f32 mode; FD / MSE / variance against the target motion move by <= 0.2 %; tokens agree for 150 steps on
    average before the first flip.  Asserted with margins."""
    from dimx import lib, prng
    from dimx.mode_compare import compare_modes
    from dimx.seq2seq_pretrain import SLMFT, mark_prefix
    dev = torch.device("cuda:0")
    v_s = torch.from_numpy(prng.normal(77, "cmp.vs", (B, T, 56))).to(dev)
    v_l = torch.from_numpy(prng.normal(77, "cmp.vl", (B, T, 56))).to(dev)
    v_a = torch.from_numpy(prng.normal(77, "cmp.va", (B, T, 768))).to(dev)
    mask = mark_prefix(torch.ones(B, T, dtype=torch.bool, device=dev))
    m_bf16 = SLMFT(numeric_mode=lib.MODE_PERF_BF16).eval()
    r = compare_modes(m_bf16, model_f32, v_s, v_l, v_a, mask, seed=991)
    print("bf16 vs f32 generated motion:", {k: r[k] for k in ("free_running_token_agreement", "mean_steps_before_first_flip")},
          {p: r["between_modes"][p]["fd_over_seed_spread"] for p in ("pose", "exp")},
          r["vs_target"]["pose_relative_shift"], r["vs_target"]["exp_relative_shift"])
    assert r["mean_steps_before_first_flip"] > 60
    for part in ("pose", "exp"):
        assert r["between_modes"][part]["fd_over_seed_spread"] < 0.6, "the mode difference must stay inside the sampling spread"
        for k, v in r["vs_target"][part + "_relative_shift"].items():
            assert v < 0.01, (part, k, v)
