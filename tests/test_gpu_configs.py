"""GPU: BASELINE.json's configurations that the other files only cover stage-wise:
  C2  B=1, T=300 ``SLMFT.forward(mode='train')`` against the CPU oracle (exact shape of the config);
  C4  the per-rank shape of the 8-GPU run: B=256, T=300 with ``batch_row_offset = 7*256`` (rank 7 of 8) -- decoded
      rows against the oracle with the same positional row, 128+128 shards against the whole, and the bf16 perf
      mode's properties at that shape;
  C5  the per-GPU shard of the long-context config: B=64, T=1500, **bf16** (split-KV decode attention on):
      determinism, range, KV-cached generation == teacher-forced logits, agreement with the f32 mode.
Sizes the oracle cannot finish in seconds go through size-independent properties (see test_gpu_fullsize.py).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _clips(B, T, seed, tag):
    from dimx import prng
    dev = torch.device("cuda:0")
    v_s = torch.from_numpy(prng.normal(seed, tag + ".vs", (B, T, 56))).to(dev)
    v_l = torch.from_numpy(prng.normal(seed, tag + ".vl", (B, T, 56))).to(dev)
    v_a = torch.from_numpy(prng.normal(seed, tag + ".va", (B, T, 768))).to(dev)
    return v_s, v_l, v_a


@pytest.fixture(scope="module")
def model_f32():
    from dimx.seq2seq_pretrain import SLMFT
    return SLMFT().eval()


@pytest.fixture(scope="module")
def model_bf16():
    from dimx import lib
    from dimx.seq2seq_pretrain import SLMFT
    return SLMFT(numeric_mode=lib.MODE_PERF_BF16).eval()


# ------------------------------------------------------------------------------------------------ C2
def test_c2_forward_train_b1_t300_matches_oracle(model_f32, full_sd):
    from oracle import ref_cpu
    B, T = 1, 300
    v_s, v_l, v_a = _clips(B, T, 52, "c2")
    mask = torch.ones(B, T, dtype=torch.bool, device=v_s.device)
    kv = ref_cpu.ar_kv_mask(B, T, 0.15, torch.Generator().manual_seed(52))
    tot, d, pred, tok = model_f32(v_s, v_l, v_a, mask, mode="train", kv_mask=kv.to(v_s.device), return_tokens=True)
    rt, rd, rpred, aux = ref_cpu.slmft_forward(full_sd, v_s.cpu(), v_l.cpu(), v_a.cpu(), mask.cpu(), "train",
                                               kv_mask=kv, return_aux=True)
    assert pred.shape == (1, T - 1, 56)
    # teacher-forced argmax tokens: identical wherever the oracle's own top-2 logit margin is not a rounding tie
    top2 = aux["logits"].topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert safe.float().mean() > 0.95
    assert torch.equal(tok.cpu()[safe], aux["logits"].argmax(-1)[safe])
    # decoded motion, unconditionally: the VQ decoder mixes all frames, so ONE flipped rounding-tie token would change
    # every frame of `pred`; the comparison therefore decodes the ORACLE's code sequence on the GPU and holds that against
    # the oracle's own decode (1e-4, the north-star tolerance) -- and the end-to-end `pred` too whenever no tie flipped
    otok = aux["logits"].argmax(-1).to(torch.int32).to(v_s.device)
    dec = model_f32.engine(v_s.device).vq_decode(1, otok, 0, 1)
    err = (dec.cpu() - rpred).abs().max().item()
    print("C2 decoded-motion max |gpu - oracle| = %.2e" % err)
    assert err < 1e-4
    if torch.equal(tok.cpu(), aux["logits"].argmax(-1)):
        assert (pred.cpu() - rpred).abs().max() < 1e-4
    assert abs(float(d["l_ce_l"]) - float(rd["l_ce_l"])) < 1e-3 * max(1.0, abs(float(rd["l_ce_l"])))
    assert abs(float(tot) - float(rt)) < 2e-3 * max(1.0, abs(float(rt)))


# ------------------------------------------------------------------------------------------------ C4
C4_B, C4_T, C4_OFF = 256, 300, 7 * 256


@pytest.fixture(scope="module")
def c4_clips():
    v_s, v_l, v_a = _clips(C4_B, C4_T, 54, "c4")
    mask = torch.ones(C4_B, C4_T, dtype=torch.bool, device=v_s.device)
    return v_s, v_l, v_a, mask


def test_c4_rank7_shard_f32_rows_match_oracle_and_subshards(model_f32, c4_clips, full_sd):
    from dimx import prng
    from oracle import ref_cpu
    v_s, v_l, v_a, mask = c4_clips
    noise = torch.from_numpy(prng.exponential(54, "c4.noise", (C4_T - 1, C4_B, 512))).cuda()
    _, _, pred, tok = model_f32(v_s, v_l, v_a, mask, mode="val", noise=noise, return_tokens=True,
                                batch_row_offset=C4_OFF)
    assert pred.shape == (C4_B, C4_T - 1, 56) and torch.isfinite(pred).all()
    assert int(tok.min()) >= 0 and int(tok.max()) < 512
    # decoded motion of single rows == oracle VQ decode of the same codes with positional row 1792 + r
    for r in (0, 131, 255):
        ref = ref_cpu.vq_decode(full_sd, tok[r:r + 1].cpu().long(), prefix="listener_vq.", row_offset=C4_OFF + r)
        assert (pred[r].cpu() - ref[0]).abs().max() < 1e-4, r
    # ... and differs from the un-offset decode (the quirk is real at this offset)
    ref0 = ref_cpu.vq_decode(full_sd, tok[0:1].cpu().long(), prefix="listener_vq.", row_offset=0)
    assert (pred[0].cpu() - ref0[0]).abs().max() > 1e-3
    # two 128-clip sub-shards with their own offsets reproduce the 256-clip shard (tokens bit-exact)
    for lo in (0, 128):
        sl = slice(lo, lo + 128)
        _, _, p, t = model_f32(v_s[sl].contiguous(), v_l[sl].contiguous(), v_a[sl].contiguous(), mask[sl].contiguous(),
                               mode="val", noise=noise[:, sl].contiguous(), return_tokens=True,
                               batch_row_offset=C4_OFF + lo)
        assert torch.equal(t, tok[sl])
        assert (p - pred[sl]).abs().max() < 1e-5


def test_c4_sampler_window_makes_seeded_shards_equal_the_whole(model_f32, c4_clips):
    """shard=(lo, total): the on-device generator is indexed by the global row, so seeded (not injected-noise)
    generation of a shard equals the same rows of the single-process batch."""
    v_s, v_l, v_a, mask = c4_clips
    n = 32
    args = [t[:n].contiguous() for t in (v_s, v_l, v_a, mask)]
    _, _, pred, tok = model_f32(*args, mode="val", seed=99, return_tokens=True)
    for lo, hi in ((0, 13), (13, 32)):
        a = [t[lo:hi].contiguous() for t in args]
        _, _, p, t = model_f32(*a, mode="val", seed=99, return_tokens=True, batch_row_offset=lo, shard=(lo, n))
        assert torch.equal(t, tok[lo:hi]) and (p - pred[lo:hi]).abs().max() < 1e-5
    # without the window the second shard draws row-0-based noise and differs
    a = [t[13:32].contiguous() for t in args]
    _, _, _, t = model_f32(*a, mode="val", seed=99, return_tokens=True, batch_row_offset=13)
    assert not torch.equal(t, tok[13:32])
    # seed=0 is a valid user seed (remapped, not silently greedy)
    _, _, _, t0 = model_f32(*args, mode="val", seed=0, return_tokens=True)
    _, _, _, tg = model_f32(*args, mode="val", greedy=True, return_tokens=True)
    assert not torch.equal(t0, tg)


def test_c4_rank7_shard_bf16_properties(model_bf16, model_f32, c4_clips):
    v_s, v_l, v_a, mask = c4_clips
    _, _, pred, tok = model_bf16(v_s, v_l, v_a, mask, mode="val", seed=77, return_tokens=True, batch_row_offset=C4_OFF)
    _, _, pred2, tok2 = model_bf16(v_s, v_l, v_a, mask, mode="val", seed=77, return_tokens=True, batch_row_offset=C4_OFF)
    assert torch.isfinite(pred).all() and int(tok.min()) >= 0 and int(tok.max()) < 512
    assert torch.equal(tok, tok2) and torch.equal(pred, pred2)          # bit-reproducible
    # the offset reaches the decoder: same codes, other positional rows -> other motion
    e = model_bf16.engine(v_s.device)
    p_off = e.vq_decode(1, tok[:8].contiguous(), C4_OFF)
    p_0 = e.vq_decode(1, tok[:8].contiguous(), 0)
    assert (p_off - pred[:8]).abs().max() < 5e-2 and (p_off - p_0).abs().max() > 1e-3
    # bf16 decode of these codes vs the f32 engine's decode of the same codes
    p_f = model_f32.engine(v_s.device).vq_decode(1, tok[:8].contiguous(), C4_OFF)
    err = (p_off - p_f).abs().max().item()
    print("C4 bf16 vs f32 VQ decode of the same codes (row offset 1792): max err %.4f" % err)
    assert err <= 3e-2          # measured 0.0124


# ------------------------------------------------------------------------------------------------ C5
def test_c5_shard_b64_t1500_bf16(model_bf16, model_f32):
    Bc, Tc = 64, 1500
    v_s, v_l, v_a = _clips(Bc, Tc, 55, "c5s")
    lens = torch.full((Bc,), Tc, dtype=torch.int64)
    lens[5], lens[17], lens[40] = 1203, 777, 64
    mask = (torch.arange(Tc)[None, :] < lens[:, None]).cuda()
    eng = model_bf16.engine(v_s.device)
    m8 = mask.to(torch.uint8).contiguous()
    _, z_l = model_bf16.forward_vq(v_s, v_l, mask, with_speaker=False)
    assert (z_l[5, 1203:] == -100).all() and int(z_l[0].min()) >= 0
    # seeded stochastic generation: deterministic, in range
    eng.encode_ctx(v_s, v_a, m8, True)
    t1 = eng.generate(z_l[:, 0].contiguous(), m8, Tc, 1.0, 52, None, 1234)
    eng.encode_ctx(v_s, v_a, m8, True)
    t2 = eng.generate(z_l[:, 0].contiguous(), m8, Tc, 1.0, 52, None, 1234)
    assert torch.equal(t1, t2) and int(t1.min()) >= 0 and int(t1.max()) < 512
    # greedy KV-cached generation (B*H = 768 pairs -> 4 waves per pair: split-KV path) == teacher-forced logits
    eng.encode_ctx(v_s, v_a, m8, True)
    g_tok, g_logits = eng.generate(z_l[:, 0].contiguous(), m8, Tc, 0.0, 52, None, 0, return_logits=True)
    assert torch.isfinite(g_logits).all()
    seq = torch.cat([z_l[:, :1].to(torch.int32), g_tok], 1).contiguous()
    eng.encode_ctx(v_s, v_a, m8, False)
    tf_logits, _, tf_arg = eng.decode_tf(seq, m8, None)
    cerr = (tf_logits - g_logits).abs().max().item()
    top2 = tf_logits.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 0.1
    cagree = (tf_arg[safe] == g_tok[safe]).float().mean().item()
    # the same teacher-forced pass in the f32 mode on the same inputs
    ef = model_f32.engine(v_s.device)
    ef.encode_ctx(v_s, v_a, m8, False)
    lf, _, af = ef.decode_tf(seq, m8, None)
    valid = mask[:, 1:]
    lerr = (tf_logits - lf).abs()[valid].max().item()
    agree = (tf_arg == af)[valid].float().mean().item()
    print("C5 shard bf16: cache vs teacher-forced max logit err %.4f (argmax agreement on clear margins %.4f); "
          "bf16 vs f32 teacher-forced: max logit err %.4f, argmax agreement %.4f" % (cerr, cagree, lerr, agree))
    # measured on MI355X (round 2): 0.0078 / 1.0000 and 0.0117 / 0.9938 -- asserted with a small margin
    assert cerr <= 2.5e-2 and cagree >= 0.999
    assert lerr <= 3e-2 and agree >= 0.99
    pred = eng.vq_decode(1, g_tok)
    assert pred.shape == (Bc, Tc - 1, 56) and torch.isfinite(pred).all()
