"""GPU: the legacy ``ListenerGenerator`` path (reference code/seq2seq.py:138-306 driven by
code/x_engine.py:65-88) through the C-ABI (variant 1) against the committed reference fixture and the CPU
oracle on the same seeded inputs.  f32 parity mode: indices / tokens bit-exact, logits to 2e-3 absolute
(f32 MFMA vs CPU fp32 summation order; logits are O(1))."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def legacy_sd():
    from dimx import weights
    return weights.synth_state_dict(weights.listener_generator_spec(), 20260928)


@pytest.fixture(scope="module")
def leng(legacy_sd):
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32, "legacy")
    e.load_state_dict(legacy_sd)
    assert e.missing_weights() == 0
    return e


@pytest.fixture(scope="module")
def leng_bf16(legacy_sd):
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PERF_BF16, "legacy")
    e.load_state_dict(legacy_sd)
    return e


def _case(B, T, lens, seed=5):
    from dimx import prng
    v_s = torch.from_numpy(prng.normal(seed, "legacy.vs", (B, T, 824)))
    v_l = torch.from_numpy(prng.normal(seed, "legacy.vl", (B, T, 56)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    return v_s, v_l, mask


def test_speaker_features_match_reference_fixture(leng):
    """x_speaker and the code indices against tests/golden/legacy_speaker_features.npz, which was produced by
    the reference's own VQSpeakerAutoEncoder (tests/golden/make_golden.py --legacy)."""
    g = np.load(os.path.join(HERE, "golden", "legacy_speaker_features.npz"))
    v = torch.from_numpy(g["v_speaker"])
    lens = g["lens"]
    B, T, _ = v.shape
    mask = torch.zeros(B, T, dtype=torch.uint8)
    for j, n in enumerate(lens):
        mask[j, :n] = 1
    x, idx = leng.legacy_speaker_features(v.cuda(), mask.cuda(), return_idx=True)
    ref_idx = torch.from_numpy(g["idx"].astype(np.int64))
    got = idx.cpu().long()
    valid = ref_idx >= 0
    assert torch.equal(got[valid], ref_idx[valid])
    assert (got[~valid] == -100).all()
    # The reference returns z + (z_q - z).detach() (straight-through form, quantizer.py:60), i.e. the codebook
    # row re-rounded through the encoder output: <= 1 ulp from the row itself.  The engine gathers the rows.
    assert np.abs(x.cpu().numpy() - g["x_speaker"]).max() < 1e-6


def test_speaker_vq_encode_entry_point(leng, legacy_sd):
    """dimx_vq_encode(which=0) of variant 1: [B,T,824] -> z [B,T,1024], idx [B,T*8] (batched, row-b PE)."""
    from oracle import ref_cpu
    v_s, _, _ = _case(2, 16, [16, 16])
    idx, z = leng.vq_encode(0, v_s.cuda(), None, pe_mode=1, return_z=True)
    h = ref_cpu.vq_encode_features(legacy_sd, v_s, 8, 6, "speaker_vq.", True, 0)
    assert (z.cpu() - h).abs().max() < 2e-4
    ref_idx, d = ref_cpu.vq_quantize(h.reshape(-1, 128), legacy_sd["speaker_vq.quantize.embedding.weight"])
    safe = ref_cpu.vq_margins(d) > 1e-4
    assert torch.equal(idx.cpu().long().view(-1)[safe], ref_idx[safe])


@pytest.mark.parametrize("B,T,lens", [(3, 24, [24, 17, 5]), (2, 64, [64, 40])])
def test_forward_matches_oracle(leng, legacy_sd, B, T, lens):
    from dimx import seq2seq
    from oracle import ref_cpu
    v_s, v_l, mask = _case(B, T, lens)
    ref_loss, ref_pred, aux = ref_cpu.listener_generator_forward(legacy_sd, v_s, v_l, mask)
    m8 = mask.to(torch.uint8).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    z_l = leng.vq_encode(1, v_l.cuda(), lens_t, pe_mode=0, pad_value=-100)
    assert torch.equal(z_l.cpu().long(), aux["z_l"])
    x_sp = leng.legacy_speaker_features(v_s.cuda(), m8)
    assert torch.equal(x_sp.cpu(), aux["x_speaker"])
    enc = leng.encode_ctx(v_s.cuda(), None, m8, False, return_x_s=True).cpu()
    for b, n in enumerate(lens):
        assert (enc[b, :n] - aux["enc"][b, :n]).abs().max() < 2e-4
    logits, row_loss, amax = leng.decode_tf(z_l, m8, None)
    assert (logits.cpu() - aux["logits"]).abs().max() < 2e-3
    # module surface: same loss and decoded motion as the oracle
    m = seq2seq.ListenerGenerator().cuda()
    loss, pred = m(v_s.cuda(), v_l.cuda(), mask.cuda())
    assert abs(loss.item() - ref_loss.item()) < 1e-3 * max(1.0, abs(ref_loss.item()))
    top2 = aux["logits"].topk(2, -1).values
    if ((top2[..., 0] - top2[..., 1]) > 1e-3).all():
        for b, n in enumerate(lens):
            assert (pred.cpu()[b, :n - 1] - ref_pred[b, :n - 1]).abs().max() < 1e-3


@pytest.mark.parametrize("B,T,lens,noisy", [(3, 24, [24, 17, 5], False), (3, 24, [24, 17, 5], True),
                                            (2, 96, [96, 70], True)])
def test_generate_matches_oracle(leng, legacy_sd, B, T, lens, noisy):
    from dimx import prng
    from oracle import ref_cpu
    v_s, v_l, mask = _case(B, T, lens, seed=8)
    noise = torch.from_numpy(prng.exponential(12, "legacy.noise", (T, B, 512))) if noisy else None
    if noisy:
        ref_tok, z_ref = ref_cpu.listener_generator_generate(legacy_sd, v_s, v_l, mask, noise)
    else:
        x_speaker = ref_cpu.legacy_speaker_features(legacy_sd, v_s, mask)
        _, z_ref = ref_cpu.forward_vq(legacy_sd, v_s, v_l, mask, with_speaker=False)
        enc = ref_cpu.xt_encoder(legacy_sd, "generator.encoder.", x_speaker, mask, causal=False, depth=6, heads=8)
        ref_tok = ref_cpu.legacy_generate(legacy_sd, z_ref[:, 0], T, enc, mask, None, temperature=0.0)
    m8 = mask.to(torch.uint8).cuda()
    leng.encode_ctx(v_s.cuda(), None, m8, True)
    tok, lg = leng.generate(z_ref[:, 0].cuda(), m8, T, 1.0 if noisy else 0.0, 52, noise.cuda() if noisy else None,
                            return_logits=True)
    assert tuple(tok.shape) == (B, T)
    same = tok.cpu().long() == ref_tok
    assert same.all(), "token mismatch: %d/%d differ" % ((~same).sum(), same.numel())
    # KV-cached generation == teacher-forced logits over the sampled sequence (GPU self-consistency):
    # decode_tf feeds T-1 tokens, so compare the first T-1 steps
    seq = torch.cat([z_ref[:, :1], tok.cpu().long()[:, :T - 1]], 1)
    leng.encode_ctx(v_s.cuda(), None, m8, False)
    tf_logits, _, _ = leng.decode_tf(seq.cuda(), m8, None)
    assert (tf_logits.cpu() - lg.cpu()[:, :T - 1]).abs().max() < 2e-3


def test_bf16_mode_agrees(leng, leng_bf16):
    v_s, v_l, mask = _case(4, 48, [48, 48, 30, 11], seed=2)
    m8 = mask.to(torch.uint8).cuda()
    lens_t = mask.sum(1).to(torch.int32).cuda()
    z_l = leng.vq_encode(1, v_l.cuda(), lens_t, pe_mode=0, pad_value=-100)
    outs = []
    for e in (leng, leng_bf16):
        e.encode_ctx(v_s.cuda(), None, m8, False)
        outs.append(e.decode_tf(z_l, m8, None)[0].cpu())
    valid = mask[:, 1:]
    err = (outs[0] - outs[1])[valid].abs().max().item()
    assert err < 0.25, "bf16 vs f32 logits differ by %g" % err
    # reproducible generation in perf mode (deterministic split-K slabs)
    leng_bf16.encode_ctx(v_s.cuda(), None, m8, True)
    a = leng_bf16.generate(z_l[:, 0].contiguous(), m8, 48, 1.0, 52, None, seed=7).cpu()
    leng_bf16.encode_ctx(v_s.cuda(), None, m8, True)
    b = leng_bf16.generate(z_l[:, 0].contiguous(), m8, 48, 1.0, 52, None, seed=7).cpu()
    assert torch.equal(a, b)


def test_evaluate_epoch_protocol(legacy_sd):
    """x_engine.evaluate_epoch over a two-batch loader == perplexity computed from the oracle's logits."""
    from dimx import seq2seq, x_engine
    from oracle import ref_cpu
    model = seq2seq.ListenerGenerator().cuda()
    batches, nll, cnt = [], 0.0, 0
    for seed, lens in ((21, [20, 13]), (22, [20, 20])):
        v_s, v_l, mask = _case(2, 20, lens, seed=seed)
        batches.append((v_s, v_l, lens, None))
        _, _, aux = ref_cpu.listener_generator_forward(legacy_sd, v_s, v_l, mask)
        lp = torch.log_softmax(aux["logits"].double(), -1)
        tgt = aux["z_l"][:, 1:]
        sel = mask[:, 1:]
        nll += float(-lp[sel].gather(1, tgt[sel][:, None]).sum())
        cnt += int(sel.sum())
    ppl = x_engine.evaluate_epoch(model, batches, torch.device("cuda:0"), generate_kw={"greedy": True}, verbose=False)
    assert abs(ppl - np.exp(nll / cnt)) < 1e-3 * np.exp(nll / cnt)


def test_train_epoch_on_the_legacy_generator(legacy_sd):
    """reference code/x_engine.py:8-36 on the drop-in module: the training forward (frozen VQ halves from the HIP engine,
    generator / listener VQ decoder / id embeddings on autograd) has the loss and gradients of autograd over the oracle;
    three AdamW steps lower the loss, touch only what the reference trains, and the engine re-packs the new weights."""
    from dimx import seq2seq, x_engine
    from dimx import train as T
    from oracle import ref_cpu
    dev = torch.device("cuda:0")
    model = seq2seq.ListenerGenerator().to(dev)
    T.set_legacy_trainable(model)
    model.train()
    assert not model.speaker_vq.training and not model.listener_vq.training
    v_s, v_l, mask = _case(2, 20, [20, 13], seed=21)
    lid = torch.tensor([3, 41])
    with torch.enable_grad():
        loss, pred = model(v_s.to(dev), v_l.to(dev), mask.to(dev), speaker_ids=None, listener_ids=lid.to(dev))
        loss.backward()
        sd = {k: v.detach().clone().requires_grad_(k.startswith(T.LEGACY_TRAINABLE_PREFIXES)) for k, v in legacy_sd.items()}
        o_loss, o_pred, _ = ref_cpu.listener_generator_forward(sd, v_s, v_l, mask, listener_ids=lid)
        o_loss.backward()
    assert abs(loss.item() - o_loss.item()) < 1e-4 * max(1.0, abs(o_loss.item()))
    assert (pred.cpu() - o_pred).abs().max().item() < 1e-3
    worst = 0.0
    for k, p in model.named_parameters():
        if sd[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        rel = (p.grad.cpu() - sd[k].grad).abs().max().item() / max(sd[k].grad.abs().max().item(), 1e-8)
        worst = max(worst, rel)
        assert rel < 1e-3, (k, rel)
    print("legacy training forward: worst relative gradient error vs autograd over the oracle %.2e" % worst)
    # the loop
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tok0, _ = model.generate(v_s.to(dev), v_l.to(dev), mask.to(dev), greedy=True)
    batch = (v_s, v_l, [20, 13], (torch.tensor([0, 1]), lid), ["a", "b"])
    opt = torch.optim.AdamW([p for _, p in T.legacy_trainable_parameters(model)], lr=2e-4)
    first = x_engine.train_epoch(model, [batch], opt, dev, clip=1.0)
    for _ in range(3):
        last = x_engine.train_epoch(model, [batch], opt, dev, clip=1.0)
    assert last < first - 0.05, (first, last)
    after = model.state_dict()
    for k in before:
        changed = not torch.equal(before[k], after[k])
        if k.startswith(("speaker_vq.", "listener_vq.encoder.", "listener_vq.quantize.", "speaker_embeddings.", "fc_speaker.")):
            assert not changed, k            # frozen, or without a gradient when speaker_ids is None
        elif k.startswith(("generator.decoder.net.attn_layers", "listener_vq.decoder.decoder_transformer", "fc_listener.")):
            assert changed, k
    model.eval()
    tok1, _ = model.generate(v_s.to(dev), v_l.to(dev), mask.to(dev), greedy=True)
    assert tok1.shape == tok0.shape and not torch.equal(tok1, tok0)       # the engine sees the trained weights


def test_fullsize_properties_B64_T300(legacy_sd):
    """size-independent properties at the C3 sequence length: batch / shard invariance of the generated tokens,
    determinism of the seeded sampler, KV-cached generation == teacher-forced logits (f32 parity mode)."""
    from dimx import engine, lib, prng
    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32, "legacy")
    e.load_state_dict(legacy_sd)
    B, T = 16, 300
    lens = [300] * 10 + [287, 211, 150, 97, 33, 8]
    v_s, v_l, mask = _case(B, T, lens, seed=31)
    v_s, v_l = v_s.cuda(), v_l.cuda()
    m8 = mask.to(torch.uint8).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    z_l = e.vq_encode(1, v_l, lens_t, pe_mode=0, pad_value=-100)
    noise = torch.from_numpy(prng.exponential(6, "legacy.full.noise", (T, B, 512))).cuda()

    def gen(sl, nz):
        e.encode_ctx(v_s[sl].contiguous(), None, m8[sl].contiguous(), True)
        return e.generate(z_l[sl, 0].contiguous(), m8[sl].contiguous(), T, 1.0, 52, nz, return_logits=True)
    tok, lg = gen(slice(0, B), noise)
    t0, _ = gen(slice(0, 8), noise[:, :8].contiguous())
    t1, _ = gen(slice(8, 16), noise[:, 8:].contiguous())
    assert torch.equal(tok[:8], t0) and torch.equal(tok[8:], t1)          # shard invariance
    tok2, _ = gen(slice(0, B), noise)
    assert torch.equal(tok, tok2)                                           # determinism
    assert int(tok.min()) >= 0 and int(tok.max()) < 512
    seq = torch.cat([z_l[:, :1], tok[:, :T - 1]], 1).contiguous()
    e.encode_ctx(v_s, None, m8, False)
    tf_logits, _, _ = e.decode_tf(seq, m8, None)
    assert (tf_logits - lg[:, :T - 1]).abs().max() < 3e-3                   # cache consistency over 299 steps


def test_maximum_context_T1024(legacy_sd):
    """max_seq_len = 1024 of the legacy generator (decoder positional table bound): bf16, finite, in range."""
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PERF_BF16, "legacy")
    e.load_state_dict(legacy_sd)
    v_s, v_l, mask = _case(2, 1024, [1024, 700], seed=9)
    m8 = mask.to(torch.uint8).cuda()
    z_l = e.vq_encode(1, v_l.cuda(), mask.sum(1).to(torch.int32).cuda(), pe_mode=0, pad_value=-100)
    e.encode_ctx(v_s.cuda(), None, m8, True)
    tok = e.generate(z_l[:, 0].contiguous(), m8, 1024, 1.0, 52, None, seed=5)
    assert tuple(tok.shape) == (2, 1024) and int(tok.min()) >= 0 and int(tok.max()) < 512
    with pytest.raises(lib.DimxError):
        e.workspace(1, 1025)
