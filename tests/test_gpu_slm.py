"""GPU: the SLM pre-training forward (reference code/seq2seq_pretrain.py:58-323) through the C-ABI
(variant 2) against the CPU oracle on the same seeded inputs and injected random masks.  f32 parity mode; encoder
outputs 2e-4, logits 2e-3 absolute, losses 1e-3 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def slm_sd():
    from dimx import weights
    return weights.synth_state_dict(weights.slm_spec(), 20260928)


def _case(B, T, lens, seed=4, ratio=0.3):
    from dimx import prng
    from oracle import ref_cpu
    v_s = torch.from_numpy(prng.normal(seed, "slm.vs", (B, T, 56)))
    v_l = torch.from_numpy(prng.normal(seed, "slm.vl", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(seed, "slm.va", (B, T, 768)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    g = torch.Generator().manual_seed(seed)
    ms = ref_cpu.slm_random_masks(mask, ratio, g)
    ml = ref_cpu.slm_random_masks(mask, ratio, g)
    return v_s, v_l, v_a, mask, ms, ml


@pytest.mark.parametrize("B,T,lens", [(3, 40, [40, 33, 12]), (2, 96, [96, 60])])
def test_slm_forward_matches_oracle(slm_sd, B, T, lens):
    from dimx.seq2seq_pretrain import SLM
    from oracle import ref_cpu
    v_s, v_l, v_a, mask, ms, ml = _case(B, T, lens)
    ref_total, ref_d, _, aux = ref_cpu.slm_forward(slm_sd, v_s, v_l, v_a, mask, ms, ml, return_aux=True)
    m = SLM().cuda()
    total, d, none, got = m(v_s.cuda(), v_l.cuda(), v_a.cuda(), mask.cuda(), mask_speaker=ms.cuda(),
                            mask_listener=ml.cuda(), return_aux=True)
    assert none is None and set(d) == set(ref_d)
    for b, n in enumerate(lens):
        assert (got["x_s"].cpu()[b, :n] - aux["x_s"][b, :n]).abs().max() < 2e-4
        assert (got["x_l"].cpu()[b, :n] - aux["x_l"][b, :n]).abs().max() < 2e-4
        xj, rj = got["x_joint"].cpu()[b], aux["x_joint"][b]
        assert (xj[:n] - rj[:n]).abs().max() < 2e-4 and (xj[T:T + n] - rj[T:T + n]).abs().max() < 2e-4
    assert (got["px_s"].cpu() - aux["px_s"]).abs().max() < 2e-3
    assert (got["px_l"].cpu() - aux["px_l"]).abs().max() < 2e-3
    for k in ("l_ce_s", "l_ce_l", "nce"):
        assert abs(float(d[k]) - float(ref_d[k])) < 1e-3 * max(1.0, abs(float(ref_d[k]))), k
    assert abs(float(d["c_acc"]) - float(ref_d["c_acc"])) < 1e-6
    # decoded motion / continuous losses: compare when every argmax decision is clear of the logit tolerance
    clear = all(((a.topk(2, -1).values[..., 0] - a.topk(2, -1).values[..., 1]) > 5e-3).all() for a in (aux["px_s"], aux["px_l"]))
    if clear:
        assert (got["pred_l"].cpu() - aux["pred_l"]).abs().max() < 1e-3
        assert abs(float(total) - float(ref_total)) < 1e-3 * abs(float(ref_total))


def test_slm_default_masks_and_bf16(slm_sd):
    """default (fresh) masks run; bf16 perf mode agrees with f32 on the encoder outputs."""
    from dimx import lib
    from dimx.seq2seq_pretrain import SLM
    v_s, v_l, v_a, mask, ms, ml = _case(4, 64, [64, 64, 50, 9], seed=2, ratio=0.15)
    outs = []
    for mode in (lib.MODE_PARITY_F32, lib.MODE_PERF_BF16):
        m = SLM(numeric_mode=mode).cuda()
        total, d, _, aux = m(v_s.cuda(), v_l.cuda(), v_a.cuda(), mask.cuda(), mask_speaker=ms.cuda(),
                             mask_listener=ml.cuda(), return_aux=True)
        assert torch.isfinite(total)
        outs.append(aux["x_joint"].cpu())
    valid = torch.cat([mask, mask], 1)
    assert (outs[0] - outs[1])[valid].abs().max() < 0.15
    m = SLM().cuda()
    total, d, _ = m(v_s.cuda(), v_l.cuda(), v_a.cuda(), mask.cuda())
    assert torch.isfinite(total) and 0.0 <= float(d["c_acc"]) <= 1.0


def test_slm_training_forward_and_train_epoch(slm_sd):
    """SLM in training (what code/train_s2s_pretrain.py:41-64 runs through x_engine_pt.train_epoch): the autograd forward
    (frozen VQ encoders on the HIP engine) reports the losses of the HIP inference forward on the same masks, its gradients
    are autograd's over the oracle, and AdamW steps through train_epoch lower the loss and leave the frozen parts alone."""
    from dimx import train as T
    from dimx import x_engine_pt
    from dimx.seq2seq_pretrain import SLM
    from oracle import ref_cpu
    dev = torch.device("cuda:0")
    v_s, v_l, v_a, mask, ms, ml = _case(3, 40, [40, 33, 12])
    m = SLM().to(dev)
    ref_total, ref_d, _ = m(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mask_speaker=ms.to(dev), mask_listener=ml.to(dev))
    T.set_slm_trainable(m)
    m.train()
    with torch.enable_grad():
        total, d, none = m(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mask_speaker=ms.to(dev), mask_listener=ml.to(dev))
        assert none is None and total.requires_grad
        total.backward()
        trainable = lambda k: not k.startswith(T.SLM_FROZEN_PREFIXES) and not k.endswith(".pe")
        sd = {k: v.detach().clone().requires_grad_(trainable(k)) for k, v in slm_sd.items()}
        o_total, _, _ = ref_cpu.slm_forward(sd, v_s, v_l, v_a, mask, ms, ml)
        o_total.backward()
    for k in ("l_ce_s", "l_ce_l", "l_cont_s", "l_cont_l", "nce"):
        assert abs(float(d[k]) - float(ref_d[k])) < 1e-3 * max(1.0, abs(float(ref_d[k]))), k
    assert abs(total.item() - o_total.item()) < 1e-4 * abs(o_total.item())
    worst = 0.0
    for k, p in m.named_parameters():
        g_o = sd[k].grad
        if g_o is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        rel = (p.grad.cpu() - g_o).abs().max().item() / max(g_o.abs().max().item(), 1e-8)
        worst = max(worst, rel)
        assert rel < 1e-3, (k, rel)
    print("SLM training forward: worst relative gradient error vs autograd over the oracle %.2e" % worst)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    src = torch.cat([v_s, v_a], dim=-1)
    batch = (src, v_l, [40, 33, 12], None, None)
    opt = torch.optim.AdamW([p for _, p in T.slm_trainable_parameters(m)], lr=1e-4)
    with torch.enable_grad():
        first = x_engine_pt.train_epoch(m, [batch], opt, dev, clip=1.0, log=lambda *_: None)
        for _ in range(3):
            last = x_engine_pt.train_epoch(m, [batch], opt, dev, clip=1.0, log=lambda *_: None)
    assert last < first, (first, last)
    after = m.state_dict()
    for k in before:
        changed = not torch.equal(before[k], after[k])
        if k.startswith(T.SLM_FROZEN_PREFIXES):
            assert not changed, k
        elif k.startswith(("encoder_l.attn_layers", "decoder_joint.net.attn_layers", "speaker_vq.decoder.decoder_transformer")):
            assert changed, k
