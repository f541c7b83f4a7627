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
