"""CPU: the SLM pre-training forward restatement (oracle/ref_cpu.py).  PARITY UNPINNED (the
x-transformers stage has no reference fixture here); held to structural properties of the reference code:
padding invariance, the joint pass really is the 2T concatenation, masked frames are what is predicted."""
import pytest
import torch

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def slm_sd():
    import dimx  # noqa: F401
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
    return v_s, v_l, v_a, mask, ref_cpu.slm_random_masks(mask, ratio, g), ref_cpu.slm_random_masks(mask, ratio, g)


def test_random_masks_follow_reference_counts():
    from oracle import ref_cpu
    mask = torch.zeros(3, 20, dtype=torch.bool)
    for j, n in enumerate([20, 13, 5]):
        mask[j, :n] = True
    m = ref_cpu.slm_random_masks(mask, 0.15, torch.Generator().manual_seed(0))
    assert m.sum(1).tolist() == [int(20 * 0.15), int(13 * 0.15), int(5 * 0.15)]
    assert not (m & ~mask).any()


def test_padding_and_masked_inputs_do_not_leak(slm_sd):
    from oracle import ref_cpu
    v_s, v_l, v_a, mask, ms, ml = _case(2, 16, [16, 9])
    t0, d0, _, a0 = ref_cpu.slm_forward(slm_sd, v_s, v_l, v_a, mask, ms, ml, return_aux=True)
    v_s2, v_l2, v_a2 = v_s.clone(), v_l.clone(), v_a.clone()
    v_s2[1, 9:] = 5.0
    v_l2[1, 9:] = -2.0
    v_a2[1, 9:] = 3.0
    v_s2[ms] = 11.0          # masked frames are zeroed before the encoders: their content cannot matter there
    t1, d1, _, a1 = ref_cpu.slm_forward(slm_sd, v_s2, v_l2, v_a2, mask, ms, ml, return_aux=True)
    assert (a0["x_s"][1, :9] - a1["x_s"][1, :9]).abs().max() < 1e-5
    assert (a0["x_joint"][0] - a1["x_joint"][0]).abs().max() < 1e-5
    assert abs(float(d0["nce"]) - float(d1["nce"])) < 1e-5


def test_host_module_surface(slm_sd):
    import dimx  # noqa: F401
    from dimx import lib
    from dimx.seq2seq_pretrain import SLM
    m = SLM()
    sd = m.state_dict()
    assert set(sd) == set(slm_sd)
    assert sd["decoder_joint.net.pos_emb.emb.weight"].shape == (2048, 1152)
    assert sd["encoder_l.project_in.weight"].shape == (384, 56)
    nce, acc = SLM.forward_contrastive(torch.eye(4)[:, None, :].repeat(1, 3, 1), torch.eye(4)[:, None, :].repeat(1, 3, 1),
                                       torch.ones(4, 3, dtype=torch.bool))
    assert float(acc) == 1.0 and float(nce) < 1e-6
    with pytest.raises(lib.DimxError):
        m(torch.zeros(1, 8, 56), torch.zeros(1, 8, 56), torch.zeros(1, 8, 768), torch.ones(1, 8, dtype=torch.bool))
