"""CPU: the legacy ListenerGenerator restatement (oracle/ref_cpu.py).

Pinned part: the speaker VQ-VAE -> x_speaker construction against the fixture produced by the reference's own
VQSpeakerAutoEncoder (tests/golden/make_golden.py --legacy).  The x-transformers stage of the legacy generator is
PARITY UNPINNED (library absent from /root/reference) and is held to the same self-consistency properties as
the SLMFT stage: cached generation == teacher-forced logits at the sampled prefix, padding invariance."""
import os

import numpy as np
import pytest
import torch

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def legacy_sd():
    import dimx  # noqa: F401
    from dimx import weights
    return weights.synth_state_dict(weights.listener_generator_spec(), 20260928)


def _case(B, T, lens, seed=5):
    from dimx import prng
    v_s = torch.from_numpy(prng.normal(seed, "legacy.vs", (B, T, 824)))
    v_l = torch.from_numpy(prng.normal(seed, "legacy.vl", (B, T, 56)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    return v_s, v_l, mask


def test_speaker_features_match_reference_fixture(golden_dir, legacy_sd):
    from oracle import ref_cpu
    g = np.load(os.path.join(golden_dir, "legacy_speaker_features.npz"))
    v = torch.from_numpy(g["v_speaker"])
    B, T, _ = v.shape
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(g["lens"]):
        mask[j, :n] = True
    x = ref_cpu.legacy_speaker_features(legacy_sd, v, mask)
    assert np.abs(x.numpy() - g["x_speaker"]).max() < 1e-6
    for j, n in enumerate(g["lens"]):
        _, idx, _ = ref_cpu.speaker_vq_encode_quant(legacy_sd, v[j][mask[j]].unsqueeze(0))
        assert np.array_equal(idx.view(-1).numpy(), g["idx"][j, :n * 8].astype(np.int64))
        assert (g["idx"][j, n * 8:] == -1).all()


def test_scramble_is_a_reinterpretation_not_a_transpose(legacy_sd):
    """x_speaker[b, t, j] = padded[b].flatten()[t*1024 + j] with padded[b] = [128, T*8] channel-major."""
    from oracle import ref_cpu
    v_s, _, mask = _case(2, 12, [12, 7])
    x = ref_cpu.legacy_speaker_features(legacy_sd, v_s, mask)
    E = legacy_sd["speaker_vq.quantize.embedding.weight"]
    for b, n in enumerate([12, 7]):
        _, idx, _ = ref_cpu.speaker_vq_encode_quant(legacy_sd, v_s[b][mask[b]].unsqueeze(0))
        pad = torch.zeros(128, 12 * 8)
        pad[:, :n * 8] = E[idx.view(-1)].t()
        assert torch.equal(x[b].reshape(-1), pad.reshape(-1))


def test_generate_equals_teacher_forced_prefix(legacy_sd):
    from dimx import prng
    from oracle import ref_cpu
    B, T, lens = 2, 14, [14, 9]
    v_s, v_l, mask = _case(B, T, lens, seed=3)
    noise = torch.from_numpy(prng.exponential(4, "legacy.noise", (T, B, 512)))
    z_pred, z_l = ref_cpu.listener_generator_generate(legacy_sd, v_s, v_l, mask, noise)
    assert tuple(z_pred.shape) == (B, T)
    x_speaker = ref_cpu.legacy_speaker_features(legacy_sd, v_s, mask)
    enc = ref_cpu.xt_encoder(legacy_sd, "generator.encoder.", x_speaker, mask, causal=False, depth=6, heads=8)
    seq = torch.cat([z_l[:, :1], z_pred], 1)                       # start + T generated
    logits = ref_cpu.legacy_decoder_logits(legacy_sd, seq[:, :-1], enc, mask)
    tok = ref_cpu.sample_tokens(logits.permute(1, 0, 2), noise)    # [T,B]
    assert torch.equal(tok.t(), z_pred)


def test_padding_never_influences_valid_outputs(legacy_sd):
    from oracle import ref_cpu
    v_s, v_l, mask = _case(2, 16, [16, 9], seed=6)
    _, _, a = ref_cpu.listener_generator_forward(legacy_sd, v_s, v_l, mask)
    v_s2, v_l2 = v_s.clone(), v_l.clone()
    v_s2[1, 9:] = 7.0
    v_l2[1, 9:] = -3.0
    _, _, b = ref_cpu.listener_generator_forward(legacy_sd, v_s2, v_l2, mask)
    assert torch.equal(a["z_l"], b["z_l"])
    assert (a["logits"][1, :8] - b["logits"][1, :8]).abs().max() < 1e-5
    assert torch.equal(a["logits"][0], b["logits"][0])


def test_host_module_surface():
    """state-dict keys / shapes of dimx.seq2seq.ListenerGenerator follow the reference naming; no CPU path."""
    import dimx  # noqa: F401
    from dimx import lib, seq2seq, weights, x_engine
    m = seq2seq.ListenerGenerator()
    sd = m.state_dict()
    spec = {n: tuple(s) for n, s, _, _ in weights.listener_generator_spec()}
    assert set(sd) == set(spec)
    for k, v in sd.items():
        assert tuple(v.shape) == spec[k], k
    assert sd["generator.decoder.net.pos_emb.emb.weight"].shape == (1024, 512)
    assert sd["speaker_vq.encoder.encoder_linear_embedding_post.net.weight"].shape == (1024, 768)
    assert m.speaker_face_quan_num == 8 and m.speaker_zquant_dim == 128
    with pytest.raises(lib.DimxError):
        m(torch.zeros(1, 4, 824), torch.zeros(1, 4, 56), torch.ones(1, 4, dtype=torch.bool))
    p = x_engine.TokenPerplexity()
    logits = torch.zeros(1, 5, 512)
    p.update(logits, torch.zeros(1, 5, dtype=torch.long))
    assert abs(p.compute() - 512.0) < 1e-6
