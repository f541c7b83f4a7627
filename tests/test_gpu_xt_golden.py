"""GPU twin of tests/test_oracle_xt_golden.py: the HIP path (f32 parity mode, through the C-ABI) against fixtures
written from the real x-transformers library; skips while the fixtures are absent."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_hip_slmft_stages_match_the_library(golden_dir, full_sd):
    p = os.path.join(golden_dir, "xt_slmft.npz")
    if not os.path.exists(p):
        pytest.skip("xt_slmft.npz not present (tools/verify_against_xtransformers.py --write-golden)")
    from dimx import engine, lib
    from test_oracle_xt_golden import _slmft_inputs
    g = np.load(p)
    v_s, v_a, z, mask, noise, lens = _slmft_inputs(g)
    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
    e.load_state_dict(full_sd)
    m8 = mask.to(torch.uint8).cuda()
    x_s = e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, False, return_x_s=True).cpu()
    for b, n in enumerate(lens):
        assert np.abs(x_s[b, :n].numpy() - g["x_s"][b, :n]).max() < 1e-4
    logits, row_loss, _ = e.decode_tf(z.cuda(), m8, torch.from_numpy(g["kv_mask"]).to(torch.uint8).cuda())
    valid = mask[:, 1:].numpy()
    assert np.abs(logits.cpu().numpy() - g["tf_logits"])[valid].max() < 1e-3
    n_valid = (z[:, 1:] != -100).sum()
    assert abs((row_loss.cpu().sum() / n_valid).item() - float(g["tf_loss"])) < 1e-4 * max(1.0, abs(float(g["tf_loss"])))
    start = z[:, 0].clamp(min=0).cuda()
    T = z.shape[1]
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    assert np.array_equal(e.generate(start, m8, T, 0.0).cpu().numpy(), g["gen_greedy"])
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    assert np.array_equal(e.generate(start, m8, T, 1.0, 52, noise.cuda()).cpu().numpy(), g["gen_sampled"])
