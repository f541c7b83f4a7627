"""CPU: the oracle (oracle/ref_cpu.py) against the golden vectors captured from the
imported reference VQ-VAE (tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu

torch.set_grad_enabled(False)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("T", [5, 27, 300])
def test_encode_indices_bit_exact(golden_dir, vq_sd, T):
    g = _load(golden_dir, "vq_encode_T%d.npz" % T)
    x = torch.from_numpy(g["x"])
    idx, z, d = ref_cpu.vq_encode(vq_sd, x, "listener_vq.", return_all=True)
    assert np.array_equal(idx.view(-1).numpy(), g["idx"].astype(np.int64))
    assert g["margin"].min() >= 1e-4
    assert np.abs(z[0].numpy() - g["z"]).max() < 2e-5


@pytest.mark.parametrize("B,L", [(1, 26), (3, 26), (1, 299), (3, 299)])
def test_decode_matches_reference(golden_dir, vq_sd, B, L):
    g = _load(golden_dir, "vq_decode_B%d_L%d.npz" % (B, L))
    idx = torch.from_numpy(g["idx"].astype(np.int64))
    out = ref_cpu.vq_decode(vq_sd, idx, "listener_vq.")
    assert np.abs(out.numpy() - g["out"]).max() < 1e-5
    if B == 3:  # the PE batch-row quirk: identical tokens decode differently per batch row
        same = ref_cpu.vq_decode(vq_sd, idx[:1].repeat(3, 1), "listener_vq.")
        assert (same[0] - same[1]).abs().max() > 1e-3


def test_forward_vq_ragged(golden_dir, vq_sd):
    g = _load(golden_dir, "vq_forward_vq_ragged.npz")
    v_s, v_l = torch.from_numpy(g["v_speaker"]), torch.from_numpy(g["v_listener"])
    B, T, _ = v_s.shape
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(g["lens"]):
        mask[j, :n] = True
    zs, zl = ref_cpu.forward_vq(vq_sd, v_s, v_l, mask)
    assert np.array_equal(zs.numpy(), g["z_speaker"].astype(np.int64))
    assert np.array_equal(zl.numpy(), g["z_listener"].astype(np.int64))
    assert (zl[2, 12:] == -100).all() and (zs[3, 5:] == 0).all()


def test_roundtrip_c1(golden_dir, vq_sd):
    g = _load(golden_dir, "vq_roundtrip_C1.npz")
    x = torch.from_numpy(g["x"])
    idx = ref_cpu.vq_encode(vq_sd, x, "listener_vq.")
    assert np.array_equal(idx.numpy(), g["idx"].astype(np.int64))
    xhat = ref_cpu.vq_decode(vq_sd, idx, "listener_vq.")
    assert np.abs(xhat.numpy() - g["xhat"]).max() < 1e-5


def test_encode_batched_pe_rows(golden_dir, vq_sd):
    g = _load(golden_dir, "vq_encode_B3_T27.npz")
    idx = ref_cpu.vq_encode(vq_sd, torch.from_numpy(g["x"]), "listener_vq.")
    assert np.array_equal(idx.numpy(), g["idx"].astype(np.int64))


def test_sampler_matches_multinomial_fixture(golden_dir):
    g = _load(golden_dir, "sampler_multinomial.npz")
    ids = ref_cpu.sample_tokens(torch.from_numpy(g["logits"]), torch.from_numpy(g["noise"]))
    assert np.array_equal(ids.numpy(), g["ids"].astype(np.int64))
