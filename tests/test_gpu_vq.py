"""GPU: the VQ-VAE stages (dimx_vq_encode / dimx_vq_argmin / dimx_vq_decode) through the C-ABI against
the golden vectors captured from the reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def eng(full_sd):
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
    e.load_state_dict(full_sd)
    assert e.missing_weights() == 0
    return e


@pytest.fixture(scope="module")
def eng_bf16(full_sd):
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
    e.load_state_dict(full_sd)
    return e


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("T", [5, 27, 300, 1500])
def test_encode_indices_bit_exact(eng, golden_dir, T):
    g = _g(golden_dir, "vq_encode_T%d.npz" % T)
    x = torch.from_numpy(g["x"]).cuda()
    idx, z = eng.vq_encode(1, x, return_z=True)
    assert np.array_equal(idx.cpu().numpy().reshape(-1), g["idx"].astype(np.int32)), \
        "%d / %d indices differ" % ((idx.cpu().numpy().reshape(-1) != g["idx"]).sum(), T)
    if g["z"].size:
        assert np.abs(z.cpu().numpy()[0] - g["z"]).max() < 1e-4


def test_argmin_kernel_matches_oracle(eng, vq_sd):
    from oracle import ref_cpu
    gen = torch.Generator().manual_seed(11)
    z = torch.randn(1003, 128, generator=gen) * 0.6
    E = vq_sd["listener_vq.quantize.embedding.weight"]
    ref_idx, d = ref_cpu.vq_quantize(z, E)
    margin = ref_cpu.vq_margins(d)
    idx, bd, mg = eng.vq_argmin(1, z.cuda(), with_stats=True)
    safe = margin > 1e-4
    assert torch.equal(idx.cpu().long()[safe], ref_idx[safe])
    assert (idx.cpu().long() == ref_idx).float().mean() > 0.999
    assert (bd.cpu() - d.min(1).values).abs().max() < 1e-3
    assert (mg.cpu() - margin).abs().max() < 1e-3


def test_encode_batched_pe_rows(eng, golden_dir):
    g = _g(golden_dir, "vq_encode_B3_T27.npz")
    idx = eng.vq_encode(1, torch.from_numpy(g["x"]).cuda(), pe_mode=1)
    assert np.array_equal(idx.cpu().numpy(), g["idx"].astype(np.int32))


def test_forward_vq_ragged(eng, golden_dir):
    g = _g(golden_dir, "vq_forward_vq_ragged.npz")
    lens = torch.from_numpy(g["lens"]).cuda()
    zl = eng.vq_encode(1, torch.from_numpy(g["v_listener"]).cuda(), lens, pe_mode=0, pad_value=-100)
    zs = eng.vq_encode(0, torch.from_numpy(g["v_speaker"]).cuda(), lens, pe_mode=0, pad_value=0)
    assert np.array_equal(zl.cpu().numpy(), g["z_listener"].astype(np.int32))
    assert np.array_equal(zs.cpu().numpy(), g["z_speaker"].astype(np.int32))


@pytest.mark.parametrize("B,L", [(1, 26), (3, 26), (1, 299), (3, 299)])
def test_decode_matches_reference(eng, golden_dir, B, L):
    g = _g(golden_dir, "vq_decode_B%d_L%d.npz" % (B, L))
    out = eng.vq_decode(1, torch.from_numpy(g["idx"].astype(np.int32)).cuda())
    err = np.abs(out.cpu().numpy() - g["out"]).max()
    assert err < 1e-4, "decode err %g (contract 1e-4)" % err


def test_decode_row_offset_reproduces_sharded_batch(eng, golden_dir):
    g = _g(golden_dir, "vq_decode_B3_L26.npz")
    idx = torch.from_numpy(g["idx"].astype(np.int32)).cuda()
    for b in range(3):  # a shard holding only clip b, decoded with batch_row_offset=b
        out = eng.vq_decode(1, idx[b:b + 1].contiguous(), row_offset=b)
        assert np.abs(out.cpu().numpy()[0] - g["out"][b]).max() < 1e-4


def test_roundtrip_c1(eng, golden_dir):
    g = _g(golden_dir, "vq_roundtrip_C1.npz")
    idx = eng.vq_encode(1, torch.from_numpy(g["x"]).cuda(), pe_mode=1)
    assert np.array_equal(idx.cpu().numpy(), g["idx"].astype(np.int32))
    xhat = eng.vq_decode(1, idx)
    assert np.abs(xhat.cpu().numpy() - g["xhat"]).max() < 1e-4


def test_bf16_mode_report(eng_bf16, golden_dir):
    """perf mode is not bit-exact: index agreement and decode error are held at their measured levels."""
    g = _g(golden_dir, "vq_roundtrip_C1.npz")
    idx = eng_bf16.vq_encode(1, torch.from_numpy(g["x"]).cuda(), pe_mode=1)
    agree = (idx.cpu().numpy() == g["idx"]).mean()
    xhat = eng_bf16.vq_decode(1, torch.from_numpy(g["idx"].astype(np.int32)).cuda())
    err = np.abs(xhat.cpu().numpy() - g["xhat"]).max()
    print("bf16 mode: index agreement %.3f, decode max err %.4f" % (agree, err))
    # measured on MI355X (round 2): agreement 0.987, decode error 0.0121 -- asserted with a small margin
    assert agree >= 0.975 and err <= 0.03
