"""GPU: edge cases and error behaviour of the C-ABI (minimum sizes, single frames, maximum context, misuse).
The reference raises Python exceptions on misuse; the library returns a negative status + dimx_last_error(),
which the host layer turns into DimxError."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def eng(full_sd):
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
    e.load_state_dict(full_sd)
    return e


def _inputs(B, T, seed=13):
    from dimx import prng
    return (torch.from_numpy(prng.normal(seed, "edge.vs", (B, T, 56))).cuda(),
            torch.from_numpy(prng.normal(seed, "edge.vl", (B, T, 56))).cuda(),
            torch.from_numpy(prng.normal(seed, "edge.va", (B, T, 768))).cuda())


def test_minimum_sequence_T3_matches_oracle(eng, full_sd):
    """T = 3 is the shortest clip the REFERENCE can process: the listener VQ-VAE decodes T-1 frames and its
    InstanceNorm1d refuses a single frame (torch: "Expected more than 1 spatial element"); the loader keeps
    clips of >= 5 frames (code/dataset/data_loader.py:122)."""
    from oracle import ref_cpu
    T = 3
    v_s, v_l, v_a = _inputs(2, T)
    mask = torch.ones(2, T, dtype=torch.bool)
    ref_total, _, ref_pred, aux = ref_cpu.slmft_forward(full_sd, v_s.cpu(), v_l.cpu(), v_a.cpu(), mask, mode="train",
                                                        return_aux=True)
    m8 = mask.to(torch.uint8).cuda()
    z = eng.vq_encode(1, v_l, torch.tensor([T, T], dtype=torch.int32).cuda(), pe_mode=0)
    assert torch.equal(z.cpu().long(), aux["z_l"])
    eng.encode_ctx(v_s, v_a, m8, False)
    logits, _, amax = eng.decode_tf(z, m8, None)
    assert tuple(logits.shape) == (2, T - 1, 512) and (logits.cpu() - aux["logits"]).abs().max() < 2e-3
    pred = eng.vq_decode(1, amax, 0)
    if torch.equal(amax.cpu().long(), aux["tokens"]):
        assert (pred.cpu() - ref_pred).abs().max() < 1e-3
    eng.encode_ctx(v_s, v_a, m8, True)
    tok = eng.generate(z[:, 0].contiguous(), m8, T, 0.0)
    assert tuple(tok.shape) == (2, T - 1) and torch.equal(tok.cpu().long()[:, 0], aux["logits"][:, 0].argmax(-1))


def test_two_frame_clip_in_a_ragged_batch(eng, full_sd):
    """the shortest encodable clip (2 frames: InstanceNorm needs > 1) next to a full one."""
    from oracle import ref_cpu
    _, v_l, _ = _inputs(2, 12)
    lens = torch.tensor([12, 2], dtype=torch.int32)
    mask = torch.arange(12)[None, :] < lens[:, None]
    _, z_ref = ref_cpu.forward_vq(full_sd, v_l.cpu(), v_l.cpu(), mask, with_speaker=False)
    z = eng.vq_encode(1, v_l, lens.cuda(), pe_mode=0, pad_value=-100)
    assert torch.equal(z.cpu().long(), z_ref) and (z.cpu()[1, 2:] == -100).all()


def test_maximum_context_T2048_generates(full_sd):
    """max_seq_len = 2048 (positional table bound): one bf16 clip end to end, finite and in range."""
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
    e.load_state_dict(full_sd)
    v_s, v_l, v_a = _inputs(1, 2048)
    m8 = torch.ones(1, 2048, dtype=torch.uint8).cuda()
    z = e.vq_encode(1, v_l, None, pe_mode=1)
    e.encode_ctx(v_s, v_a, m8, True)
    tok = e.generate(z[:, 0].contiguous(), m8, 2048, 1.0, 52, None, seed=3)
    assert tuple(tok.shape) == (1, 2047) and int(tok.min()) >= 0 and int(tok.max()) < 512
    pred = e.vq_decode(1, tok, 0)
    assert torch.isfinite(pred).all()
    with pytest.raises(lib.DimxError, match="workspace_bytes"):
        e.workspace(1, 2049)


def test_misuse_is_reported_not_crashed(eng):
    from dimx import lib
    v_s, v_l, v_a = _inputs(2, 8)
    m8 = torch.ones(2, 8, dtype=torch.uint8).cuda()
    z = eng.vq_encode(1, v_l, None, pe_mode=1)
    # decode without a context of the same shape / layout
    eng.encode_ctx(v_s, v_a, m8, True)
    with pytest.raises(lib.DimxError, match="encode_ctx"):
        eng.decode_tf(z, m8, None)
    with pytest.raises(lib.DimxError, match="n_samples"):
        eng.generate(z[:, 0].contiguous(), m8, 8, 1.0, n_samples=3)
    # too small / misaligned workspace, null pointers
    ws, wsb = eng.workspace(2, 8)
    st = eng._s()
    idx = torch.empty(2, 8, dtype=torch.int32, device="cuda")
    rc = eng.lib.dimx_vq_encode(eng.h, 1, lib.ptr(v_l), None, 2, 8, 1, 0, -100, lib.ptr(idx), None, ws, 1024, st)
    assert rc != 0 and b"workspace" in eng.lib.dimx_last_error()
    rc = eng.lib.dimx_vq_encode(eng.h, 1, lib.ptr(v_l), None, 2, 8, 1, 0, -100, lib.ptr(idx), None,
                                ctypes.c_void_p(ws.value + 8), wsb - 8, st)
    assert rc != 0 and b"aligned" in eng.lib.dimx_last_error()
    rc = eng.lib.dimx_vq_encode(eng.h, 1, None, None, 2, 8, 1, 0, -100, lib.ptr(idx), None, ws, wsb, st)
    assert rc != 0 and b"null" in eng.lib.dimx_last_error()
    rc = eng.lib.dimx_vq_encode(eng.h, 2, lib.ptr(v_l), None, 2, 8, 1, 0, -100, lib.ptr(idx), None, ws, wsb, st)
    assert rc != 0
    # a handle without weights refuses to run instead of computing on garbage
    from dimx import engine
    empty = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
    assert empty.missing_weights() > 0
    with pytest.raises(lib.DimxError, match="not loaded"):
        empty.vq_encode(1, v_l, None, pe_mode=1)
    # wrong-shaped tensor under a known key
    with pytest.raises(lib.DimxError, match="expected"):
        empty.load_state_dict({"norm_s.weight": torch.zeros(7)})
    with pytest.raises(lib.DimxError, match="unknown weight key"):
        empty.load_state_dict({"definitely.not.a.key": torch.zeros(3)})
