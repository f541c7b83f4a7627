"""CPU: the C-ABI library loads and exports every symbol include/dimx.h declares (no compute calls), the
config / state-dict surface matches the reference fixtures, the host metrics match the reference, and the
product path refuses to run without a GPU instead of falling back."""
import json
import os
import re

import numpy as np
import pytest
import torch

import dimx  # noqa: F401
from dimx import config, lib, metrics, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dimx.h")).read()
    declared = set(re.findall(r"\b(dimx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"dimx_ctx"}
    l = lib.load()
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    for name in declared:
        assert hasattr(l, name), name
    assert l.dimx_version() >= 100
    d = lib.default_dims()
    assert (d.vq_hidden, d.vq_heads, d.vq_zdim, d.dim, d.dim_a, d.heads, d.num_tokens) == (384, 8, 128, 384, 768, 12, 512)


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dimx.engine import Engine
    from dimx.seq2seq_pretrain import SLMFT
    with pytest.raises(lib.DimxError):
        Engine("cuda:0")
    m = SLMFT()
    with pytest.raises(lib.DimxError):
        m(torch.zeros(1, 4, 56), torch.zeros(1, 4, 56), torch.zeros(1, 4, 768), torch.ones(1, 4, dtype=torch.bool))


def test_checkpoints_of_other_xtransformers_releases_are_never_loaded_in_part():
    """SURVEY A.2 [XT?]: project_in.bias / to_logits.bias are adopted as parameters, a zero LayerNorm bias (the renamed `beta`
    buffer of code/finetune_s2s_pretrain.py:49-57) is dropped, a non-zero one is refused -- also under strict=False, which
    is how the reference loads its checkpoints."""
    from dimx.seq2seq_pretrain import SLMFT
    m = SLMFT()
    base = {k: v.clone() for k, v in m.state_dict().items()}
    sd = dict(base)
    sd["encoder_s.project_in.bias"] = torch.full((384,), 0.25)
    sd["decoder_joint.net.to_logits.bias"] = torch.arange(512, dtype=torch.float32)
    sd["encoder_joint.attn_layers.layers.0.0.0.bias"] = torch.zeros(384)
    sd["decoder_joint.net.attn_layers.final_norm.bias"] = torch.zeros(1152)
    v0 = m._weights_version()
    m.load_state_dict(sd, strict=False)
    own = m.state_dict()
    assert torch.equal(own["encoder_s.project_in.bias"], sd["encoder_s.project_in.bias"])
    assert torch.equal(own["decoder_joint.net.to_logits.bias"], sd["decoder_joint.net.to_logits.bias"])
    assert "encoder_joint.attn_layers.layers.0.0.0.bias" not in own and m._weights_version() != v0
    assert "decoder_joint.net.to_logits.bias" in dict(m.named_parameters())
    # strict load of the same dict works too (the zero LayerNorm biases are not "unexpected")
    m.load_state_dict(sd, strict=True)
    bad = dict(base)
    bad["encoder_s.attn_layers.final_norm.bias"] = torch.full((384,), 1e-3)
    for strict in (True, False):
        with pytest.raises(lib.DimxError):
            m.load_state_dict(bad, strict=strict)
    # a checkpoint without the optional tensors takes them away again
    m.load_state_dict(base, strict=True)
    assert set(m.state_dict()) == set(base)
    wrong = dict(base)
    wrong["encoder_s.project_in.bias"] = torch.zeros(56)
    with pytest.raises(lib.DimxError):
        m.load_state_dict(wrong, strict=False)
    assert set(m.state_dict()) == set(base)


def test_config_surface(golden_dir, tmp_path):
    cfg = config.load_cfg_from_cfg_file(config.DEFAULT_CONFIG)
    ref = json.load(open(os.path.join(golden_dir, "cfg_roundtrip.json")))["reference_flat_cfg"]
    for k in ("arch", "in_dim", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
              "quant_factor", "face_quan_num", "neg", "INaffine", "n_embed", "zquant_dim"):
        assert cfg[k] == ref[k] and getattr(cfg, k) == ref[k]
    new = config.merge_cfg_from_list(cfg, ["NETWORK.hidden_size", "512", "neg", "0.1"])
    assert new.hidden_size == 512 and new.neg == 0.1 and cfg.hidden_size == 384
    with pytest.raises(AssertionError):
        config.merge_cfg_from_list(cfg, ["nope", "1"])
    with pytest.raises(AssertionError):
        config.load_cfg_from_cfg_file(str(tmp_path / "x.txt"))


def test_state_dict_surface():
    from dimx.seq2seq_pretrain import SLMFT
    spec = weights.slmft_spec()
    n_params = sum(int(np.prod(s)) for n, s, k, _ in spec if k != "pe")
    assert 149e6 < n_params < 151e6   # ~150 M (VQ 23.26 M x2, encoders 10.4-10.5 M x3, decoder 72.0 M)
    m = SLMFT()
    sd = m.state_dict()
    assert list(sd.keys()).count("listener_vq.quantize.embedding.weight") == 1
    assert set(sd) == {n for n, *_ in spec}
    assert sd["decoder_joint.net.attn_layers.layers.4.1.to_k.weight"].shape == (768, 1152)
    assert sd["encoder_joint.project_in.weight"].shape == (384, 384)
    # strict load of a perturbed dict round-trips and bumps the version the engine watches
    v0 = m._weights_version()
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["norm_s.bias"] += 1.0
    m.load_state_dict(sd2, strict=True)
    assert m._weights_version() != v0 and torch.equal(m.state_dict()["norm_s.bias"], sd2["norm_s.bias"])


def test_prng_is_stable():
    from dimx import prng
    a = prng.uniform(1, "x", (4,), -1, 1)
    assert np.allclose(a, prng.uniform(1, "x", (4,), -1, 1)) and not np.allclose(a, prng.uniform(2, "x", (4,), -1, 1))
    # pinned values: a change of the generator silently invalidates every golden fixture
    assert prng.fnv1a64("abc") == 0xE71FA2190541574B
    assert abs(float(prng.uniform(20260928, "listener_vq.quantize.embedding.weight", (1,), -1, 1)[0]) -
               float(weights.synth_state_dict([("listener_vq.quantize.embedding.weight", (512, 128), "codebook", 0)])
                     ["listener_vq.quantize.embedding.weight"][0, 0])) == 0.0
    e = prng.exponential(3, "n", (10000,))
    assert e.min() > 0 and abs(e.mean() - 1.0) < 0.05


def test_metrics_match_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics_small.npz"))
    for i in range(3):
        gt, pr = g["gt%d" % i], g["pr%d" % i]
        assert abs(metrics.clip_fd(gt, pr) - g["fd"][i]) < 1e-6 * max(1, abs(g["fd"][i]))
        assert abs(metrics.calculate_variance(pr) - g["var"][i]) < 1e-6 * g["var"][i]
        assert abs(metrics.sts(gt, pr) - g["sts"][i]) < 1e-5 * g["sts"][i]
    s = metrics.summarize([g["gt0"], g["gt1"]], [g["pr0"], g["pr1"]])
    assert set(s) == {"pose", "exp"} and s["exp"]["fd"] > 0


def test_compact_by_mask_matches_reference_indexing():
    from dimx.seq2seq_pretrain import compact_by_mask
    x = torch.arange(2 * 6 * 3, dtype=torch.float32).view(2, 6, 3)
    mask = torch.tensor([[1, 1, 1, 0, 0, 0], [1, 0, 1, 1, 0, 1]], dtype=torch.bool)
    xc, lens = compact_by_mask(x, mask)
    assert lens.tolist() == [3, 4]
    for i in range(2):
        assert torch.equal(xc[i, :lens[i]], x[i][mask[i]])


def test_train_epoch_never_changes_backend_behind_the_caller():
    """for a model of this package train_epoch(backward='auto') lands on the HIP step or raises with the
    reason; the PyTorch-autograd restatement (dimx.train, the checker) runs only with backward='autograd'.  A foreign
    nn.Module has no HIP backend to leave: the reference's generic loop runs it on torch."""
    import torch
    from dimx import lib, x_engine_pt
    from dimx.seq2seq_pretrain import SLMFT
    m = SLMFT()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-5)
    with pytest.raises(lib.DimxError, match="not on a ROCm GPU"):
        x_engine_pt.train_epoch(m, [], opt, torch.device("cpu"), log=lambda *_: None)
    with pytest.raises(lib.DimxError, match="not on a ROCm GPU"):
        x_engine_pt.train_epoch(m, [], opt, torch.device("cpu"), log=lambda *_: None, backward="hip")

    class Foreign(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))
    f = Foreign()
    logs = []
    out = x_engine_pt.train_epoch(f, [], torch.optim.SGD(f.parameters(), lr=0.1), torch.device("cpu"), log=logs.append)
    assert out != out and any("not a dimx model" in ln for ln in logs)    # empty loader: nan mean, and the route was announced


def test_split_count_of_the_parity_modes_decode_gemms_depends_on_the_projection_only():
    """a rank's shard of a batch must reproduce the rows of the whole batch bit for bit in the f32 parity mode, and the
    number of split-K slabs decides the summation order -- so it may depend on (N, K) only, never on the rows M (host logic of
    csrc/gemm.hip gemm_plan_splits / csrc/gemm_x3.hip gemm_x3_plan, no GPU needed); at most 8 slabs (the consumers' limit)."""
    from dimx import lib as L
    lib = L.load()
    F32 = L.F32
    for N, K in ((2304, 1152), (1152, 768), (768, 1152), (1152, 4608), (512, 1152), (1536, 512), (512, 2048)):
        for flags in (5, 16 | 5):                     # exact-f32 MFMA kernel, split-bf16 kernel
            counts = {lib.dimx_op_gemm_slabs(F32, M, N, K, flags) for M in (1, 4, 32, 64, 128, 200, 256)}
            assert len(counts) == 1 and 1 <= next(iter(counts)) <= 8, (N, K, flags, counts)
        assert lib.dimx_op_gemm_slabs(F32, 256, N, K, 16 | 5 | (3 << 16)) == 3        # a forced count is honoured
