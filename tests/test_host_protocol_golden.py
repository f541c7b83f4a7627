"""CPU: the host side of the evaluation protocol against what the REFERENCE's own loops return.

tests/golden/host_protocol.npz was written by tests/golden/make_golden.py --host-protocol, which lifts
``evaluate_test_epoch`` / ``evaluate_finetune_epoch`` (code/x_engine_pt.py:201-277) and ``pad_collate``
(code/dataset/data_loader.py:429-439) out of the reference by AST and runs them around tests/stub_model.StubSLMFT.
Here dimx.x_engine_pt / dimx.dataset.data_loader run around the same stub: batch protocol, best-of-10 selection by
Frechet distance (incl. the strict '<' and candidate order), per-clip cuts and list order must all coincide."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stub_model  # noqa: E402


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "host_protocol.npz"))


def _cat(lst):
    return np.concatenate([np.asarray(a) for a in lst], 0)


@pytest.mark.parametrize("batched", [True, False])
def test_evaluate_test_epoch_selects_what_the_reference_selects(gold, batched):
    from dimx import x_engine_pt
    yt, yp, xs, ids = x_engine_pt.evaluate_test_epoch(stub_model.StubSLMFT(), stub_model.protocol_batches(),
                                                      torch.device("cpu"), beam_size=10, batched_samples=batched)
    assert list(ids) == list(gold["test_ids"])
    assert [a.shape[0] for a in yp] == list(gold["test_pred_lens"]) == [a.shape[0] for a in yt]
    assert np.array_equal(_cat(yp), gold["test_pred"])          # the SAME candidate won for every clip, bit for bit
    assert np.allclose([float(a.astype(np.float64).sum()) for a in yt], gold["test_true_sum"], rtol=0, atol=1e-9)
    assert np.allclose([float(a.astype(np.float64).sum()) for a in xs], gold["test_x_sum"], rtol=0, atol=1e-9)


def test_evaluate_finetune_epoch_matches_the_reference(gold):
    from dimx import x_engine_pt
    yt, yp, xs, ids = x_engine_pt.evaluate_finetune_epoch(stub_model.StubSLMFT(), stub_model.protocol_batches(),
                                                          torch.device("cpu"))
    assert [a.shape[0] for a in yp] == list(gold["ft_pred_lens"])
    assert np.array_equal(_cat(yp), gold["ft_pred"])
    assert np.allclose([float(a.astype(np.float64).sum()) for a in yt], gold["ft_true_sum"], rtol=0, atol=1e-9)
    assert np.allclose([float(a.astype(np.float64).sum()) for a in xs], gold["ft_x_sum"], rtol=0, atol=1e-9)


def test_pad_collate_matches_the_reference(gold):
    from dimx.dataset import data_loader as dl
    xx, yy, lens, (sp, li), names = dl.pad_collate(stub_model.collate_items())
    assert np.array_equal(xx.numpy(), gold["coll_x"]) and np.array_equal(yy.numpy(), gold["coll_y"])
    assert list(lens) == list(gold["coll_lens"]) and list(names) == list(gold["coll_names"])
    assert np.array_equal(sp.numpy(), gold["coll_speaker"]) and np.array_equal(li.numpy(), gold["coll_listener"])


# ---------------------------------------------------------------------------------------------------------------------
# training / validation loops (tests/golden/train_protocol.npz: the reference's own loops around the stubs, plain SGD)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tgold(golden_dir):
    return np.load(os.path.join(golden_dir, "train_protocol.npz"))


def test_x_engine_pt_train_and_evaluate_epoch_leave_what_the_reference_leaves(tgold, capsys):
    """dimx.x_engine_pt.train_epoch (torch-optimiser route) / evaluate_epoch around the stub == reference
    code/x_engine_pt.py:9-60 / :134-165: same parameters after two epochs (SGD 0.05, clip 0.5, StepLR), same validation
    mean, same scheduler state."""
    from dimx import x_engine_pt
    m = stub_model.StubTrainPT()
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.9)
    dev = torch.device("cpu")
    with torch.enable_grad():
        for ep in range(2):
            m.train()
            x_engine_pt.train_epoch(m, stub_model.protocol_batches(), opt, dev, scheduler=sched, clip=0.5, print_freq=1, epoch=ep)
    assert np.allclose(m.w_v.detach().numpy(), tgold["pt_w_v"], rtol=0, atol=1e-7)
    assert np.allclose(m.w_a.detach().numpy(), tgold["pt_w_a"], rtol=0, atol=1e-7)
    assert abs(opt.param_groups[0]["lr"] - float(tgold["pt_lr"])) < 1e-15
    val = x_engine_pt.evaluate_epoch(m, stub_model.protocol_batches_with_ids(), dev, log=lambda *_: None)
    assert abs(val - float(tgold["pt_val"])) < 1e-6 and not m.training
    modes = [md for md, _ in m.seen]
    assert modes.count("train") == len(modes)             # both loops call the model with its default / 'train' mode
    mask = m.seen[0][1]
    assert mask.shape == (4, 96) and mask.sum(1).tolist() == [96, 80, 71, 64]


def test_x_engine_training_loops_leave_what_the_reference_leaves(tgold):
    """dimx.x_engine.train_epoch / train_continuous_epoch / evaluate_continuous_epoch == reference code/x_engine.py:8-62,
    90-105 around the stubs (SGD 0.1; clip 0.3 / none)."""
    from dimx import x_engine
    dev = torch.device("cpu")
    m = stub_model.StubTrainLegacy()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for ep in range(2):
        x_engine.train_epoch(m, stub_model.legacy_batches(), opt, dev, clip=0.3, print_freq=2, epoch=ep)
    assert np.allclose(m.w.detach().numpy(), tgold["lg_w"], rtol=0, atol=1e-7)
    assert np.allclose(m.emb.detach().numpy(), tgold["lg_emb"], rtol=0, atol=1e-7)
    _, sid, lid = m.calls[0]
    assert sid is None and lid.tolist() == [0, 2, 4]
    m = stub_model.StubTrainContinuous()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for ep in range(2):
        x_engine.train_continuous_epoch(m, stub_model.legacy_batches(), opt, dev, clip=0.0, print_freq=2, epoch=ep)
    assert np.allclose(m.w.detach().numpy(), tgold["ct_w"], rtol=0, atol=1e-7)
    val = x_engine.evaluate_continuous_epoch(m, [b[:4] for b in stub_model.legacy_batches()], dev, verbose=False)
    assert abs(val - float(tgold["ct_val"])) < 1e-6


def test_legacy_loop_prints_the_reference_lines(tgold, capsys):
    """the running-mean lines of code/x_engine.py:32-36 (every print_freq batches, mean since the last print)."""
    from dimx import x_engine
    m = stub_model.StubTrainLegacy()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    x_engine.train_epoch(m, stub_model.legacy_batches(), opt, torch.device("cpu"), clip=0.3, print_freq=2, epoch=0)
    got = [l for l in capsys.readouterr().out.splitlines() if l.startswith("Epoch: [0]")]
    want = [str(l) for l in tgold["printed"] if str(l).startswith("Epoch: [0]")][:len(got)]
    assert got == want and len(got) == 3


def test_device_frechet_backend_agrees_with_the_reference_arithmetic(gold):
    """dimx.metrics.frechet_distances_torch (batched float64 torch, eigenvalue form) == the reference's numpy / scipy arithmetic
    to 1e-6 relative on full-rank clips (the reference's np.mean of float32 frames is a float32 mean: 7e-9 observed), and evaluate_test_epoch(fd_backend="device") selects the candidates the reference's
    own loop selected (tests/golden/host_protocol.npz)."""
    from dimx import metrics, x_engine_pt
    g = torch.Generator().manual_seed(3)
    B, S, L, F = 5, 4, 90, 56
    lens = [90, 77, 64, 90, 58]
    yt = torch.randn(B, L, F, generator=g)
    yp = 0.6 * yt[:, None] + 0.5 * torch.randn(B, S, L, F, generator=g)
    fd = metrics.frechet_distances_torch(yt, yp, lens)
    for j in range(B):
        for s_i in range(S):
            ref = metrics.clip_fd(yt[j, :lens[j]].numpy(), yp[j, s_i, :lens[j]].numpy())
            assert abs(float(fd[j, s_i]) - ref) <= 1e-6 * abs(ref), (j, s_i, float(fd[j, s_i]), ref)
    yt_, yp_, xs, ids = x_engine_pt.evaluate_test_epoch(stub_model.StubSLMFT(), stub_model.protocol_batches(), torch.device("cpu"),
                                                        beam_size=10, fd_backend="device")
    assert list(ids) == list(gold["test_ids"])
    assert np.array_equal(_cat(yp_), gold["test_pred"])
