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
