"""CPU: the host-side ends of the evaluation path: loader batch format, pad_collate,
print_metrics / print_metrics_full against numbers captured from the reference's own mymetrics.py
(tests/golden/mymetrics_small.json, written by tests/golden/make_golden.py --mymetrics)."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import torch

import dimx  # noqa: F401
from dimx import mymetrics, prng
from dimx.dataset import data_loader as dl

SEED = 20260928


def test_print_metrics_match_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "mymetrics_small.json")))
    lens, exp = g["lens"], g["expected"]
    gts = [prng.normal(SEED, "golden.mm.gt%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    prs = [a + 0.3 * prng.normal(SEED, "golden.mm.pr%d" % i, a.shape) for i, a in enumerate(gts)]
    xs = [prng.normal(SEED, "golden.mm.x%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    buf = io.StringIO()
    with redirect_stdout(buf):
        ret = mymetrics.print_metrics(gts, prs, xs)
        mymetrics.print_metrics_full(gts, prs, xs)
    got = {}
    for line in buf.getvalue().strip().splitlines():
        k, v = line.split(":")
        got[k.strip()] = [float(t) for t in v.split()]
    assert set(got) == set(exp) - {"return"}             # same labels, same order of magnitude of output
    for k, v in got.items():
        assert np.allclose(v, exp[k], rtol=1e-6, atol=1e-9), (k, v, exp[k])
    assert np.allclose(ret, exp["return"], rtol=1e-6)


def test_print_metrics_match_reference_on_a_full_evaluation_batch(golden_dir):
    """SURVEY Appendix C's metrics_256.npz: 256 ragged clips (20..299 frames), every scalar print_metrics / print_metrics_full of the
    imported reference printed for them (tests/golden/make_golden.py --metrics-256) and print_metrics' return value."""
    g = np.load(os.path.join(golden_dir, "metrics_256.npz"))
    lens = [int(v) for v in g["lens"]]
    gts = [prng.normal(SEED, "golden.m256.gt%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    prs = [0.6 * a + 0.5 * prng.normal(SEED, "golden.m256.pr%d" % i, a.shape) for i, a in enumerate(gts)]
    xs = [prng.normal(SEED, "golden.m256.x%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    buf = io.StringIO()
    with redirect_stdout(buf):
        ret = mymetrics.print_metrics(gts, prs, xs)
        mymetrics.print_metrics_full(gts, prs, xs)
    got = {}
    for line in buf.getvalue().strip().splitlines():
        k, v = line.split(":")
        got[k.strip()] = [float(t) for t in v.split()]
    exp = {str(k): [x for x in row if not np.isnan(x)] for k, row in zip(g["labels"], g["values"])}
    assert list(got) == [str(k) for k in g["labels"]]          # same labels in the same order
    for k, v in got.items():
        assert np.allclose(v, exp[k], rtol=1e-6, atol=1e-9), (k, v, exp[k])
    assert np.allclose(ret, g["ret"], rtol=1e-6)


def test_pad_collate_and_synthetic_loader():
    batch = []
    for i, n in enumerate([7, 12, 5]):
        batch.append((torch.from_numpy(prng.normal(SEED, "pc.x%d" % i, (n, 824))),
                      torch.from_numpy(prng.normal(SEED, "pc.y%d" % i, (n, 56))), "clip%d" % i, i, 2 * i, i % 3))
    xx, yy, x_lens, (sid, lid), names = dl.pad_collate(batch)
    assert tuple(xx.shape) == (3, 12, 824) and tuple(yy.shape) == (3, 12, 56) and x_lens == [7, 12, 5]
    assert sid.tolist() == [0, 1, 2] and lid.tolist() == [0, 2, 4] and names == ["clip0", "clip1", "clip2"]
    assert float(xx[0, 7:].abs().sum()) == 0.0 and torch.equal(xx[2, :5], batch[2][0])
    loaders = dl.get_vico_dataloaders(4, synthetic={"n_clips": 6, "max_len": 40, "min_len": 5})
    assert set(loaders) == {"train", "valid", "all"}
    src, tgt, src_len, (s_ids, l_ids), ids = next(iter(loaders["valid"]))
    assert src.shape[0] == 4 and src.shape[2] == 824 and tgt.shape[2] == 56 and len(src_len) == 4
    assert torch.all(src[0, :src_len[0], :56] == 1.0)     # the ViCo protocol's constant speaker stream
    assert len(loaders["all"].dataset) == 12


def test_vico_dataset_reads_reference_pickles(tmp_path):
    """write two clips in the reference's on-disk format and read them back through ViCoDataset."""
    import pickle
    import pandas as pd
    rows = []
    for i, (cid, split) in enumerate((("a01", "test"), ("b02", "train"), ("c03", "test"))):
        n = 9 + i
        d = {"video_speaker": np.full((n, 56), 3.0, np.float32), "audio": np.ones((n, 768), np.float32) * i,
             "video_listener": np.arange(n * 56, dtype=np.float32).reshape(n, 56)}
        if cid != "c03":                                   # c03 is listed but missing on disk -> skipped
            with open(tmp_path / (cid + ".pkl"), "wb") as f:
                pickle.dump(d, f)
        rows.append(["positive", cid, 0, 0, 10 + i, 20 + i, split])
    pd.DataFrame(rows).to_csv(tmp_path / "meta.csv", index=False)
    ds = dl.ViCoDataset(str(tmp_path), str(tmp_path / "meta.csv"), mode="test")
    assert len(ds) == 1
    x, y, path, spk, lst, sent = ds[0]
    assert tuple(x.shape) == (9, 824) and torch.all(x[:, :56] == 1.0) and torch.all(x[:, 56:] == 0.0)
    assert y[1, 0].item() == 56.0 and spk == 20 and lst == 10 and sent == 1 and path.endswith("a01.pkl")


def test_postprocess_matches_reference(golden_dir, tmp_path):
    from dimx import postprocess2emoca as pp
    x = prng.normal(SEED, "golden.post.x", (37, 56)).astype(np.float64)
    y = pp.smooth_logits_matrix(x)
    ref = np.load(os.path.join(golden_dir, "postprocess_smooth.npz"))["y"]
    assert np.allclose(y, ref, rtol=0, atol=1e-12)
    assert (y[:5] == 0).all() and (y[-4:] == 0).all() and (y[5] != 0).any()
    data = {"y_pred": [x.astype(np.float32)], "y_true": [x.astype(np.float32)], "data_ids": ["/a/b/clip7.pkl"]}
    n = pp.export_predictions(data, str(tmp_path / "p"), str(tmp_path / "g"))
    assert n == 37
    pose = np.load(tmp_path / "p" / "clip7" / "10" / "pose.npy")
    exp = np.load(tmp_path / "g" / "clip7" / "10" / "exp.npy")
    assert pose.shape == (6,) and exp.shape == (50,) and np.allclose(pose, y[10, :6], atol=1e-6)


def test_example_driver_compiles():
    """examples/test_s2s_pretrain.py (the reference driver on the drop-ins) is valid Python and names only
    modules that exist in the package."""
    import importlib
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    py_compile.compile(os.path.join(root, "examples", "test_s2s_pretrain.py"), doraise=True)
    for mod in ("dimx.dataset.data_loader", "dimx.mymetrics", "dimx.seq2seq_pretrain", "dimx.x_engine_pt",
                "dimx.x_engine", "dimx.seq2seq", "dimx.postprocess2emoca"):
        importlib.import_module(mod)
