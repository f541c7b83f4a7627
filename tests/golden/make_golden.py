"""Generate the committed golden fixtures by importing the REFERENCE VQ-VAE
(/root/reference/code) in the build container.  The reference never travels to the
GPU box; only the small .npz files written here do.  Weights are not stored: both
sides regenerate them from dimx.prng (seed below).

Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"
sys.path.insert(0, ROOT)

import numpy as np
import torch

import dimx  # noqa: E402
from dimx import prng, weights  # noqa: E402
from oracle import ref_cpu  # noqa: E402

SEED = 20260928
torch.manual_seed(0)
torch.set_grad_enabled(False)


def load_reference():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from base import config as ref_config
        from models import get_model
        cfg = ref_config.load_cfg_from_cfg_file("./config.yaml")
        models = {}
        for which in ("listener_vq.", "speaker_vq."):
            m = get_model(cfg).eval()
            sd = weights.synth_state_dict(weights.vq_spec(prefix=which), SEED, strip_prefix=which)
            ref_keys = set(m.state_dict().keys())
            assert ref_keys == set(sd.keys()), (ref_keys ^ set(sd.keys()))
            for k, v in m.state_dict().items():
                assert tuple(v.shape) == tuple(sd[k].shape), k
            # the computed pe buffer must equal the reference's own buffer bit for bit
            assert torch.equal(m.state_dict()["encoder.encoder_pos_embedding.pe"],
                               sd["encoder.encoder_pos_embedding.pe"])
            m.load_state_dict(sd, strict=True)
            models[which] = m
        return cfg, models, ref_config
    finally:
        os.chdir(cwd)


def main():
    cfg, models, ref_config = load_reference()
    lst = models["listener_vq."]
    spk = models["speaker_vq."]
    full_sd = weights.synth_state_dict(weights.vq_spec(prefix="listener_vq.") +
                                       weights.vq_spec(prefix="speaker_vq."), SEED)
    report = {}

    # ---- config surface -------------------------------------------------
    my_cfg = dimx.config.load_cfg_from_cfg_file(dimx.config.DEFAULT_CONFIG)
    ref_cfg = dict(cfg)
    shared = {k: ref_cfg[k] for k in my_cfg if k in ref_cfg}
    assert all(my_cfg[k] == shared[k] for k in shared), "config.yaml deviates from the reference"
    with open(os.path.join(HERE, "cfg_roundtrip.json"), "w") as f:
        json.dump({"reference_flat_cfg": {k: ref_cfg[k] for k in sorted(ref_cfg)
                                          if isinstance(ref_cfg[k], (int, float, str, bool, type(None)))}},
                  f, indent=1, sort_keys=True)

    # ---- encode fixtures -------------------------------------------------
    for T in (5, 27, 300, 1500):
        x = torch.from_numpy(prng.normal(SEED, "golden.enc.x.T%d" % T, (1, T, 56)))
        quant, _, info = lst.encode(x)
        idx = info[2].view(-1)
        o_idx, o_z, o_d = ref_cpu.vq_encode(full_sd, x, "listener_vq.", return_all=True)
        margin = ref_cpu.vq_margins(o_d)
        h = lst.encoder(x)                      # reference pre-quant features
        assert torch.equal(idx, o_idx.view(-1)), "oracle indices differ from reference at T=%d" % T
        zerr = (h - o_z).abs().max().item()
        assert zerr < 2e-5, zerr
        assert margin.min().item() >= 1e-4, ("reseed: min top-2 margin", margin.min().item())
        np.savez_compressed(os.path.join(HERE, "vq_encode_T%d.npz" % T),
                            x=x.numpy(), idx=idx.numpy().astype(np.int16),
                            margin=margin.numpy().astype(np.float32),
                            z=h[0].numpy().astype(np.float32) if T <= 300 else np.zeros((0,), np.float32))
        report["encode_T%d" % T] = {"min_margin": margin.min().item(), "z_err_oracle_vs_ref": zerr}

    # ---- decode fixtures (PE batch-row quirk for B=3) ---------------------
    E = lst.quantize.embedding.weight
    for B, L in ((1, 26), (3, 26), (1, 299), (3, 299)):
        idx = torch.from_numpy(prng.integers(SEED, "golden.dec.idx.B%d.L%d" % (B, L), (B, L), 0, 512))
        out = lst.decode(E[idx].permute(0, 2, 1))
        o_out = ref_cpu.vq_decode(full_sd, idx, "listener_vq.")
        err = (out - o_out).abs().max().item()
        assert err < 1e-5, err
        np.savez_compressed(os.path.join(HERE, "vq_decode_B%d_L%d.npz" % (B, L)),
                            idx=idx.numpy().astype(np.int16), out=out.numpy().astype(np.float32))
        report["decode_B%d_L%d" % (B, L)] = {"err_oracle_vs_ref": err}

    # ---- forward_vq ragged re-enactment (code/seq2seq_pretrain.py:480-494) -----
    B, T = 4, 40
    lens = [40, 33, 12, 5]
    v_s = torch.from_numpy(prng.normal(SEED, "golden.fvq.vs", (B, T, 56)))
    v_l = torch.from_numpy(prng.normal(SEED, "golden.fvq.vl", (B, T, 56)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    zs, zl = [], []
    import torch.nn.functional as F
    for i in range(B):
        sf = spk.encode(v_s[i][mask[i]].unsqueeze(0))[2][2].squeeze()
        zs.append(F.pad(sf, (0, T - sf.shape[-1]), value=0))
        lf = lst.encode(v_l[i][mask[i]].unsqueeze(0))[2][2].squeeze()
        zl.append(F.pad(lf, (0, T - lf.shape[-1]), value=-100))
    zs, zl = torch.stack(zs), torch.stack(zl)
    o_zs, o_zl = ref_cpu.forward_vq(full_sd, v_s, v_l, mask)
    assert torch.equal(zs, o_zs) and torch.equal(zl, o_zl)
    np.savez_compressed(os.path.join(HERE, "vq_forward_vq_ragged.npz"), v_speaker=v_s.numpy(),
                        v_listener=v_l.numpy(), lens=np.array(lens, np.int32),
                        z_speaker=zs.numpy().astype(np.int16), z_listener=zl.numpy().astype(np.int16))

    # ---- C1 round trip (VQAutoEncoder.forward, stage1_BIWI.py:49-55) -------
    x = torch.from_numpy(prng.normal(SEED, "golden.c1.x", (1, 300, 56)))
    dec, _, info = lst(x)
    idx = info[2].view(1, -1)
    o_idx = ref_cpu.vq_encode(full_sd, x, "listener_vq.")
    o_dec = ref_cpu.vq_decode(full_sd, o_idx, "listener_vq.")
    assert torch.equal(idx, o_idx)
    err = (dec - o_dec).abs().max().item()
    assert err < 1e-5, err
    np.savez_compressed(os.path.join(HERE, "vq_roundtrip_C1.npz"), x=x.numpy(),
                        idx=idx.numpy().astype(np.int16), xhat=dec.numpy().astype(np.float32))
    report["roundtrip_C1"] = {"err_oracle_vs_ref": err}

    # ---- batched encode with the PE batch-row quirk (public encode API, B=3) ----
    x = torch.from_numpy(prng.normal(SEED, "golden.encB3.x", (3, 27, 56)))
    idx = lst.encode(x)[2][2].view(3, 27)
    o_idx, _, o_d = ref_cpu.vq_encode(full_sd, x, "listener_vq.", return_all=True)
    assert torch.equal(idx, o_idx)
    np.savez_compressed(os.path.join(HERE, "vq_encode_B3_T27.npz"), x=x.numpy(),
                        idx=idx.numpy().astype(np.int16),
                        margin=ref_cpu.vq_margins(o_d).numpy().astype(np.float32))

    # ---- sampler primitive: torch.multinomial == argmax(p / q) on CPU --------
    g = torch.Generator().manual_seed(1234)
    logits = torch.randn(16, 512, generator=g) * 3
    probs = torch.softmax(ref_cpu.top_k_filter(logits, 52), -1)
    g1 = torch.Generator().manual_seed(99)
    ids = torch.multinomial(probs, 1, generator=g1).view(-1)
    g2 = torch.Generator().manual_seed(99)
    q = torch.empty_like(probs).exponential_(1, generator=g2)
    ids2 = (probs / q).argmax(-1)
    assert torch.equal(ids, ids2), "multinomial != argmax(p/q) on this torch build"
    np.savez_compressed(os.path.join(HERE, "sampler_multinomial.npz"), logits=logits.numpy(),
                        probs=probs.numpy(), noise=q.numpy(), ids=ids.numpy().astype(np.int16))

    # ---- metrics (code/metrics/eval_utils.py) --------------------------------
    sys.path.insert(0, REF)
    cwd = os.getcwd(); os.chdir(REF)
    from metrics import eval_utils as ref_metrics
    os.chdir(cwd)
    rng = np.random.RandomState(7)
    gts = [rng.randn(n, 56).astype(np.float32) for n in (40, 55, 70)]
    prs = [g + 0.3 * rng.randn(*g.shape).astype(np.float32) for g in gts]
    fds, vars_, stss = [], [], []
    for gt, pr in zip(gts, prs):
        m1, s1 = ref_metrics.calculate_activation_statistics(gt)
        m2, s2 = ref_metrics.calculate_activation_statistics(pr)
        fds.append(ref_metrics.calculate_frechet_distance(m1, s1, m2, s2))
        vars_.append(ref_metrics.calculate_variance(pr))
        stss.append(ref_metrics.sts(gt, pr))
    np.savez_compressed(os.path.join(HERE, "metrics_small.npz"),
                        **{"gt%d" % i: g for i, g in enumerate(gts)},
                        **{"pr%d" % i: p for i, p in enumerate(prs)},
                        fd=np.array(fds, np.float64), var=np.array(vars_, np.float64),
                        sts=np.array(stss, np.float64))

    with open(os.path.join(HERE, "golden_report.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__" and not any(f in sys.argv for f in ("--legacy", "--mymetrics", "--metrics-256", "--postprocess", "--host-protocol",
                                                             "--train-protocol")):
    main()


def legacy_fixtures():
    """VQSpeakerAutoEncoder (arch stage1_BIWI_speaker, config_speaker_old.yaml) imported from the
    reference: the x_speaker construction of seq2seq.ListenerGenerator (per-sample encode, pad, raw view)."""
    import torch.nn.functional as F
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from base import config as ref_config
        from models import get_model
        cfg = ref_config.load_cfg_from_cfg_file("./config_speaker_old.yaml")
        spk = get_model(cfg).eval()
    finally:
        os.chdir(cwd)
    spec = weights.legacy_speaker_vq_spec(prefix="speaker_vq.")
    sd = weights.synth_state_dict(spec, SEED, strip_prefix="speaker_vq.")
    ref_sd = spk.state_dict()
    for k, v in sd.items():
        assert k in ref_sd and tuple(ref_sd[k].shape) == tuple(v.shape), k
    missing, unexpected = spk.load_state_dict(sd, strict=False)      # decoder_v / decoder_a stay at their init
    assert not unexpected and all(m.startswith("decoder_") for m in missing), (missing[:3], unexpected[:3])
    full = weights.synth_state_dict(spec, SEED)
    B, T = 3, 24
    lens = [24, 17, 5]
    v = torch.from_numpy(prng.normal(SEED, "golden.legacy.vs", (B, T, 824)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    xs, idxs, margins = [], [], []
    for i in range(B):
        quant, _, info = spk.encode(v[i][mask[i]].unsqueeze(0))
        xs.append(F.pad(quant, (0, T * 8 - quant.shape[-1]), value=0))
        idxs.append(F.pad(info[2].view(-1), (0, T * 8 - info[2].numel()), value=-1))
        _, _, d = ref_cpu.speaker_vq_encode_quant(full, v[i][mask[i]].unsqueeze(0))
        margins.append(ref_cpu.vq_margins(d).min().item())
    x = torch.cat(xs, 0)
    x = x.view(B, -1, 8, 128).contiguous().view(B, -1, 1024).contiguous()
    o_x = ref_cpu.legacy_speaker_features(full, v, mask)
    assert min(margins) >= 1e-4, margins
    err = (x - o_x).abs().max().item()
    assert err < 1e-6, err
    np.savez_compressed(os.path.join(HERE, "legacy_speaker_features.npz"), v_speaker=v.numpy(),
                        lens=np.array(lens, np.int32), x_speaker=x.numpy().astype(np.float32),
                        idx=torch.stack(idxs).numpy().astype(np.int16))
    print("legacy_speaker_features: oracle vs reference err", err, "min margin", min(margins))


if __name__ == "__main__" and "--legacy" in sys.argv:
    legacy_fixtures()


def mymetrics_fixture():
    """print_metrics / print_metrics_full of the reference (code/mymetrics.py:7-130) on seeded synthetic per-clip
    lists; the printed numbers are captured from stdout.  (The reference loader module cannot be imported here:
    it needs librosa, which this image lacks -- pad_collate is tested against its documented behaviour instead.)"""
    import contextlib
    import io
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import mymetrics as ref_mm
    finally:
        os.chdir(cwd)
    lens = [40, 33, 57, 21, 64, 48]
    gts = [prng.normal(SEED, "golden.mm.gt%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    prs = [g + 0.3 * prng.normal(SEED, "golden.mm.pr%d" % i, g.shape) for i, g in enumerate(gts)]
    xs = [prng.normal(SEED, "golden.mm.x%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ret = ref_mm.print_metrics(gts, prs, xs)
        ref_mm.print_metrics_full(gts, prs, xs)
    out = {}
    for line in buf.getvalue().strip().splitlines():
        k, v = line.split(":")
        out[k.strip()] = [float(t) for t in v.split()]
    out["return"] = [float(ret[0]), float(ret[1])]
    with open(os.path.join(HERE, "mymetrics_small.json"), "w") as f:
        json.dump({"lens": lens, "expected": out}, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__" and "--mymetrics" in sys.argv:
    mymetrics_fixture()


def metrics_256_fixture():
    """``metrics_256.npz`` -- the reference's print_metrics / print_metrics_full (code/mymetrics.py:7-130) on a
    full evaluation batch of 256 seeded per-clip lists (ragged lengths 20..299, the size BASELINE C3 evaluates), every scalar they
    print + print_metrics' return value.  The inputs are regenerated from the integer PRNG by the test (the fixture stores the
    lengths and the numbers only)."""
    import contextlib
    import io
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import mymetrics as ref_mm
    finally:
        os.chdir(cwd)
    lens = [int(v) for v in prng.integers(SEED, "golden.m256.lens", (256,), 20, 300)]
    gts = [prng.normal(SEED, "golden.m256.gt%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    prs = [0.6 * g + 0.5 * prng.normal(SEED, "golden.m256.pr%d" % i, g.shape) for i, g in enumerate(gts)]
    xs = [prng.normal(SEED, "golden.m256.x%d" % i, (n, 56)).astype(np.float64) for i, n in enumerate(lens)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ret = ref_mm.print_metrics(gts, prs, xs)
        ref_mm.print_metrics_full(gts, prs, xs)
    labels, values = [], []
    for line in buf.getvalue().strip().splitlines():
        k, v = line.split(":")
        labels.append(k.strip())
        values.append([float(t) for t in v.split()])
    width = max(len(v) for v in values)
    arr = np.full((len(values), width), np.nan)
    for i, v in enumerate(values):
        arr[i, :len(v)] = v
    np.savez_compressed(os.path.join(HERE, "metrics_256.npz"), lens=np.asarray(lens, np.int32), labels=np.asarray(labels),
                        values=arr, ret=np.asarray([float(ret[0]), float(ret[1])]))
    print("metrics_256:", dict(zip(labels, values)), "return", ret)


if __name__ == "__main__" and "--metrics-256" in sys.argv:
    metrics_256_fixture()


def postprocess_fixture():
    """smooth_logits_matrix of reference code/postprocess2emoca.py:7-31.  The script runs its export loop at import
    time, so only that function definition is taken out of the parsed module and executed on a seeded input."""
    import ast
    src = open(os.path.join(REF, "postprocess2emoca.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "smooth_logits_matrix"]
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "postprocess2emoca.py", "exec"), ns)
    x = prng.normal(SEED, "golden.post.x", (37, 56)).astype(np.float64)
    y = ns["smooth_logits_matrix"](x.copy())
    np.savez_compressed(os.path.join(HERE, "postprocess_smooth.npz"), y=y)
    print("postprocess_smooth: zero head rows", int((np.abs(y[:5]).sum(1) == 0).sum()), "zero tail rows",
          int((np.abs(y[-4:]).sum(1) == 0).sum()))


if __name__ == "__main__" and "--postprocess" in sys.argv:
    postprocess_fixture()


def host_protocol_fixture():
    """the reference's OWN evaluation loops and collate function, lifted out of their modules
    by AST (code/x_engine_pt.py imports torcheval and code/dataset/data_loader.py reads files at import; neither
    import works here), executed around tests/stub_model.StubSLMFT on seeded batches."""
    import ast
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, REF)
    import stub_model
    from metrics.eval_utils import calculate_activation_statistics, calculate_frechet_distance

    def lift(path, names, ns):
        tree = ast.parse(open(os.path.join(REF, path)).read())
        fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert len(fns) == len(names), (path, names)
        exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
        return ns

    ns = lift("x_engine_pt.py", ("evaluate_test_epoch", "evaluate_finetune_epoch"),
              {"torch": torch, "np": np, "tqdm": (lambda it, *a, **k: it),
               "calculate_activation_statistics": calculate_activation_statistics,
               "calculate_frechet_distance": calculate_frechet_distance})
    dev = torch.device("cpu")
    out = {}
    yt, yp, xs, ids = ns["evaluate_test_epoch"](stub_model.StubSLMFT(), stub_model.protocol_batches(), dev)
    out["test_ids"] = np.array(ids)
    for name, lst in (("test_true", yt), ("test_pred", yp), ("test_x", xs)):
        out[name + "_lens"] = np.array([a.shape[0] for a in lst])
        out[name + ("" if name == "test_pred" else "_sum")] = (np.concatenate(lst, 0) if name == "test_pred" else
                                                                np.array([float(a.astype(np.float64).sum()) for a in lst]))
    yt, yp, xs, ids = ns["evaluate_finetune_epoch"](stub_model.StubSLMFT(), stub_model.protocol_batches(), dev)
    for name, lst in (("ft_true", yt), ("ft_pred", yp), ("ft_x", xs)):
        out[name + "_lens"] = np.array([a.shape[0] for a in lst])
        out[name + ("" if name == "ft_pred" else "_sum")] = (np.concatenate(lst, 0) if name == "ft_pred" else
                                                              np.array([float(a.astype(np.float64).sum()) for a in lst]))
    cns = lift(os.path.join("dataset", "data_loader.py"), ("pad_collate",), {"torch": torch})
    xx, yy, lens, (sp, li), names = cns["pad_collate"](stub_model.collate_items())
    out.update(coll_x=xx.numpy(), coll_y=yy.numpy(), coll_lens=np.array(lens), coll_speaker=sp.numpy(),
               coll_listener=li.numpy(), coll_names=np.array(names))
    np.savez_compressed(os.path.join(HERE, "host_protocol.npz"), **out)
    print("host_protocol: %d + %d clips, best-of-10 preds %s, collate %s" % (
        len(out["test_pred_lens"]), len(out["ft_pred_lens"]), out["test_pred"].shape, out["coll_x"].shape))


if __name__ == "__main__" and "--host-protocol" in sys.argv:
    host_protocol_fixture()


def train_protocol_fixture():
    """The reference's OWN training / validation loops (code/x_engine_pt.py:9-60, 134-165; code/x_engine.py:8-62, 90-105),
    lifted out of their modules by AST like the evaluation loops above, run around the differentiable stubs of
    tests/stub_model.py with plain SGD: the parameters they leave behind and the values they return are the fixture."""
    import ast
    import io
    import contextlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stub_model

    def lift(path, names, ns):
        tree = ast.parse(open(os.path.join(REF, path)).read())
        fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert len(fns) == len(names), (path, names)
        exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
        return ns

    base = {"torch": torch, "np": np, "nn": torch.nn, "tqdm": (lambda it, *a, **k: it)}
    pt = lift("x_engine_pt.py", ("train_epoch", "evaluate_epoch"), dict(base))
    lg = lift("x_engine.py", ("train_epoch", "train_continuous_epoch", "evaluate_continuous_epoch"), dict(base))
    dev = torch.device("cpu")
    out = {}
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink), torch.enable_grad():
        m = stub_model.StubTrainPT()
        opt = torch.optim.SGD(m.parameters(), lr=0.05)
        sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.9)
        for ep in range(2):
            m.train()
            pt["train_epoch"](m, stub_model.protocol_batches(), opt, dev, scheduler=sched, clip=0.5, print_freq=1, epoch=ep)
        out["pt_w_v"], out["pt_w_a"] = m.w_v.detach().numpy().copy(), m.w_a.detach().numpy().copy()
        out["pt_val"] = np.float64(pt["evaluate_epoch"](m, stub_model.protocol_batches_with_ids(), dev))
        out["pt_lr"] = np.float64(opt.param_groups[0]["lr"])
        m = stub_model.StubTrainLegacy()
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
        for ep in range(2):
            lg["train_epoch"](m, stub_model.legacy_batches(), opt, dev, clip=0.3, print_freq=2, epoch=ep)
        out["lg_w"], out["lg_emb"] = m.w.detach().numpy().copy(), m.emb.detach().numpy().copy()
        m = stub_model.StubTrainContinuous()
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
        for ep in range(2):
            lg["train_continuous_epoch"](m, stub_model.legacy_batches(), opt, dev, clip=0.0, print_freq=2, epoch=ep)
        out["ct_w"] = m.w.detach().numpy().copy()
        out["ct_val"] = np.float64(lg["evaluate_continuous_epoch"](m, [b[:4] for b in stub_model.legacy_batches()], dev))
    out["printed"] = np.array(sink.getvalue().splitlines())
    np.savez_compressed(os.path.join(HERE, "train_protocol.npz"), **out)
    print("train_protocol: pt val %.6f, continuous val %.6f, %d printed lines" % (out["pt_val"], out["ct_val"], len(out["printed"])))


if __name__ == "__main__" and "--train-protocol" in sys.argv:
    train_protocol_fixture()
