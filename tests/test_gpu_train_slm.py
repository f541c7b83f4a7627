"""GPU: the SLM pre-training step on the HIP kernels (dimx.train_hip.SlmHipTrainer -> csrc/train.hip: slm_run) against PyTorch
autograd over ``dimx.train.slm_loss`` (itself checked against autograd over the CPU oracle in tests/test_gpu_slm.py) on the same
inputs and the same injected random masks -- reference loop code/train_s2s_pretrain.py:41-64, model
code/seq2seq_pretrain.py:300-323.  f32 parity mode: all five loss terms and every trained tensor's gradient <= 1e-3 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, T, lens, seed=4, ratio=0.3):
    from dimx import prng
    from oracle import ref_cpu
    v_s = torch.from_numpy(prng.normal(seed, "slm.vs", (B, T, 56)))
    v_l = torch.from_numpy(prng.normal(seed, "slm.vl", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(seed, "slm.va", (B, T, 768)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    g = torch.Generator().manual_seed(seed)
    ms = ref_cpu.slm_random_masks(mask, ratio, g)
    ml = ref_cpu.slm_random_masks(mask, ratio, g)
    return v_s, v_l, v_a, mask, ms, ml


def _model(mode):
    from dimx import train as T
    from dimx.seq2seq_pretrain import SLM
    m = SLM(numeric_mode=mode).cuda()
    T.set_slm_trainable(m)
    m.train()
    return m


@pytest.mark.parametrize("B,T,lens", [(3, 40, [40, 33, 12]), (2, 70, [70, 41])])
def test_slm_hip_gradients_match_autograd(B, T, lens):
    from dimx import lib
    from dimx import train as Tr
    from dimx.train_hip import SlmHipTrainer
    dev = torch.device("cuda:0")
    model = _model(lib.MODE_PARITY_F32)
    v_s, v_l, v_a, mask, ms, ml = (t.to(dev) for t in _case(B, T, lens))
    with torch.no_grad():
        z_s, z_l = model.forward_vq(v_s, v_l, mask)
    with torch.enable_grad():
        a_total, a_d, _ = model(v_s, v_l, v_a, mask, mask_speaker=ms, mask_listener=ml, z_s=z_s, z_l=z_l)
        a_total.backward()
    tr = SlmHipTrainer(model)
    total, d = tr.forward_backward(v_s, v_l, v_a, mask, mask_speaker=ms, mask_listener=ml, z_s=z_s, z_l=z_l)
    for k in ("l_ce_s", "l_ce_l", "l_cont_s", "l_cont_l", "nce"):
        assert abs(float(d[k]) - float(a_d[k])) < 1e-4 * max(1.0, abs(float(a_d[k]))), (k, float(d[k]), float(a_d[k]))
    assert abs(float(d["c_acc"]) - float(a_d["c_acc"])) < 1e-6
    assert abs(total.item() - a_total.item()) < 1e-4 * abs(a_total.item())
    named = dict(model.named_parameters())
    trained = {n for n, _ in Tr.slm_trainable_parameters(model)}
    in_layout = {n for n, _, _ in tr.layout}
    assert in_layout <= trained
    worst, worst_name = 0.0, ""
    for name in sorted(trained):
        g_a = named[name].grad
        if name not in in_layout:                     # project_out of the three encoders is never called
            assert g_a is None or float(g_a.abs().max()) == 0.0, name
            continue
        g_h = tr.grad(name)
        assert g_a is not None, name
        rel = (g_h - g_a).abs().max().item() / max(g_a.abs().max().item(), 1e-8)
        if rel > worst:
            worst, worst_name = rel, name
        assert rel < 1e-3, (name, rel)
    print("SLM HIP training step B=%d T=%d: worst relative gradient error vs autograd %.2e (%s) over %d tensors" % (
        B, T, worst, worst_name, len(in_layout)))
    # a second call gives the same bits (no float atomics, fixed summation orders)
    g1 = tr.grads.clone()
    tr.forward_backward(v_s, v_l, v_a, mask, mask_speaker=ms, mask_listener=ml, z_s=z_s, z_l=z_l)
    assert torch.equal(g1, tr.grads)


@pytest.mark.parametrize("B,T,lens", [(1, 24, [24]), (3, 30, [30, 5, 2])])
def test_slm_hip_edge_shapes_match_autograd(B, T, lens):
    """a single clip (the InfoNCE matrix is 1 x 1: nce = 0, no gradient through it) and clips too short to get a masked frame
    (int(len * ratio) = 0: they contribute no targets, only keys): losses and gradients still equal autograd's."""
    from dimx import lib
    from dimx.train_hip import SlmHipTrainer
    dev = torch.device("cuda:0")
    model = _model(lib.MODE_PARITY_F32)
    v_s, v_l, v_a, mask, ms, ml = (t.to(dev) for t in _case(B, T, lens, seed=13))
    assert bool(ms.any()) and bool(ml.any())
    with torch.no_grad():
        z_s, z_l = model.forward_vq(v_s, v_l, mask)
    with torch.enable_grad():
        a_total, a_d, _ = model(v_s, v_l, v_a, mask, mask_speaker=ms, mask_listener=ml, z_s=z_s, z_l=z_l)
        a_total.backward()
    tr = SlmHipTrainer(model)
    total, d = tr.forward_backward(v_s, v_l, v_a, mask, mask_speaker=ms, mask_listener=ml, z_s=z_s, z_l=z_l)
    assert torch.isfinite(total) and abs(total.item() - a_total.item()) < 1e-4 * max(1.0, abs(a_total.item()))
    if B == 1:
        assert abs(float(d["nce"])) < 1e-6 and float(d["c_acc"]) == 1.0
    named = dict(model.named_parameters())
    worst = 0.0
    for name, _, _ in tr.layout:
        g_a = named[name].grad
        g_h = tr.grad(name)
        if g_a is None or float(g_a.abs().max()) == 0.0:
            assert float(g_h.abs().max()) < 1e-7, name
            continue
        rel = (g_h - g_a).abs().max().item() / max(g_a.abs().max().item(), 1e-8)
        worst = max(worst, rel)
        assert rel < 1e-3, (name, rel)
    print("SLM HIP step, edge shape B=%d lens=%s: worst relative gradient error %.2e" % (B, lens, worst))


def test_slm_hip_training_reduces_the_loss_and_bf16_agrees():
    from dimx import lib
    from dimx import train as Tr
    from dimx.train_hip import SlmHipTrainer
    dev = torch.device("cuda:0")
    v_s, v_l, v_a, mask, ms, ml = (t.to(dev) for t in _case(4, 48, [48, 48, 30, 11], seed=9))
    mf = _model(lib.MODE_PARITY_F32)
    tf = SlmHipTrainer(mf, lr=1e-4)
    kw = dict(mask_speaker=ms, mask_listener=ml)
    l0, _ = tf.train_step(v_s, v_l, v_a, mask, **kw)
    for _ in range(4):
        l1, d1 = tf.train_step(v_s, v_l, v_a, mask, **kw)
    assert l1.item() < l0.item() - 0.05, (l0.item(), l1.item())
    before = {k: v.detach().clone() for k, v in mf.state_dict().items()}
    tf.sync_to_model()
    after = mf.state_dict()
    for k in before:
        changed = not torch.equal(before[k], after[k])
        if k.startswith(Tr.SLM_FROZEN_PREFIXES) or k.endswith("project_out.weight"):
            assert not changed, k
        elif k.startswith(("encoder_l.attn_layers", "encoder_joint.attn_layers", "decoder_joint.net.attn_layers",
                           "speaker_vq.decoder.decoder_transformer", "listener_vq.decoder.decoder_transformer", "norm_l.", "patch_embed_l")):
            assert changed, k
    # bf16 operands: same losses to bf16 accuracy, the gradients of everything in front of the arg-max within 8 % in norm
    ga = SlmHipTrainer(_model(lib.MODE_PARITY_F32))
    gb = SlmHipTrainer(_model(lib.MODE_PERF_BF16))
    la, da = ga.forward_backward(v_s, v_l, v_a, mask, **kw)
    lb, db = gb.forward_backward(v_s, v_l, v_a, mask, **kw)
    pre = torch.cat([torch.arange(off, off + numel) for name, off, numel in ga.layout if "_vq." not in name]).to(dev)
    rel = ((gb.grads[pre] - ga.grads[pre]).norm() / ga.grads[pre].norm()).item()
    print("SLM bf16 step: total %.5f vs f32 %.5f (ce %.4f/%.4f vs %.4f/%.4f, nce %.4f vs %.4f), relative gradient difference %.3f" % (
        lb.item(), la.item(), float(db["l_ce_s"]), float(db["l_ce_l"]), float(da["l_ce_s"]), float(da["l_ce_l"]), float(db["nce"]),
        float(da["nce"]), rel))
    assert abs(float(db["l_ce_s"]) - float(da["l_ce_s"])) < 5e-2 and abs(float(db["l_ce_l"]) - float(da["l_ce_l"])) < 5e-2
    assert abs(float(db["nce"]) - float(da["nce"])) < 5e-2 and rel < 0.08


def test_train_epoch_with_a_torch_adamw_runs_the_slm_on_the_hip_step():
    """the reference's own call (code/train_s2s_pretrain.py:41-64): x_engine_pt.train_epoch(model, loader, torch.optim.AdamW(...),
    device, clip=1.0) builds a SlmHipTrainer that stands in for the AdamW; the moments come back in optimizer.state, the trained
    weights in the module.  The random frame masks are drawn inside the step, so the check is the protocol and the optimisation."""
    from dimx import lib, x_engine_pt
    from dimx import train as Tr
    from dimx.train_hip import SlmHipTrainer
    dev = torch.device("cuda:0")
    v_s, v_l, v_a, mask, _, _ = _case(3, 40, [40, 33, 12])
    batch = (torch.cat([v_s, v_a], dim=-1), v_l, [40, 33, 12], None, None)
    m = _model(lib.MODE_PARITY_F32)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    opt = torch.optim.AdamW([p for _, p in Tr.slm_trainable_parameters(m)], lr=1e-4)
    logs = []
    torch.manual_seed(0)
    first = x_engine_pt.train_epoch(m, [batch], opt, dev, clip=1.0, log=logs.append)
    for _ in range(4):
        last = x_engine_pt.train_epoch(m, [batch], opt, dev, clip=1.0, log=logs.append)
    assert not any("autograd" in l for l in logs), logs
    assert isinstance(m._dimx_hip_trainer[1], SlmHipTrainer) and m._dimx_hip_trainer[1].step_count == 5
    assert all(k in logs[0] for k in ("l_ce_s", "l_ce_l", "l_cont_s", "l_cont_l", "nce", "c_acc"))
    assert last < first, (first, last)
    after = m.state_dict()
    for k in before:
        changed = not torch.equal(before[k], after[k])
        if k.startswith(Tr.SLM_FROZEN_PREFIXES):
            assert not changed, k
        elif k.startswith(("encoder_s.attn_layers", "decoder_joint.net.attn_layers", "listener_vq.decoder.decoder_transformer")):
            assert changed, k
    st = opt.state[dict(m.named_parameters())["decoder_joint.net.to_logits.weight"]]
    assert float(st["step"]) == 5.0 and float(st["exp_avg_sq"].abs().max()) > 0.0
