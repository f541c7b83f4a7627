"""GPU: the legacy ListenerGenerator's training step on the HIP kernels (dimx.train_hip.LegacyHipTrainer -> csrc/train.hip:
legacy_run, vqdec_fwd / vqdec_bwd) against PyTorch autograd over ``dimx.train.legacy_loss`` (itself checked against autograd over
the CPU oracle in tests/test_gpu_legacy.py) on the same inputs -- reference loop code/x_engine.py:8-36, model
code/seq2seq.py:235-278.  f32 parity mode: loss, decoded motion and every trained tensor's gradient <= 1e-3 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, T, lens, seed=5):
    from dimx import prng
    v_s = torch.from_numpy(prng.normal(seed, "legacy.vs", (B, T, 824)))
    v_l = torch.from_numpy(prng.normal(seed, "legacy.vl", (B, T, 56)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    return v_s, v_l, mask


def _model(mode):
    from dimx import seq2seq
    from dimx import train as T
    m = seq2seq.ListenerGenerator(numeric_mode=mode).cuda()
    T.set_legacy_trainable(m)
    m.train()
    return m


@pytest.mark.parametrize("with_ids", [True, False])
def test_legacy_hip_gradients_match_autograd(with_ids):
    from dimx import lib
    from dimx import train as T
    from dimx.train_hip import LegacyHipTrainer
    dev = torch.device("cuda:0")
    model = _model(lib.MODE_PARITY_F32)
    v_s, v_l, mask = _case(2, 20, [20, 13], seed=21)
    lid = torch.tensor([3, 41]).to(dev) if with_ids else None
    args = (v_s.to(dev), v_l.to(dev), mask.to(dev))
    with torch.enable_grad():
        a_loss, a_pred, a_logits = model(*args, speaker_ids=None, listener_ids=lid, return_logits=True)
        a_loss.backward()
    tr = LegacyHipTrainer(model)
    loss, parts, pred, logits = tr.forward_backward(*args, listener_ids=lid, return_logits=True)
    if with_ids:
        logits = logits[:, 1:]
    assert (logits - a_logits.detach()).abs().max().item() < 1e-3
    assert abs(loss.item() - a_loss.item()) < 1e-4 * max(1.0, abs(a_loss.item())), (loss.item(), a_loss.item(), parts)
    assert (pred - a_pred.detach()).abs().max().item() < 1e-3
    named = dict(model.named_parameters())
    trained = {n for n, _ in T.legacy_trainable_parameters(model)}
    in_layout = {n for n, _, _ in tr.layout}
    worst = 0.0
    for name in sorted(trained):
        g_a = named[name].grad
        if name not in in_layout:     # speaker_ids is None in the loop (speaker_embeddings / fc_speaker), project_out is never called
            assert g_a is None or float(g_a.abs().max()) == 0.0, name
            continue
        g_h = tr.grad(name)
        if g_a is None or float(g_a.abs().max()) == 0.0:                  # the id-conditioning tensors without ids
            assert float(g_h.abs().max()) == 0.0, name
            continue
        rel = (g_h - g_a).abs().max().item() / max(g_a.abs().max().item(), 1e-8)
        worst = max(worst, rel)
        assert rel < 1e-3, (name, rel)
    print("legacy HIP training step (ids %s): worst relative gradient error vs autograd %.2e over %d tensors" % (with_ids, worst, len(in_layout)))


def test_legacy_hip_training_reduces_the_loss_and_bf16_agrees():
    from dimx import lib
    from dimx.train_hip import LegacyHipTrainer
    dev = torch.device("cuda:0")
    v_s, v_l, mask = _case(3, 24, [24, 17, 9], seed=33)
    lid = torch.tensor([5, 5, 77]).to(dev)          # a repeated id: the embedding gradient accumulates in clip order
    args = (v_s.to(dev), v_l.to(dev), mask.to(dev))
    mf = _model(lib.MODE_PARITY_F32)
    tf = LegacyHipTrainer(mf, lr=2e-4)
    l0, _ = tf.train_step(*args, listener_ids=lid)
    for _ in range(4):
        l1, _ = tf.train_step(*args, listener_ids=lid)
    assert l1.item() < l0.item() - 0.05, (l0.item(), l1.item())
    before = {k: v.detach().clone() for k, v in mf.state_dict().items()}
    tf.sync_to_model()
    after = mf.state_dict()
    for k in before:
        changed = not torch.equal(before[k], after[k])
        if k.startswith(("speaker_vq.", "listener_vq.encoder.", "listener_vq.quantize.", "speaker_embeddings.", "fc_speaker.")):
            assert not changed, k
        elif k.startswith(("generator.decoder.net.attn_layers", "listener_vq.decoder.decoder_transformer", "fc_listener.")):
            assert changed, k
    # bf16 operands: same loss to bf16 accuracy, gradients within 5 % of the f32 ones; a second call gives the same bits
    ga = LegacyHipTrainer(_model(lib.MODE_PARITY_F32))
    gb = LegacyHipTrainer(_model(lib.MODE_PERF_BF16))
    la = ga.forward_backward(*args, listener_ids=lid)[0].item()
    lb = gb.forward_backward(*args, listener_ids=lid)[0].item()
    # (the generator's tensors: the VQ decoder's gradient depends on the arg-max codes, and a bf16 logit row may pick another code)
    gen = torch.cat([torch.arange(off, off + numel) for name, off, numel in ga.layout if name.startswith("generator.")]).to(dev)
    rel = ((gb.grads[gen] - ga.grads[gen]).norm() / ga.grads[gen].norm()).item()
    rel_all = ((gb.grads - ga.grads).norm() / ga.grads.norm()).item()
    print("legacy bf16 step: loss %.5f vs f32 %.5f, relative gradient difference %.3f (generator), %.3f (all)" % (lb, la, rel, rel_all))
    assert abs(lb - la) < 5e-2 and rel < 0.08
    g1 = gb.grads.clone()
    gb.forward_backward(*args, listener_ids=lid)
    assert torch.equal(g1, gb.grads)


def test_x_engine_train_epoch_with_a_torch_adamw_runs_on_the_hip_step():
    """the reference's own call (code/x_engine.py:8-36 from code/train_vq_decoder... scripts: train_epoch(model, loader, AdamW, device,
    clip)) lands on LegacyHipTrainer; against the same epochs on the autograd restatement: same losses, same trained weights."""
    from dimx import lib, x_engine
    from dimx import train as T
    from dimx.train_hip import LegacyHipTrainer
    dev = torch.device("cuda:0")
    v_s, v_l, mask = _case(2, 20, [20, 13], seed=21)
    batch = (v_s, v_l, [20, 13], (torch.tensor([0, 1]), torch.tensor([3, 41])), ["a", "b"])
    runs = {}
    for how in ("auto", "autograd"):
        m = _model(lib.MODE_PARITY_F32)
        opt = torch.optim.AdamW([p for _, p in T.legacy_trainable_parameters(m)], lr=2e-4)
        losses = [x_engine.train_epoch(m, [batch], opt, dev, clip=1.0, backward=how) for _ in range(3)]
        runs[how] = (losses, {k: v.detach().clone() for k, v in m.state_dict().items()}, m, opt)
    assert isinstance(runs["auto"][2]._dimx_hip_trainer[1], LegacyHipTrainer)
    assert not hasattr(runs["autograd"][2], "_dimx_hip_trainer")
    for a, b in zip(runs["auto"][0], runs["autograd"][0]):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (runs["auto"][0], runs["autograd"][0])
    worst = 0.0
    for k, v in runs["autograd"][1].items():
        if v.dtype.is_floating_point:
            worst = max(worst, (runs["auto"][1][k] - v).abs().max().item())
    print("legacy train_epoch, HIP step vs autograd after 3 AdamW steps: max |weight difference| %.2e" % worst)
    assert worst < 5e-5       # three steps of lr 2e-4: an AdamW step moves a weight by ~lr whatever the gradient's size
    st = runs["auto"][3].state[dict(runs["auto"][2].named_parameters())["generator.decoder.net.to_logits.weight"]]
    assert float(st["step"]) == 3.0
