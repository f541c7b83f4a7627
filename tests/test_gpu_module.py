"""GPU: the drop-in nn.Module surface (SLMFT / VQAutoEncoder / evaluation engine) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _clips(B, T, lens, seed=5):
    from dimx import prng
    v_s = torch.from_numpy(prng.normal(seed, "m.vs", (B, T, 56)))
    v_l = torch.from_numpy(prng.normal(seed, "m.vl", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(seed, "m.va", (B, T, 768)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    return v_s, v_l, v_a, mask


@pytest.fixture(scope="module")
def model():
    from dimx.seq2seq_pretrain import SLMFT
    return SLMFT().eval()


def test_slmft_forward_train_and_val_match_oracle(model, full_sd):
    from dimx import prng
    from oracle import ref_cpu
    B, T, lens = 3, 48, [48, 40, 9]
    v_s, v_l, v_a, mask = _clips(B, T, lens)
    dev = torch.device("cuda:0")
    kv = ref_cpu.ar_kv_mask(B, T, 0.15, torch.Generator().manual_seed(2))
    tot, d, pred = model(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mode="train", kv_mask=kv.to(dev))
    rt, rd, rpred = ref_cpu.slmft_forward(full_sd, v_s, v_l, v_a, mask, "train", kv_mask=kv)
    assert pred.shape == (B, T - 1, 56)
    for b, n in enumerate(lens):
        assert (pred[b, :n - 1].cpu() - rpred[b, :n - 1]).abs().max() < 1e-4
    assert abs(float(tot) - float(rt)) < 1e-3 and abs(float(d["l_ce_l"]) - float(rd["l_ce_l"])) < 1e-3
    noise = torch.from_numpy(prng.exponential(8, "m.noise", (T - 1, B, 512)))
    tot, d, pred, tok = model(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mode="val", noise=noise.to(dev),
                              return_tokens=True)
    rt, rd, rpred, aux = ref_cpu.slmft_forward(full_sd, v_s, v_l, v_a, mask, "val", noise=noise, return_aux=True)
    assert torch.equal(tok.cpu(), aux["tokens"])
    assert (pred.cpu() - rpred).abs().max() < 1e-4
    assert d["l_ce_l"] == 0.0 and abs(float(tot) - float(rt)) < 1e-3
    # default call draws fresh randomness (like the reference): two calls differ, greedy calls agree
    a = model(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mode="val", return_tokens=True)[3]
    b = model(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mode="val", return_tokens=True)[3]
    g1 = model(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mode="val", greedy=True, return_tokens=True)[3]
    g2 = model(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mode="val", greedy=True, return_tokens=True)[3]
    assert not torch.equal(a, b) and torch.equal(g1, g2)


def test_non_prefix_mask_is_compacted_like_the_reference(model, full_sd):
    from oracle import ref_cpu
    v_s, v_l, v_a, _ = _clips(2, 20, [20, 20])
    mask = torch.ones(2, 20, dtype=torch.bool)
    mask[1, 3] = False
    mask[1, 11:14] = False
    _, zl = model.forward_vq(v_s.cuda(), v_l.cuda(), mask.cuda(), with_speaker=False)
    _, ref = ref_cpu.forward_vq(full_sd, v_s, v_l, mask, with_speaker=False)
    assert torch.equal(zl.cpu(), ref)


def test_vq_autoencoder_module_roundtrip(golden_dir):
    import os
    from dimx import config
    from dimx.models import get_model
    g = np.load(os.path.join(golden_dir, "vq_roundtrip_C1.npz"))
    m = get_model(config.load_cfg_from_cfg_file(config.DEFAULT_CONFIG)).eval()
    x = torch.from_numpy(g["x"]).cuda()
    dec, loss, info = m(x)
    assert np.array_equal(info[2].view(-1).cpu().numpy(), g["idx"].astype(np.int64).reshape(-1))
    assert np.abs(dec.cpu().numpy() - g["xhat"]).max() < 1e-4
    assert info[1].shape == (300, 512) and float(info[1].sum()) == 300.0
    quant, idx = m.get_quant(x)
    assert quant.shape == (1, 128, 300)
    assert np.abs(m.decode_to_img(idx, (1, 300, 128)).cpu().numpy() - g["xhat"]).max() < 1e-4


def test_evaluation_engine_protocol(model):
    from dimx import x_engine_pt
    torch.manual_seed(1234)     # the sampling seeds are drawn from torch's generator
    B, T = 3, 32
    lens = [32, 20, 11]
    v_s, v_l, v_a, mask = _clips(B, T, lens)
    src = torch.cat([v_s, v_a], -1) * mask[..., None]
    loader = [(src, v_l * mask[..., None], lens, None, ["a", "b", "c"])]
    yt, yp, xs, ids = x_engine_pt.evaluate_finetune_epoch(model, loader, torch.device("cuda:0"))
    assert ids == ["a", "b", "c"] and [a.shape for a in yp] == [(n - 1, 56) for n in lens]
    assert all(a.shape == b.shape for a, b in zip(yt, yp)) and xs[1].shape == (19, 56)
    yt2, yp2, xs2, ids2 = x_engine_pt.evaluate_test_epoch(model, loader, torch.device("cuda:0"), beam_size=3)
    yt3, yp3, _, _ = x_engine_pt.evaluate_test_epoch(model, loader, torch.device("cuda:0"), beam_size=4)   # batched samples
    assert [a.shape for a in yp3] == [(n - 1, 56) for n in lens] and np.isfinite(yp3[1]).all()
    assert [a.shape for a in yp2] == [(n - 1, 56) for n in lens] and np.isfinite(yp2[0]).all()
    tok, pred = x_engine_pt.generate_sharded(model, v_s.cuda(), v_l.cuda(), v_a.cuda(), mask.cuda(), greedy=True)
    assert tok.shape == (B, T - 1) and pred.shape == (B, T - 1, 56)


def test_decode_attention_kernel(model):
    from dimx import engine
    g = torch.Generator().manual_seed(0)
    B, H, T = 5, 12, 77
    q = torch.randn(B, H * 64, generator=g)
    k = torch.randn(B, H, 80, 64, generator=g)
    v = torch.randn(B, H, 80, 64, generator=g)
    km = (torch.rand(B, T, generator=g) > 0.3).to(torch.uint8)
    km[:, 0] = 1
    s = torch.einsum("bhd,bhjd->bhj", q.view(B, H, 64).double(), k[:, :, :T].double()) * 0.125
    s = s.masked_fill(km[:, None, :] == 0, -1e300)
    ref = torch.einsum("bhj,bhjd->bhd", s.softmax(-1), v[:, :, :T].double()).reshape(B, H * 64)
    for ns in (0, 1, 2, 4):   # waves per (clip, head): the split-KV combine must not change the result
        out = engine.op_decode_attn(q.cuda(), k.cuda(), v.cuda(), T, 0.125, km.cuda(), nsplit=ns)
        assert (out.cpu().double() - ref).abs().max() < 2e-5, ns
    outb = engine.op_decode_attn(q.cuda().bfloat16(), k.cuda().bfloat16(), v.cuda().bfloat16(), T, 0.125, km.cuda())
    assert (outb.float().cpu().double() - ref).abs().max() < 5e-2


def test_rccl_single_rank_allgather_path():
    """The N > 1 collective path on the real backend: a 1-rank "nccl" (= RCCL) process group exercises
    init_from_env / all_gather_rows / max_over_ranks / barrier exactly as bench.py --gpus N does per rank."""
    import os
    import socket
    import torch.distributed as dist
    from dimx import dist as dd
    if dist.is_initialized():
        pytest.skip("a process group already exists")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        t = torch.arange(12, dtype=torch.int32, device="cuda:0").view(4, 3)
        out = torch.empty_like(t)
        dist.all_gather_into_tensor(out, t)
        assert torch.equal(out, t)
        x = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        assert float(x) == 3.5
        dist.barrier()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_multi_sample_generation_matches_independent_runs(model, full_sd):
    """n_samples=S in one pass == S separate single-sample passes with the corresponding noise slices (what the
    reference's best-of-N loop does), token for token, and the decoded coefficients use the clip's PE row."""
    from dimx import prng
    B, T, S = 3, 40, 4
    lens = [40, 31, 8]
    v_s, v_l, v_a, mask = _clips(B, T, lens, seed=12)
    dev = torch.device("cuda:0")
    noise = torch.from_numpy(prng.exponential(3, "ms.noise", (T - 1, B * S, 512)))
    args = (v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev))
    tot, d, pred, tok = model(*args, mode="val", noise=noise.to(dev), n_samples=S, return_tokens=True)
    assert pred.shape == (B, S, T - 1, 56) and tok.shape == (B, S, T - 1)
    for s_i in range(S):
        nz = noise.view(T - 1, B, S, 512)[:, :, s_i].contiguous()
        _, _, p1, t1 = model(*args, mode="val", noise=nz.to(dev), return_tokens=True)
        assert torch.equal(tok[:, s_i], t1), "sample %d differs from the independent run" % s_i
        assert (pred[:, s_i] - p1).abs().max() < 1e-5
    # distinct samples really differ
    assert not torch.equal(tok[:, 0], tok[:, 1])


def test_train_epoch_on_gpu_updates_the_engine_weights():
    """SURVEY 8 row f3 on the device: x_engine_pt.train_epoch (AdamW, clip 1.0) with the listener codes and the decoded
    motion from the HIP engine and the transformer's backward on PyTorch-ROCm autograd; the next HIP inference call
    must see the updated parameters (the engine re-packs changed weights)."""
    from dimx import train as T
    from dimx import x_engine_pt
    from dimx.seq2seq_pretrain import SLMFT
    dev = torch.device("cuda:0")
    B, Tn, lens = 4, 32, [32, 32, 20, 11]
    v_s, v_l, v_a, mask = _clips(B, Tn, lens, seed=33)
    src = torch.cat([v_s, v_a], -1) * mask[..., None]
    loader = [(src, v_l * mask[..., None], lens, None, ["a", "b", "c", "d"])] * 3
    with torch.enable_grad():
        m = SLMFT().to(dev)
        m.eval()
        args = (v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev))
        _, _, before = m(*args, mode="val", greedy=True)
        opt = T.make_optimizer(m, lr=1e-4)
        l0 = x_engine_pt.train_epoch(m, loader[:1], opt, dev, clip=1.0, log=lambda *_: None)
        l1 = x_engine_pt.train_epoch(m, loader, opt, dev, clip=1.0, log=lambda *_: None)
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0
    m.eval()
    with torch.no_grad():
        _, _, after = m(*args, mode="val", greedy=True)
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    # the frozen VQ-VAEs did not move
    ref = SLMFT()
    for k, v in ref.state_dict().items():
        if k.startswith("listener_vq.") and v.dtype.is_floating_point:
            assert torch.equal(v, m.state_dict()[k].cpu()), k


def test_train_epoch_on_the_hip_training_step_matches_the_autograd_restatement():
    """x_engine_pt.train_epoch driven by dimx.train_hip.HipTrainer (forward, backward, clip, AdamW all in libdimx_hip.so)
    against the same epoch on the PyTorch-autograd restatement (dimx.train, the checker): same loss trajectory, same trained
    parameters to optimiser rounding, and the module / inference engine see the trained weights after the epoch."""
    from dimx import train as T
    from dimx import x_engine_pt
    from dimx.seq2seq_pretrain import SLMFT
    from dimx.train_hip import HipTrainer
    dev = torch.device("cuda:0")
    B, Tn, lens = 4, 32, [32, 32, 20, 11]
    v_s, v_l, v_a, mask = _clips(B, Tn, lens, seed=33)
    src = torch.cat([v_s, v_a], -1) * mask[..., None]
    loader = [(src, v_l * mask[..., None], lens, None, ["a", "b", "c", "d"])] * 3
    m_hip, m_ref = SLMFT().to(dev), SLMFT().to(dev)
    m_hip.mask_prob = m_ref.mask_prob = 0.0           # no random key mask: the two paths must see the same problem
    tr = HipTrainer(m_hip, lr=1e-4, clip=1.0)
    l_hip = x_engine_pt.train_epoch(m_hip, loader, tr, dev, log=lambda *_: None)
    with torch.enable_grad():
        opt = T.make_optimizer(m_ref, lr=1e-4)
        l_ref = x_engine_pt.train_epoch(m_ref, loader, opt, dev, clip=1.0, log=lambda *_: None, backward="autograd")
    assert getattr(m_ref, "_dimx_hip_trainer", None) is None                    # the checker really ran on autograd
    assert np.isfinite(l_hip) and abs(l_hip - l_ref) < 5e-3 * max(1.0, abs(l_ref)), (l_hip, l_ref)   # CE + continuous loss
    sd_h, sd_r = m_hip.state_dict(), m_ref.state_dict()
    moved, worst = 0.0, 0.0
    fresh = SLMFT().state_dict()
    for name, _, _ in tr.layout:
        d0 = (sd_r[name].cpu() - fresh[name]).abs().max().item()
        moved = max(moved, d0)
        worst = max(worst, (sd_h[name].cpu() - sd_r[name].cpu()).abs().max().item())
    # three Adam steps of lr 1e-4 move every trained element by ~3e-4; the two paths agree to a small fraction of that except
    # where a gradient element is ~0 and Adam's g / (|g| + eps) turns rounding noise into a sign (a few elements)
    assert moved > 1e-4 and worst <= 2.5 * moved, (moved, worst)
    close = torch.cat([(sd_h[n].cpu() - sd_r[n].cpu()).abs().reshape(-1) for n, _, _ in tr.layout])
    assert (close < 2e-5).float().mean().item() > 0.99
    for k, v in fresh.items():                          # the frozen VQ-VAEs did not move
        if k.startswith(("listener_vq.", "speaker_vq.")) and v.dtype.is_floating_point:
            assert torch.equal(v, sd_h[k].cpu()), k
    m_hip.eval()
    with torch.no_grad():
        _, _, after = m_hip(v_s.to(dev), v_l.to(dev), v_a.to(dev), mask.to(dev), mode="val", greedy=True)
    assert torch.isfinite(after).all()


def test_train_epoch_with_a_torch_adamw_runs_on_the_hip_step():
    """the reference's own call shape -- train_epoch(model, loader, torch.optim.AdamW(model.parameters(),
    lr), device, clip=1.0) (code/finetune_s2s_pretrain.py:118-132) -- lands on the hand-written HIP training step: a HipTrainer
    stands in for the AdamW (hyper-parameters from its param_groups at every step, so a torch scheduler drives it; moments and
    step count exported to optimizer.state), and the result equals the PyTorch-autograd epoch to optimiser rounding."""
    from dimx import x_engine_pt
    from dimx.seq2seq_pretrain import SLMFT
    from dimx.train_hip import HipTrainer
    dev = torch.device("cuda:0")
    B, Tn, lens = 4, 32, [32, 32, 20, 11]
    v_s, v_l, v_a, mask = _clips(B, Tn, lens, seed=33)
    src = torch.cat([v_s, v_a], -1) * mask[..., None]
    loader = [(src, v_l * mask[..., None], lens, None, ["a", "b", "c", "d"])] * 3
    m_hip, m_ref = SLMFT().to(dev), SLMFT().to(dev)
    m_hip.mask_prob = m_ref.mask_prob = 0.0
    opt = torch.optim.AdamW(m_hip.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)          # lr 1e-4, 5e-5, 2.5e-5 over the three batches
    logs = []
    l_hip = x_engine_pt.train_epoch(m_hip, loader, opt, dev, scheduler=sched, clip=1.0, log=logs.append)
    owner, tr = m_hip._dimx_hip_trainer
    assert owner is opt and isinstance(tr, HipTrainer) and tr.step_count == 3 and abs(tr.lr - 2.5e-5) < 1e-12 and tr.clip == 1.0
    assert not any("autograd" in ln for ln in logs)
    named = dict(m_hip.named_parameters())
    st = opt.state[named["decoder_joint.net.to_logits.weight"]]
    assert int(st["step"]) == 3 and float(st["exp_avg"].abs().max()) > 0 and float(st["exp_avg_sq"].abs().max()) > 0
    with torch.enable_grad():
        opt_r = torch.optim.AdamW(m_ref.parameters(), lr=1e-4)
        sched_r = torch.optim.lr_scheduler.StepLR(opt_r, step_size=1, gamma=0.5)
        l_ref = x_engine_pt.train_epoch(m_ref, loader, opt_r, dev, scheduler=sched_r, clip=1.0, log=lambda *_: None, backward="autograd")
    assert abs(l_hip - l_ref) < 5e-3 * max(1.0, abs(l_ref)), (l_hip, l_ref)
    sd_h, sd_r = m_hip.state_dict(), m_ref.state_dict()
    close = torch.cat([(sd_h[n].cpu() - sd_r[n].cpu()).abs().reshape(-1) for n, _, _ in tr.layout])
    assert (close < 2e-5).float().mean().item() > 0.99
    # a second epoch continues with the same trainer (moments kept); another optimiser cannot be stepped by the HIP kernels: the
    # default route RAISES (no silent second backend), the autograd restatement runs only when asked for
    x_engine_pt.train_epoch(m_hip, loader[:1], opt, dev, clip=1.0, log=lambda *_: None)
    assert m_hip._dimx_hip_trainer[1] is tr and tr.step_count == 4
    sgd = torch.optim.SGD(m_ref.parameters(), lr=1e-5)
    from dimx import lib as _lib
    with pytest.raises(_lib.DimxError, match="not torch.optim.AdamW"):
        x_engine_pt.train_epoch(m_ref, loader[:1], sgd, dev, clip=1.0, log=lambda *_: None)
    with torch.enable_grad():
        assert np.isfinite(x_engine_pt.train_epoch(m_ref, loader[:1], sgd, dev, clip=1.0, log=lambda *_: None, backward="autograd"))


def test_reference_sub_apis_with_the_reference_call_shapes(model, full_sd):
    """SURVEY 8b sub-APIs: forward_encoder alone (dimx_encode_speaker), forward_decoder(x_s, z_l, x_a, mask, mode)
    positionally with the x_s that forward_encoder returned (dimx_set_context), and VQAutoEncoder.decode(quant) on
    latents that are NOT codebook rows (dimx_vq_decode_latent)."""
    from dimx import prng
    from oracle import ref_cpu
    B, T, lens = 3, 40, [40, 31, 9]
    v_s, v_l, v_a, mask = _clips(B, T, lens, seed=44)
    dev = torch.device("cuda:0")
    x_s = model.forward_encoder(v_s.to(dev), mask.to(dev))
    ref_xs = ref_cpu.slmft_forward_encoder(full_sd, v_s, mask)
    for b, n in enumerate(lens):
        assert (x_s[b, :n].cpu() - ref_xs[b, :n]).abs().max() < 1e-4
    _, z_l = model.forward_vq(v_s.to(dev), v_l.to(dev), mask.to(dev), with_speaker=False)
    kv = ref_cpu.ar_kv_mask(B, T, 0.15, torch.Generator().manual_seed(9))
    loss, logits = model.forward_decoder(x_s, z_l, v_a.to(dev), mask.to(dev), "train", kv_mask=kv.to(dev))
    ctx = ref_cpu.slmft_context(full_sd, ref_xs, v_a)
    r_loss, r_logits = ref_cpu.ar_forward(full_sd, z_l.cpu(), ctx, mask, kv)
    valid = mask[:, 1:]
    assert (logits.cpu() - r_logits)[valid].abs().max() < 1e-3 and abs(float(loss) - float(r_loss)) < 1e-3
    noise = torch.from_numpy(prng.exponential(44, "sub.noise", (T - 1, B, 512)))
    _, tok = model.forward_decoder(x_s, z_l, v_a.to(dev), mask.to(dev), "val", noise=noise.to(dev))
    r_tok = ref_cpu.ar_generate(full_sd, z_l[:, 0].cpu(), T - 1, ctx, mask, noise)
    assert torch.equal(tok.cpu(), r_tok)
    # decode(quant) on arbitrary latents: the oracle decoder run on the same latents (a codebook made of them)
    lat = torch.from_numpy(prng.normal(45, "sub.lat", (2, 128, 26)))          # [B,128,L] like encode returns
    out = model.listener_vq.decode(lat.to(dev))
    sd2 = dict(full_sd)
    sd2["listener_vq.quantize.embedding.weight"] = lat.permute(0, 2, 1).reshape(-1, 128)
    ref = ref_cpu.vq_decode(sd2, torch.arange(2 * 26).view(2, 26), prefix="listener_vq.")
    assert out.shape == (2, 26, 56) and (out.cpu() - ref).abs().max() < 1e-4


def test_hip_trainer_adopts_parameters_written_between_epochs():
    """the flat arenas are the master copy between sync_to_model() calls; a checkpoint loaded into the module in between must not be
    overwritten at the end of the next epoch: train_epoch notices the change (tensor version counters) and reloads the arena."""
    from dimx import lib, prng, x_engine_pt
    from dimx.seq2seq_pretrain import SLMFT
    dev = torch.device("cuda:0")
    m = SLMFT(numeric_mode=lib.MODE_PARITY_F32).to(dev)
    B, T = 2, 24
    v_s = torch.from_numpy(prng.normal(3, "adopt.vs", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(3, "adopt.va", (B, T, 768)))
    v_l = torch.from_numpy(prng.normal(3, "adopt.vl", (B, T, 56)))
    batch = (torch.cat([v_s, v_a], dim=-1), v_l, [T, 17], None, None)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    logs = []
    with torch.enable_grad():
        x_engine_pt.train_epoch(m, [batch], opt, dev, clip=1.0, log=logs.append)
        tr = m._dimx_hip_trainer[1]
        assert not tr.refresh_from_model_if_changed()
        name = "decoder_joint.net.to_logits.weight"
        with torch.no_grad():
            dict(m.named_parameters())[name].fill_(0.125)            # stands for load_state_dict of a checkpoint
        x_engine_pt.train_epoch(m, [batch], opt, dev, clip=1.0, log=logs.append)
    assert any("arena reloaded" in ln for ln in logs)
    w = dict(m.named_parameters())[name]
    assert (w - 0.125).abs().max().item() < 1e-3 and not torch.equal(w, torch.full_like(w, 0.125))   # one AdamW step away from the new value
