"""CPU: the legacy generator's training loop (reference code/x_engine.py:8-62, SURVEY 8 row f1's training half).
  * dimx.train.legacy_loss (generator + listener VQ-VAE decoder + id embeddings, differentiable) against torch autograd
    over the CPU oracle's ListenerGenerator.forward with listener_ids, as train_epoch calls it: loss and every trainable
    tensor's gradient <= 1e-3 relative; with speaker_ids too; without ids it is the evaluation forward;
  * the two loops' protocol (batch unpacking, mask from src_len, zero_grad / backward / clip / step / scheduler, mean loss)
    on stub models.
The frozen halves (speaker features, listener codes) are injected from the oracle here; on a GPU box they come from the HIP
engine (tests/test_gpu_legacy.py)."""
import pytest
import torch


@pytest.fixture(scope="module")
def legacy_sd():
    import dimx  # noqa: F401
    from dimx import weights
    return weights.synth_state_dict(weights.listener_generator_spec(), 20260928)


def _case(B=2, T=20, lens=(20, 13), seed=9):
    from dimx import prng
    v_s = torch.from_numpy(prng.normal(seed, "lt.vs", (B, T, 824)))
    v_l = torch.from_numpy(prng.normal(seed, "lt.vl", (B, T, 56)))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    return v_s, v_l, mask


@pytest.mark.parametrize("ids", ["listener", "both", "none"])
def test_legacy_loss_gradients_match_autograd_over_the_oracle(legacy_sd, ids):
    from dimx import train as T
    from dimx import weights as W
    from oracle import ref_cpu
    v_s, v_l, mask = _case()
    lid = torch.tensor([3, 41]) if ids in ("listener", "both") else None
    sid = torch.tensor([7, 0]) if ids == "both" else None
    trainable = lambda k: k.startswith(T.LEGACY_TRAINABLE_PREFIXES)
    with torch.enable_grad():
        sd = {k: v.detach().clone().requires_grad_(trainable(k)) for k, v in legacy_sd.items()}
        o_loss, o_pred, aux = ref_cpu.listener_generator_forward(sd, v_s, v_l, mask, speaker_ids=sid, listener_ids=lid)
        o_loss.backward()
        P = {k: v.detach().clone().requires_grad_(trainable(k)) for k, v in legacy_sd.items()}
        x_speaker = ref_cpu.legacy_speaker_features(legacy_sd, v_s, mask)
        loss, pred, logits = T.legacy_loss(P, W.LegacyDims(), W.VQDims(), x_speaker, aux["z_l"], v_l, mask,
                                           P["listener_vq.decoder.decoder_pos_embedding.pe"], speaker_ids=sid, listener_ids=lid)
        loss.backward()
    assert abs(loss.item() - o_loss.item()) <= 1e-5 * max(1.0, abs(o_loss.item()))
    assert torch.equal(logits.argmax(-1), aux["logits"].argmax(-1))
    assert (pred - o_pred).abs().max().item() < 1e-4
    checked = 0
    for k, p in P.items():
        if not trainable(k):
            continue
        g_o = sd[k].grad
        if g_o is None:                  # speaker_embeddings / fc_speaker without speaker_ids, ... : no gradient either side
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        scale = g_o.abs().max().item()
        assert (p.grad - g_o).abs().max().item() <= 1e-3 * max(scale, 1e-8), k
        checked += 1
    assert checked > 150
    # the listener VQ-VAE's decoder trains through the continuous loss, the generator through the cross entropy
    assert P["listener_vq.decoder.vertice_map_reverse.weight"].grad.abs().max() > 0
    assert P["generator.decoder.net.to_logits.weight"].grad.abs().max() > 0
    if lid is not None:
        assert P["fc_listener.weight"].grad.abs().max() > 0 and P["listener_embeddings.weight"].grad[3].abs().max() > 0
        assert float(P["listener_embeddings.weight"].grad[5].abs().max()) == 0.0


class _StubLegacy(torch.nn.Module):
    """Anything with the legacy model's call signature: loss = mean over valid frames of (w . src - tgt)^2."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(6, 4))
        self.calls = []

    def forward(self, src, tgt, mask, speaker_ids=None, listener_ids=None):
        self.calls.append((mask.clone(), speaker_ids, listener_ids))
        err = (src @ self.w.t() - tgt)[mask]
        return (err ** 2).mean().reshape(1), None


class _StubContinuous(_StubLegacy):
    def forward(self, src, tgt, mask):
        return super().forward(src, tgt, mask)[0]


def _loader(n=6, B=3, T=5):
    g = torch.Generator().manual_seed(4)
    w = torch.randn(6, 4, generator=g)
    out = []
    for i in range(n):
        src = torch.randn(B, T, 4, generator=g)
        lens = [T, T - 1, 2]
        out.append((src, src @ w.t(), lens, (torch.arange(B), torch.arange(B) + 10), ["id%d" % i] * B))
    return out


def test_train_epoch_protocol_and_optimisation(capsys):
    from dimx import x_engine
    m = _StubLegacy()
    opt = torch.optim.SGD(m.parameters(), lr=0.2)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.99)
    first = x_engine.train_epoch(m, _loader(), opt, torch.device("cpu"), scheduler=sched, clip=5.0, print_freq=2, epoch=3)
    for _ in range(6):
        last = x_engine.train_epoch(m, _loader(), opt, torch.device("cpu"), clip=5.0, print_freq=100)
    assert last < 0.2 * first
    mask, sid, lid = m.calls[0]
    assert sid is None and torch.equal(lid, torch.arange(3) + 10)          # reference :24: speaker_ids=None
    assert mask.tolist() == [[True] * 5, [True] * 4 + [False], [True, True, False, False, False]]
    out = capsys.readouterr().out
    assert "Epoch: [3][0/6]" in out and "Epoch: [3][4/6]" in out
    assert abs(opt.param_groups[0]["lr"] - 0.2 * 0.99 ** 6) < 1e-12
    assert m.training


def test_train_continuous_epoch_protocol():
    from dimx import x_engine
    m = _StubContinuous()
    opt = torch.optim.SGD(m.parameters(), lr=0.2)
    a = x_engine.train_continuous_epoch(m, _loader(), opt, torch.device("cpu"), clip=0.0)
    b = x_engine.train_continuous_epoch(m, _loader(), opt, torch.device("cpu"), clip=0.0)
    assert b < a


def test_evaluate_continuous_epoch_protocol():
    from dimx import x_engine
    m = _StubContinuous()
    batches = [b[:4] for b in _loader(3)]                         # reference :94: (src, tgt, src_len, _)
    with torch.no_grad():
        want = sum(float(m(b[0], b[1], x_engine._mask_from_lens(b[0], b[2], "cpu"))) for b in batches) / 3
    m.train()
    got = x_engine.evaluate_continuous_epoch(m, batches, torch.device("cpu"), verbose=False)
    assert abs(got - want) < 1e-9 and not m.training
