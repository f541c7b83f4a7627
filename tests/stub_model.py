"""Deterministic CPU stand-in for SLMFT used to pin the HOST side of the evaluation protocol:
tests/golden/make_golden.py runs the reference's own ``evaluate_test_epoch`` / ``evaluate_finetune_epoch`` loops
(code/x_engine_pt.py:201-277, taken out of the module by AST because the module imports torcheval) around this stub
and stores what they return; tests/test_host_protocol_golden.py runs dimx.x_engine_pt around the same stub.

The k-th stochastic forward of the model returns sample k: the reference draws its beam_size samples with beam_size
calls, dimx with one n_samples=beam_size call -- both must see the same sample sequence."""
import torch


class StubSLMFT(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.calls = 0

    @staticmethod
    def _sample(k, v_s, tgt):
        B, T, C = tgt.shape
        t = torch.arange(T - 1, dtype=torch.float64)[None, :, None]
        c = torch.arange(C, dtype=torch.float64)[None, None, :]
        b = torch.arange(B, dtype=torch.float64)[:, None, None]
        noise = torch.sin(t * (0.37 + 0.011 * k) + c * 1.3 + b * 0.7 + 2.1 * k) * (0.35 + 0.05 * (k % 4))
        return (0.6 * tgt[:, 1:].double() + 0.1 * v_s[:, 1:].double() + noise).float()

    def forward(self, v_speaker, v_listener, v_audio, mask, mode="train", n_samples=1, **kw):
        S = int(n_samples)
        if mode == "train":                      # evaluate_finetune_epoch: deterministic teacher-forced output
            return torch.zeros(()), {}, self._sample(-1, v_speaker, v_listener)
        if S > 1:
            pred = torch.stack([self._sample(self.calls + s, v_speaker, v_listener) for s in range(S)], 1)
        else:
            pred = self._sample(self.calls, v_speaker, v_listener)
        self.calls += S
        return torch.zeros(()), {}, pred


def protocol_batches():
    """Two loader batches in the reference's format: (src [B,T,824] zero-padded, tgt [B,T,56], src_len, _, data_ids)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dimx  # noqa: F401
    from dimx import prng
    out = []
    for i, lens in enumerate(([96, 80, 71, 64], [90, 66, 75])):
        B, T = len(lens), max(lens)
        src = torch.from_numpy(prng.normal(77 + i, "proto.src", (B, T, 824)))
        tgt = torch.from_numpy(prng.normal(77 + i, "proto.tgt", (B, T, 56)))
        for j, n in enumerate(lens):
            src[j, n:] = 0
            tgt[j, n:] = 0
        out.append((src, tgt, list(lens), None, ["clip_%d_%d" % (i, j) for j in range(B)]))
    return out


def collate_items():
    """Ragged dataset items in the reference's format (x [L,Cx], y [L,Cy], name, speaker_id, listener_id, sentiment);
    narrow feature widths keep the fixture small (the collate function is width-agnostic)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dimx  # noqa: F401
    from dimx import prng
    items = []
    for i, n in enumerate((17, 5, 23, 11)):
        items.append((torch.from_numpy(prng.normal(5, "coll.x.%d" % i, (n, 12))),
                      torch.from_numpy(prng.normal(5, "coll.y.%d" % i, (n, 7))), "n%d" % i, 3 + i, 40 - i, i % 3))
    return items


# ---------------------------------------------------------------------------------------------------------------------
# differentiable stand-ins for the TRAINING loops (tests/golden/make_golden.py --train-protocol runs the reference's own
# train_epoch / evaluate_epoch / train_continuous_epoch / evaluate_continuous_epoch around them; tests/
# test_host_protocol_golden.py runs dimx's).  Deterministic initial weights, no random number anywhere.
# ---------------------------------------------------------------------------------------------------------------------
def _det(*shape):
    n = 1
    for k in shape:
        n *= k
    return (torch.sin(torch.arange(n, dtype=torch.float64) * 0.37 + 0.2) * 0.1).float().reshape(shape)


class StubTrainPT(torch.nn.Module):
    """x_engine_pt protocol: model(src_s_v, tgt, src_s_a, mask, mode='train') -> (loss [1], dict of six terms, None)."""

    def __init__(self):
        super().__init__()
        self.w_v = torch.nn.Parameter(_det(56, 56))
        self.w_a = torch.nn.Parameter(_det(56, 16))
        self.seen = []

    def forward(self, v_speaker, v_listener, v_audio, mask, mode="train", **kw):
        self.seen.append((mode, mask.clone()))
        pred = v_speaker @ self.w_v.t() + v_audio[..., :16] @ self.w_a.t()
        err = (pred - v_listener)[mask]
        loss = (err ** 2).mean().reshape(1)
        z = torch.zeros(())          # the reference's evaluate_epoch takes .mean() of every term (code/x_engine_pt.py:161)
        d = {"l_ce_s": z, "l_ce_l": loss.detach() * 0.25, "l_cont_s": z, "l_cont_l": err.abs().mean().detach(),
             "nce": z + 0.5, "c_acc": z}
        return loss, d, None


class StubTrainLegacy(torch.nn.Module):
    """x_engine protocol: model(src, tgt, mask, speaker_ids=None, listener_ids=None) -> (loss [1], pred)."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(_det(6, 4))
        self.emb = torch.nn.Parameter(_det(32, 6))
        self.calls = []

    def forward(self, src, tgt, mask, speaker_ids=None, listener_ids=None):
        self.calls.append((mask.clone(), speaker_ids, listener_ids))
        pred = src @ self.w.t()
        if listener_ids is not None:
            pred = pred + self.emb[listener_ids][:, None, :]
        err = (pred - tgt)[mask]
        return (err ** 2).mean().reshape(1), pred


class StubTrainContinuous(StubTrainLegacy):
    """the ContinuousTransformer protocol: model(src, tgt, mask) -> loss."""

    def forward(self, src, tgt, mask):
        return super().forward(src, tgt, mask)[0]


def protocol_batches_with_ids():
    """protocol_batches() with (speaker_ids, listener_ids) in the fourth slot (code/x_engine_pt.py:149 unpacks them)."""
    return [(b[0], b[1], b[2], (torch.arange(len(b[2])), torch.arange(len(b[2])) + 3), b[4]) for b in protocol_batches()]


def legacy_batches(n=5, B=3, T=7):
    """(src, tgt, src_len, (speaker_ids, listener_ids), data_ids) -- what code/x_engine.py:15 unpacks."""
    out = []
    for i in range(n):
        src = _det(B, T, 4) * (3.0 + i) + 0.05 * i
        tgt = torch.cos(torch.arange(B * T * 6, dtype=torch.float64) * 0.11 + i).float().reshape(B, T, 6)
        lens = [T, T - 2, 3]
        out.append((src, tgt, lens, (torch.arange(B) + i, torch.arange(B) * 2 + i), ["c%d_%d" % (i, j) for j in range(B)]))
    return out
