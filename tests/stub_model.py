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
