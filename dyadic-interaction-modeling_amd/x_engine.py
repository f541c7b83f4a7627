"""Evaluation engine of the legacy generator with the batch protocol of reference ``code/x_engine.py:65-88``.

``evaluate_epoch(model, loader, device)``: for every batch ``(src [B,T,824], tgt [B,T,56], src_len, ids)``
build the prefix mask, run the teacher-forced forward and ``generate``, and accumulate the token perplexity
of the ground-truth listener codes ``z_gt[:, 1:]`` under the teacher-forced logits.

Reference quirk (documented, not reproduced): ``x_engine.py:78`` feeds the SECOND return value of
``model(src, tgt, mask)`` -- the decoded continuous motion [B,T-1,56] -- into ``Perplexity`` as if it were
the [B,T-1,512] logits; with 512 code classes that indexes out of range.  The logits the metric needs are the
teacher-forced decoder logits, which ``ListenerGenerator.forward`` keeps as ``last_logits``.

``model`` may be the bare module or a wrapper exposing ``.module`` (the reference calls
``model.module.generate`` on its DataParallel wrapper).  Training loops (``train_epoch``) are out of scope.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _mask_from_lens(src, src_len, device):
    mask = torch.zeros((src.shape[0], src.shape[1]), dtype=torch.bool)
    for j in range(src.shape[0]):
        mask[j, :src_len[j]] = True
    return mask.to(device)


class TokenPerplexity:
    """exp(mean NLL) over all target tokens seen so far -- torcheval.metrics.Perplexity semantics
    (sum of per-token negative log-likelihoods / token count, exponentiated at compute())."""

    def __init__(self):
        self.nll = 0.0
        self.count = 0

    def update(self, logits, target):
        lp = F.log_softmax(logits.double().reshape(-1, logits.shape[-1]), dim=-1)
        t = target.reshape(-1).long()
        self.nll += float(-lp.gather(1, t[:, None]).sum())
        self.count += int(t.numel())

    def compute(self):
        return float(np.exp(self.nll / max(self.count, 1)))


def evaluate_epoch(model, loader, device, generate_kw=None, verbose=True):
    """-> perplexity (float).  Also returns nothing else, like the reference; the mean loss is printed."""
    model.eval()
    inner = getattr(model, "module", model)
    losses = []
    metric = TokenPerplexity()
    generate_kw = generate_kw or {}
    with torch.no_grad():
        for batch in loader:
            src, tgt, src_len = batch[0], batch[1], batch[2]
            src = src.to(device)
            tgt = tgt.to(device)
            mask = _mask_from_lens(src, src_len, device)
            loss, _ = model(src, tgt, mask)
            logits = inner.last_logits
            z_pred, z_gt = inner.generate(src, tgt, mask, **generate_kw)
            for j in range(z_gt.shape[0]):
                sel = mask[j, 1:]
                if int(sel.sum()) == 0:
                    continue
                metric.update(logits[j][sel].unsqueeze(0).cpu(), z_gt[j][1:][sel].unsqueeze(0).cpu())
            losses.append(float(loss.mean().item()))
    ppl = metric.compute()
    if verbose:
        print("Validation: Loss {loss:.4f}\tPerplexity {perplexity:.3f}\t".format(loss=np.mean(losses), perplexity=ppl))
    return ppl


def train_epoch(*args, **kwargs):
    """Import-compatibility placeholder for reference code/x_engine.py:8-36 (the LEGACY ListenerGenerator's loop).
    The training step that is built is the DIM-Listener one: dimx.x_engine_pt.train_epoch / dimx.train."""
    raise NotImplementedError("the legacy ListenerGenerator's training loop is not built; the DIM-Listener (SLMFT) "
                              "training step is dimx.x_engine_pt.train_epoch")


train_continuous_epoch = train_epoch
