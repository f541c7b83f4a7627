"""Evaluation engine of the legacy generator with the batch protocol of reference ``code/x_engine.py:65-88``.

``evaluate_epoch(model, loader, device)``: for every batch ``(src [B,T,824], tgt [B,T,56], src_len, ids)``
build the prefix mask, run the teacher-forced forward and ``generate``, and accumulate the token perplexity
of the ground-truth listener codes ``z_gt[:, 1:]`` under the teacher-forced logits.

Reference quirk (documented, not reproduced): ``x_engine.py:78`` feeds the SECOND return value of
``model(src, tgt, mask)`` -- the decoded continuous motion [B,T-1,56] -- into ``Perplexity`` as if it were
the [B,T-1,512] logits; with 512 code classes that indexes out of range.  The logits the metric needs are the
teacher-forced decoder logits, which ``ListenerGenerator.forward`` keeps as ``last_logits``.

``model`` may be the bare module or a wrapper exposing ``.module`` (the reference calls
``model.module.generate`` on its DataParallel wrapper).  ``train_epoch`` / ``train_continuous_epoch`` are the
reference's loops (:8-62).  ``train_epoch`` over a ListenerGenerator on a GPU with a torch.optim.AdamW lands on the
hand-written HIP training step (dimx.train_hip.LegacyHipTrainer: forward, backward, clip and AdamW in libdimx_hip.so);
``backward="autograd"`` keeps the PyTorch-autograd restatement (dimx.train.legacy_loss), its checker.
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


def evaluate_continuous_epoch(model, loader, device, verbose=True):
    """reference code/x_engine.py:90-105: mean of ``model(src, tgt, mask)`` (a model that returns its loss alone)."""
    model.eval()
    losses = []
    with torch.no_grad():
        for batch in loader:
            src, tgt, src_len = batch[0].to(device), batch[1].to(device), batch[2]
            losses.append(float(model(src, tgt, _mask_from_lens(src, src_len, device)).mean().item()))
    if verbose:
        print("Validation: Loss {loss:.4f}\t".format(loss=np.mean(losses)))
    return float(np.mean(losses)) if losses else float("nan")


def _train_loop(model, loader, optimizer, device, scheduler, clip, print_freq, epoch, step_loss):
    """Loop body shared by the two reference loops: zero_grad, loss.mean().backward(), clip_grad_norm_, step, scheduler,
    running mean printed every ``print_freq`` batches (reference code/x_engine.py:14-36, :40-62).  N > 1: gradients are
    averaged over the ranks in flat buckets before the clip (dimx.train.all_reduce_grads); the reference's DataParallel
    wrapper is replaced by one process per GPU."""
    from . import train as T
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    try:
        T.assert_same_batch_count(len(loader), device)
    except TypeError:
        pass
    losses, seen = [], []
    for i, batch in enumerate(loader):
        optimizer.zero_grad()
        with torch.enable_grad():        # a training loop differentiates whatever the caller's ambient grad mode is
            loss = step_loss(batch)
            loss.mean().backward()
        T.all_reduce_grads(params)
        if clip > 0:
            torch.nn.utils.clip_grad_norm_(params, clip)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        losses.append(loss.mean().item())
        seen.append(losses[-1])
        if i % print_freq == 0:
            print("Epoch: [{0}][{1}/{2}]\tLoss {loss_avg:.4f}\t".format(epoch, i, len(loader), loss_avg=np.mean(losses)))
            losses = []
    return float(np.mean(seen)) if seen else float("nan")


def _train_epoch_hip(model, loader, trainer, device, scheduler, clip, print_freq, epoch, optimizer=None):
    """the reference's loop body on the HIP step: the loop only feeds batches; lr / betas / eps / weight_decay follow the torch
    optimizer the trainer stands in for (schedulers work), the trained weights and the moments go back at the end of the epoch."""
    from . import dist as ddist
    from .x_engine_pt import _adopt_hyperparameters, _set_epoch
    model.train()
    changed = trainer.refresh_from_model_if_changed()
    if changed:
        print("train_epoch: the module's parameters changed since the HIP trainer last synchronised: arena reloaded from the module")
    if optimizer is not None and type(optimizer) is torch.optim.AdamW and (changed or trainer.optimizer_state_changed(optimizer)):
        trainer.import_optimizer_state(optimizer)   # optimizer.load_state_dict / an autograd epoch in between
    trainer.clip = float(clip or 0.0)
    try:
        ddist.assert_same_batch_count(len(loader), device)
    except TypeError:
        pass
    _set_epoch(loader, epoch)
    losses, seen = [], []
    for i, batch in enumerate(loader):
        src, tgt, src_len, (_speaker_ids, listener_ids) = batch[0], batch[1], batch[2], batch[3]
        src, tgt, listener_ids = src.to(device), tgt.to(device), listener_ids.to(device)
        if optimizer is not None:
            _adopt_hyperparameters(trainer, optimizer)
            optimizer._opt_called = True
        loss, _ = trainer.train_step(src, tgt, _mask_from_lens(src, src_len, device), listener_ids=listener_ids)
        if scheduler is not None:
            scheduler.step()
        losses.append(loss)                      # device scalars: the host synchronises once per print_freq batches
        if i % print_freq == 0:
            vals = [float(v) for v in torch.stack(losses).cpu()]
            seen += vals
            print("Epoch: [{0}][{1}/{2}]\tLoss {loss_avg:.4f}\t".format(epoch, i, len(loader), loss_avg=np.mean(vals)))
            losses = []
    if losses:
        seen += [float(v) for v in torch.stack(losses).cpu()]
    trainer.sync_to_model()
    if optimizer is not None and type(optimizer) is torch.optim.AdamW:
        trainer.export_optimizer_state(optimizer)
    return float(np.mean(seen)) if seen else float("nan")


def train_epoch(model, loader, optimizer, device, scheduler=None, clip=0.0, print_freq=100, epoch=0, backward="auto"):
    """reference code/x_engine.py:8-36: batches ``(src, tgt, src_len, (speaker_ids, listener_ids), data_ids)``;
    ``model(src, tgt, mask, speaker_ids=None, listener_ids=listener_ids) -> (loss, pred)``.  Returns the epoch's mean
    loss (the reference returns None; the value is extra).  The reference's own call (a torch.optim.AdamW over the module's
    parameters, the module on a GPU) runs on ``LegacyHipTrainer``; a ``LegacyHipTrainer`` may also be passed as ``optimizer``."""
    from .train_hip import LegacyHipTrainer
    if backward not in ("auto", "hip", "autograd"):
        raise ValueError("backward must be 'auto', 'hip' or 'autograd'")
    if isinstance(optimizer, LegacyHipTrainer):
        return _train_epoch_hip(model, loader, optimizer, device, scheduler, clip, print_freq, epoch)
    if backward != "autograd":
        from .x_engine_pt import _hip_trainer_for
        from .x_engine_pt import _no_hip_step
        tr = _hip_trainer_for(model, optimizer, device, print)
        if tr is not None:
            return _train_epoch_hip(model, loader, tr, device, scheduler, clip, print_freq, epoch, optimizer=optimizer)
        if backward == "hip":
            from . import lib as L
            raise L.DimxError("train_epoch(backward='hip'): %s" % _hip_trainer_for.last_reason)
        _no_hip_step(model, "train_epoch", print)   # raises for a dimx model: the autograd route is an explicit opt-in

    def step_loss(batch):
        src, tgt, src_len, (_speaker_ids, listener_ids) = batch[0], batch[1], batch[2], batch[3]
        src, tgt, listener_ids = src.to(device), tgt.to(device), listener_ids.to(device)
        loss, _ = model(src, tgt, _mask_from_lens(src, src_len, device), speaker_ids=None, listener_ids=listener_ids)
        return loss
    return _train_loop(model, loader, optimizer, device, scheduler, clip, print_freq, epoch, step_loss)


def train_continuous_epoch(model, loader, optimizer, device, scheduler=None, clip=0.0, print_freq=100, epoch=0):
    """reference code/x_engine.py:38-62: the same loop for a model whose ``model(src, tgt, mask)`` returns the loss
    alone (the reference's ContinuousTransformer, which is not part of this build; any such module works)."""
    def step_loss(batch):
        src, tgt, src_len = batch[0].to(device), batch[1].to(device), batch[2]
        return model(src, tgt, _mask_from_lens(src, src_len, device))
    return _train_loop(model, loader, optimizer, device, scheduler, clip, print_freq, epoch, step_loss)
