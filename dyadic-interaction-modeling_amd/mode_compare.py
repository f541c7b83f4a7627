"""What the bf16 perf mode costs in the quantities the reference reports.

Per-token agreement is the wrong instrument for free-running generation: after the first flipped token an autoregressive sequence is
a different sequence.  What the reference measures on generated motion is distributional -- per-clip Frechet distance, MSE, variance,
STS on pose[0:6] / exp[6:56] (reference code/metrics/eval_utils.py:12-46, code/mymetrics.py:7-88; ``dimx.metrics.summarize`` here).
So: the SAME clips and the SAME sampler seed through both numeric modes, and three comparisons with those definitions:

  * ``between_modes``  -- FD / MSE / STS of the bf16 mode's motion against the f32 mode's motion, clip by clip;
  * ``between_seeds``  -- the same numbers for two f32 generations that differ only in the sampler seed: the spread the model's own
                          sampling gives (the yardstick: a mode difference below it is inside the sampling noise);
  * ``vs_target``      -- each mode's motion against the batch's listener motion (the reference's ground truth; synthetic here):
                          the numbers a user of ``print_metrics`` would see, side by side.
"""
import numpy as np
import torch

from . import metrics as M


def _clips(pred, lens=None):
    p = pred.detach().float().cpu().numpy().astype(np.float64)
    return [p[i] if lens is None else p[i, :lens[i]] for i in range(p.shape[0])]


def compare_modes(model_bf16, model_f32, v_speaker, v_listener, v_audio, mask, seed=20260928, lens=None):
    """-> dict (JSON-able).  Both models on the same device, eval mode; mask [B, T] bool; lens: valid frames - 1 per clip (None = all)."""
    with torch.no_grad():
        _, _, pb, tb = model_bf16(v_speaker, v_listener, v_audio, mask, mode="val", seed=seed, return_tokens=True)
        _, _, pf, tf = model_f32(v_speaker, v_listener, v_audio, mask, mode="val", seed=seed, return_tokens=True)
        _, _, pf2, _ = model_f32(v_speaker, v_listener, v_audio, mask, mode="val", seed=seed + 1, return_tokens=True)
    B = pb.shape[0]
    tb, tf = tb.reshape(B, -1), tf.reshape(B, -1)
    same = (tb == tf)
    first_flip = torch.where(same.all(1), torch.full((B,), same.shape[1], device=same.device), (~same).float().argmax(1))
    gt = _clips(v_listener[:, 1:], lens)
    cb, cf, cf2 = _clips(pb, lens), _clips(pf, lens), _clips(pf2, lens)
    out = {
        "clips": int(B), "frames": int(mask.shape[1]), "seed": int(seed),
        "free_running_token_agreement": float(same.float().mean().item()),
        "mean_steps_before_first_flip": float(first_flip.float().mean().item()),
        "between_modes": M.summarize(cf, cb),
        "between_seeds": M.summarize(cf, cf2),
        "vs_target": {"bf16": M.summarize(gt, cb), "f32": M.summarize(gt, cf)},
    }
    for part in ("pose", "exp"):
        m, s = out["between_modes"][part]["fd"], out["between_seeds"][part]["fd"]
        out["between_modes"][part]["fd_over_seed_spread"] = float(m / s) if s > 0 else None
        fb, ff = out["vs_target"]["bf16"][part], out["vs_target"]["f32"][part]
        out["vs_target"][part + "_relative_shift"] = {k: float(abs(fb[k] - ff[k]) / max(abs(ff[k]), 1e-12)) for k in ("fd", "mse", "var")}
    return out
