"""Host-side metrics of the evaluation step (numpy/scipy, float64), restating
reference code/metrics/eval_utils.py:6-91.  They consume the per-clip arrays the engine returns after the
(all-gathered) generation; they are not on the GPU path (excluded from clips/s)."""
import numpy as np
from scipy import linalg


def calculate_activation_statistics(activations):
    """eval_utils.py:6-10: mean over frames and unbiased covariance (np.cov, rowvar=False)."""
    return np.mean(activations, axis=0), np.cov(activations, rowvar=False)


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """eval_utils.py:12-46."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def clip_fd(gt, pred):
    m1, s1 = calculate_activation_statistics(gt)
    m2, s2 = calculate_activation_statistics(pred)
    return calculate_frechet_distance(m1, s1, m2, s2)


def frechet_distances_torch(y_true, y_pred, lens):
    """The same Frechet distance for a whole batch at once, in torch float64 on whatever device the tensors live on (the GPU in
    ``evaluate_test_epoch(fd_backend="device")``): y_true [B, L, F], y_pred [B, S, L, F], lens[j] = valid frames of clip j ->
    fd [B, S].  Mean and unbiased covariance over the valid frames as eval_utils.py:6-10; the trace of the matrix square root
    as sum sqrt(eig(A^T S2 A)) with S1 = A A^T (A = V sqrt(L) from eigh) -- the eigenvalues of S1 S2 without a non-symmetric
    Schur form.  Mathematically the reference's quantity; NOT its arithmetic: scipy's sqrtm differs in the last digits and has
    failure modes on rank-deficient covariances (clips shorter than F + 1 frames: complex results, the eps retry, the
    "Imaginary component" ValueError) that this form does not have.  The host path above stays the default."""
    import torch
    B, S, L, F = y_pred.shape
    dev = y_pred.device
    n = torch.as_tensor(lens, dtype=torch.float64, device=dev)
    m = (torch.arange(L, device=dev)[None, :] < n[:, None]).to(torch.float64)           # [B, L]

    def stats(x, mm, nn):                      # x [..., L, F] float64, mm [..., L], nn [...]
        mu = (x * mm[..., None]).sum(-2) / nn[..., None]
        c = (x - mu[..., None, :]) * mm[..., None]
        return mu, c.transpose(-1, -2) @ c / (nn[..., None, None] - 1.0)

    mu1, s1 = stats(y_true.to(torch.float64), m, n)                                      # [B, F], [B, F, F]
    mu2, s2 = stats(y_pred.to(torch.float64), m[:, None].expand(B, S, L), n[:, None].expand(B, S))
    lam, V = torch.linalg.eigh(s1)
    A = V * lam.clamp(min=0).sqrt()[..., None, :]                                        # S1 = A A^T
    M = A.transpose(-1, -2)[:, None] @ s2 @ A[:, None]                                   # [B, S, F, F] symmetric PSD
    M = 0.5 * (M + M.transpose(-1, -2))
    tr_sqrt = torch.linalg.eigvalsh(M).clamp(min=0).sqrt().sum(-1)
    diff = mu1[:, None] - mu2
    tr = lambda t: torch.diagonal(t, dim1=-2, dim2=-1).sum(-1)
    return (diff * diff).sum(-1) + tr(s1)[:, None] + tr(s2) - 2.0 * tr_sqrt


def calculate_variance(activations):
    """eval_utils.py:48-49."""
    return np.sum(np.var(activations, axis=0))


def sts(x, y, timestep=0.1):
    """eval_utils.py:85-91, vectorised (same value as the reference's double loop)."""
    dx = np.diff(np.asarray(x, dtype=np.float64), axis=0)
    dy = np.diff(np.asarray(y, dtype=np.float64), axis=0)
    return np.sqrt(np.sum((dx - dy) ** 2) / timestep)


def summarize(y_trues, y_preds):
    """FD / MSE / variance / STS averaged over clips on pose[0:6] and exp[6:56]
    (the quantities reference code/mymetrics.py:7-88 prints that need no clustering)."""
    out = {}
    for name, sl in (("pose", slice(0, 6)), ("exp", slice(6, 56))):
        fds, mses, vars_, stss = [], [], [], []
        for gt, pr in zip(y_trues, y_preds):
            g, p = gt[:, sl], pr[:, sl]
            fds.append(clip_fd(g, p))
            mses.append(np.mean((g - p) ** 2))
            vars_.append(calculate_variance(p))
            stss.append(sts(g, p))
        out[name] = {"fd": float(np.mean(fds)), "mse": float(np.mean(mses)), "var": float(np.mean(vars_)),
                     "sts": float(np.mean(stss))}
    return out
