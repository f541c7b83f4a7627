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
