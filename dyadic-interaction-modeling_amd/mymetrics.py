"""``print_metrics`` / ``print_metrics_full`` with the output format and values of reference
``code/mymetrics.py:7-130`` (what ``test_s2s_pretrain.py:68-69`` calls after ``evaluate_test_epoch``).

Inputs are the per-clip lists the evaluation engine returns: ``y_true[i]``, ``y_pred[i]`` ``[len_i-1, 56]`` and the
speaker motion ``x[i]`` ``[len_i-1, >=56]`` (numpy).  Host-side numpy/scipy/sklearn, float64 like the reference;
the per-clip double loop of the reference's ``sts`` is vectorised (same value).  Both functions also RETURN what
they print (the reference returns only ``(fid_pose, fid_exp)`` from ``print_metrics``; that pair stays the return
value, the full dict is available through ``compute_metrics``).
"""
import numpy as np

from .metrics import calculate_activation_statistics, calculate_frechet_distance, sts


def calcuate_sid(gt, pred, type="exp"):
    """reference code/metrics/eval_utils.py:51-83 (name kept, typo included): entropy of the histogram of the
    predictions over a KMeans codebook (k = 40 exp / 20 pose, random_state 0) fitted on the ground truth."""
    from sklearn.cluster import KMeans
    k = 40 if type == "exp" else 20
    sl = slice(6, None) if type == "exp" else slice(0, 6)
    merge_gt = np.concatenate(gt, axis=0)[:, sl]
    km = KMeans(n_clusters=k, random_state=0, n_init="auto").fit(merge_gt)
    lab = km.predict(np.concatenate(pred, axis=0)[:, sl])
    hist = np.bincount(lab, minlength=k).astype(np.float64)
    hist = hist / hist.sum()
    return float(-np.sum(hist * np.log2(hist + 1e-6)))


def _fd(a, b):
    mu1, s1 = calculate_activation_statistics(a)
    mu2, s2 = calculate_activation_statistics(b)
    return calculate_frechet_distance(mu1, s1, mu2, s2)


def compute_metrics(y_true, y_pred, x, with_sid=True):
    gt, pred = y_true, y_pred
    pose, exp = slice(0, 6), slice(6, None)
    out = {}
    out["fid_pose"] = float(np.mean([_fd(g[:, pose], p[:, pose]) for g, p in zip(gt, pred)]))
    out["fid_exp"] = float(np.mean([_fd(g[:, exp], p[:, exp]) for g, p in zip(gt, pred)]))
    out["pfid_pose"] = float(np.mean([_fd(np.concatenate([xi[:, 0:6], g[:, pose]], -1),
                                          np.concatenate([xi[:, 0:6], p[:, pose]], -1)) for g, p, xi in zip(gt, pred, x)]))
    out["pfid_exp"] = float(np.mean([_fd(np.concatenate([xi[:, 6:], g[:, exp]], -1),
                                         np.concatenate([xi[:, 6:], p[:, exp]], -1)) for g, p, xi in zip(gt, pred, x)]))
    out["mse_pose"] = float(np.mean([np.mean((g[:, pose] - p[:, pose]) ** 2) for g, p in zip(gt, pred)]))
    out["mse_exp"] = float(np.mean([np.mean((g[:, exp] - p[:, exp]) ** 2) for g, p in zip(gt, pred)]))
    if with_sid:
        out["sid_pose"] = (calcuate_sid(gt, pred, "pose"), calcuate_sid(gt, gt, "pose"))
        out["sid_exp"] = (calcuate_sid(gt, pred, "exp"), calcuate_sid(gt, gt, "exp"))
    g = np.concatenate(gt, axis=0).reshape(-1, 56)
    p = np.concatenate(pred, axis=0).reshape(-1, 56)
    out["var_pose"] = (float(np.var(g[:, pose].reshape(-1))), float(np.var(p[:, pose].reshape(-1))))
    out["var_exp"] = (float(np.var(g[:, exp].reshape(-1))), float(np.var(p[:, exp].reshape(-1))))
    xs = np.concatenate(x, axis=0)[:, 0:56]

    def pcc(a, b):
        return np.corrcoef(a.reshape(-1), b.reshape(-1))[0, 1]
    out["rpcc_pose"] = float(abs(pcc(g[:, pose], xs[:, pose]) - pcc(p[:, pose], xs[:, pose])))
    out["rpcc_exp"] = float(abs(pcc(g[:, exp], xs[:, exp]) - pcc(p[:, exp], xs[:, exp])))
    out["sts_pose"] = float(sts(g[:, pose], p[:, pose]))
    out["sts_exp"] = float(sts(g[:, exp], p[:, exp]))
    return out


def print_metrics(y_true, y_pred, x):
    m = compute_metrics(y_true, y_pred, x)
    print("fid_pose: ", m["fid_pose"])
    print("fid_exp: ", m["fid_exp"])
    print("pfid_pose: ", m["pfid_pose"])
    print("pfid_exp: ", m["pfid_exp"])
    print("mse_pose: ", m["mse_pose"])
    print("mse_exp: ", m["mse_exp"])
    print("sid_pose: ", *m["sid_pose"])
    print("sid_exp: ", *m["sid_exp"])
    print("var_pose: ", *m["var_pose"])
    print("var_exp: ", *m["var_exp"])
    print("rpcc pose: ", m["rpcc_pose"])
    print("rpcc exp: ", m["rpcc_exp"])
    print("sts pose: ", m["sts_pose"])
    print("sts exp: ", m["sts_exp"])
    return m["fid_pose"], m["fid_exp"]


def compute_metrics_full(y_true, y_pred, x):
    gt, pred = y_true, y_pred
    out = {"fid": float(np.mean([_fd(g, p) for g, p in zip(gt, pred)])),
           "pfid": float(np.mean([_fd(np.concatenate([xi, g], -1), np.concatenate([xi, p], -1))
                                  for g, p, xi in zip(gt, pred, x)])),
           "mse": float(np.mean([np.mean((g - p) ** 2) for g, p in zip(gt, pred)]))}
    g = np.concatenate(gt, axis=0).reshape(-1, 56)
    p = np.concatenate(pred, axis=0).reshape(-1, 56)
    out["var"] = (float(np.var(g.reshape(-1))), float(np.var(p.reshape(-1))))
    return out


def print_metrics_full(y_true, y_pred, x):
    m = compute_metrics_full(y_true, y_pred, x)
    print("fid: ", m["fid"])
    print("pfid: ", m["pfid"])
    print("mse: ", m["mse"])
    print("var: ", *m["var"])
    return m
