"""Output side of the path: generated listener coefficients -> the per-frame ``pose.npy`` / ``exp.npy`` tree the
PIRenderer/EMOCA stage reads (reference ``code/postprocess2emoca.py:7-104``).

``smooth_logits_matrix`` is the reference's temporal smoothing: a length-``window_size`` moving average per
coefficient, written to rows ``[w/2, T - w/2]``; the first ``w/2`` and the last ``w/2 - 1`` rows stay ZERO (the
reference allocates ``zeros_like`` and never fills them) -- reproduced.  ``export_predictions`` is the script's
loop as a function: it takes the dict ``examples/test_s2s_pretrain.py`` (reference ``test_s2s_pretrain.py:76-83``)
pickles.
"""
import os

import numpy as np


def smooth_logits_matrix(input_matrix, window_size=10):
    x = np.asarray(input_matrix)
    T, C = x.shape
    w = window_size
    out = np.zeros_like(x)
    if T < w:
        raise ValueError("smooth_logits_matrix needs at least window_size=%d frames (got %d), like the reference" % (w, T))
    kernel = np.ones((w,)) / w
    for j in range(C):
        out[int(w / 2):(T - int(w / 2)) + 1, j] = np.convolve(x[:, j], kernel, mode="valid")
    return out


def export_predictions(data, output_dir_pred="output_data_listener_new", output_dir_gt="output_data_listener_new_gt",
                       smooth=True, window_size=10):
    """data: {'y_pred': [..[T_i,56]], 'y_true': [...], 'data_ids': [...]} -> writes
    <dir>/<clip id>/<frame>/pose.npy (6) and exp.npy (50); returns the number of frames written per tree."""
    n = 0
    for pred, gt, did in zip(data["y_pred"], data["y_true"], data["data_ids"]):
        clip = str(did).split("/")[-1].split(".")[0]
        for arr, root in ((pred, output_dir_pred), (gt, output_dir_gt)):
            a = smooth_logits_matrix(arr, window_size) if smooth else np.asarray(arr)
            for f, coeff in enumerate(a):
                d = os.path.join(root, clip, str(f))
                os.makedirs(d, exist_ok=True)
                np.save(os.path.join(d, "pose.npy"), coeff[:6])
                np.save(os.path.join(d, "exp.npy"), coeff[6:])
        n += len(pred)
    return n
