"""utils/plotting_fns.py mirror -- the NUMBERS of the reference's evaluation figures, without the figures (matplotlib / LaTeX
rendering is out of scope).  Every function keeps the reference's name and arguments; where the reference draws and
saves ``savename`` (a .png), these compute the quantities the figure shows, return them, and -- when ``savename`` is given -- store
them as ``<savename without extension>.npz`` so that the evaluation entry points (test_predictor.py, compare_predictors.py) leave
their results on disk where the reference leaves its plots."""
from __future__ import annotations

import os

import numpy as np


def _save(savename, **arrays):
    if savename is not None:
        os.makedirs(os.path.dirname(os.path.abspath(savename)), exist_ok=True)
        np.savez(os.path.splitext(savename)[0] + '.npz', **arrays)


def photoz_prediction_metrics(z_pred, z_true, threshold=0.15):
    """utils/plotting_fns.py:394-402 -> (resid, bias, mad, frac_out): normalised residual (z_pred - z_true) / (1 + z_true), its
    mean, 1.4826 x the median absolute deviation about its median, the fraction of |resid| > threshold."""
    resid = (z_pred - z_true) / (1 + z_true)
    bias = np.mean(resid)
    mae_bias = np.median(resid)
    mad = 1.4826 * np.median(np.abs(resid - mae_bias))
    frac_out = np.sum(np.abs(resid) > threshold) / z_pred.size
    return resid, bias, mad, frac_out


def _binned(z_pred, z_true, by, lo, hi, n_bins, threshold):
    bins = np.linspace(lo, hi, n_bins + 1)
    mids = 0.5 * (bins[:-1] + bins[1:])
    out = np.full((3, n_bins), np.nan)
    counts = np.zeros(n_bins, dtype=np.int64)
    for i in range(n_bins):
        b = np.where((bins[i] <= by) & (by < bins[i + 1]))[0]
        counts[i] = len(b)
        if len(b):
            _, out[0, i], out[1, i], out[2, i] = photoz_prediction_metrics(z_pred[b], z_true[b], threshold=threshold)
    return mids, out, counts


def evaluate_z(z_pred, z_true, n_bins=8, z_range=(0.2, 2), y_lims=None, threshold=0.15, snr=None, savename=None):
    """utils/plotting_fns.py:525-564: the metrics of the whole set (threshold 0.15, as the reference hard-codes for the full-set
    panel) and per redshift bin (and per S/N bin over (5, 25) when ``snr`` is given: snr_plots, :566-606).  Empty bins are NaN (the
    reference divides by zero there).  -> dict."""
    z_pred, z_true = np.asarray(z_pred).reshape(-1), np.asarray(z_true).reshape(-1)
    _, bias, mad, frac_out = photoz_prediction_metrics(z_pred, z_true, threshold=0.15)
    mids, per_bin, counts = _binned(z_pred, z_true, z_true, z_range[0], z_range[1], n_bins, threshold)
    res = {'bias': bias, 'mad': mad, 'frac_out': frac_out, 'z_bin_mids': mids, 'z_bin_bias': per_bin[0], 'z_bin_mad': per_bin[1],
           'z_bin_frac_out': per_bin[2], 'z_bin_counts': counts}
    if snr is not None:
        smids, sper, scounts = _binned(z_pred, z_true, np.asarray(snr).reshape(-1), 5, 25, n_bins, threshold)
        res.update({'snr_bin_mids': smids, 'snr_bin_bias': sper[0], 'snr_bin_mad': sper[1], 'snr_bin_frac_out': sper[2],
                    'snr_bin_counts': scounts})
    _save(savename, **res)
    return res


def plot_resid_hexbin(label_keys, tgt_stellar_labels, pred_stellar_labels, y_lims=(2,), gridsize=(100, 50), max_counts=30, cmap=None,
                      n_std=3, savename=None):
    """utils/plotting_fns.py:339-391: per label the residual pred - target and the mean / standard deviation the panel annotates."""
    tgt, pred = np.asarray(tgt_stellar_labels), np.asarray(pred_stellar_labels)
    diff = pred[:, :len(label_keys)] - tgt[:, :len(label_keys)]
    res = {'resid_mean': diff.mean(0), 'resid_std': diff.std(0)}
    _save(savename, resid=diff, **res)
    return res


def confusion_matrix(tgt_class, pred_class, n_classes=None):
    """Counts [true class, predicted class] (sklearn.metrics.confusion_matrix with labels 0..n-1)."""
    t, p = np.asarray(tgt_class).astype(np.int64).reshape(-1), np.asarray(pred_class).astype(np.int64).reshape(-1)
    n = int(max(t.max(initial=0), p.max(initial=0)) + 1) if n_classes is None else n_classes
    cm = np.zeros((n, n), dtype=np.int64)
    np.add.at(cm, (t, p), 1)
    return cm


def plot_conf_mat(tgt_class, pred_class, labels, savename):
    """utils/plotting_fns.py:326-337: the confusion matrix of the classifier (+ its accuracy)."""
    cm = confusion_matrix(tgt_class, pred_class, n_classes=len(labels))
    acc = float(np.trace(cm)) / max(1, int(cm.sum()))
    _save(savename, confusion_matrix=cm, accuracy=acc, labels=np.array(labels))
    return cm


def plot_progress(losses, y_lims=None, x_lim=None, lp=False, fontsize=18, savename=None):
    """utils/plotting_fns.py:24-87: the curves of the training-progress figure = the checkpoint's ``losses`` lists."""
    res = {k: np.asarray(v, dtype=np.float64) for k, v in dict(losses).items() if len(v)}
    _save(savename, **res)
    return res
