"""Figures of an evaluated run: driven paths, error over time, KITTI error curves, per-sequence bars.

The counterparts of `/root/reference/deepclr/evaluation/plot.py:37-223` for `scripts/evaluation.py:113-140`, which
saves every figure the `Evaluator.plot_*` methods return. Same figures and axes, computed from the column arrays of
`MetricsContainer` (numpy binning instead of a pandas group-by). One deliberate difference: the reference's
error-over-time rotation trace converts to degrees twice (`plot.py:126,133`); here it is converted once.

Host-side only; matplotlib is needed for this module alone.
"""
from typing import Any, Dict, Optional, Tuple

import matplotlib
import matplotlib.pyplot as plt
import numpy as np

from .evaluation import MetricsContainer, Sequence

CM = 0.393701                    # inches per centimetre
SIZE_CM = (15.0, 12.0)
DPI = 300
SPEED_BINS = 11                  # plot.py:184: 12 edges between the slowest and the fastest segment


def _figure(rows: int = 1, cols: int = 1, scale: float = 1.0, projection: Optional[str] = None,
            **kwargs: Any) -> Tuple[matplotlib.figure.Figure, Any]:
    fig = plt.figure(figsize=(SIZE_CM[0] * CM * scale, SIZE_CM[1] * CM * scale), dpi=kwargs.pop('dpi', DPI),
                     facecolor='w', edgecolor='w')
    if projection is not None:
        return fig, fig.add_subplot(1, 1, 1, projection=projection)
    axes = fig.subplots(rows, cols, **kwargs)
    return fig, axes


def plot_sequence(sequence: Sequence, title: Optional[str] = None) -> matplotlib.figure.Figure:
    """Ground-truth and predicted path in 3D on a common cubic range."""
    pred, gt = sequence.prediction.get_path(), sequence.ground_truth.get_path()
    fig, ax = _figure(projection='3d')
    lo, hi = min(pred.min(), gt.min()), max(pred.max(), gt.max())
    ax.plot(gt[:, 0], gt[:, 1], gt[:, 2], 'g-', label='Ground Truth')
    ax.plot(pred[:, 0], pred[:, 1], pred[:, 2], 'r-', label='Prediction')
    ax.set_xlabel('x'), ax.set_ylabel('y'), ax.set_zlabel('z')
    ax.set_xlim(lo, hi), ax.set_ylim(lo, hi), ax.set_zlim(lo, hi)
    ax.legend()
    if title:
        fig.suptitle(title)
    return fig


def plot_sequence_2d(sequence: Sequence, title: Optional[str] = None) -> matplotlib.figure.Figure:
    """Top view (x, y) of both paths, square around the ground truth with a 5 m margin."""
    pred, gt = sequence.prediction.get_path(), sequence.ground_truth.get_path()
    fig, ax = _figure()
    span = gt[:, :2].max(axis=0) - gt[:, :2].min(axis=0)
    centre = gt[:, :2].min(axis=0) + span / 2
    half = span.max() / 2 + 5
    ax.plot(gt[:, 0], gt[:, 1], '-', color=(0, 0.447, 0.741), label='Ground Truth')
    ax.plot(pred[:, 0], pred[:, 1], '--', color=(0.85, 0.325, 0.098), label='Prediction')
    ax.set_xlabel('x'), ax.set_ylabel('y')
    ax.set_xlim(centre[0] - half, centre[0] + half), ax.set_ylim(centre[1] - half, centre[1] + half)
    ax.legend()
    if title:
        fig.suptitle(title)
    return fig


def plot_error_over_time(errors: MetricsContainer) -> matplotlib.figure.Figure:
    """Per-pair translation [m] and rotation [deg] error against the pair index."""
    fig, (top, bottom) = _figure(2, 1, scale=2.0)
    top.plot(errors.arrays['translation'])
    top.set_title('Translation Error'), top.set_ylabel('e_t')
    bottom.plot(np.rad2deg(errors.arrays['rotation']))
    bottom.set_title('Rotation Error'), bottom.set_ylabel('e_r [deg]')
    return fig


def kitti_curves(errors: MetricsContainer) -> Dict[str, np.ndarray]:
    """The numbers behind `plot_kitti_errors`: mean translation / rotation error per segment length, and per speed
    bin (11 equal bins between the slowest and fastest segment, right-closed, so the slowest segment itself falls
    outside the first bin as in `pandas.cut`; empty bins are NaN)."""
    a = errors.arrays
    lengths = np.unique(a['segment_length'])
    by_len = np.array([[a[k][a['segment_length'] == v].mean() for k in ('translation', 'rotation')]
                       for v in lengths]).reshape(-1, 2)
    edges = np.unique(np.linspace(a['speed'].min(), a['speed'].max(), SPEED_BINS + 1))
    which = np.searchsorted(edges, a['speed'], side='left') - 1
    by_speed = np.full((max(len(edges) - 1, 0), 2), np.nan)
    for b in range(len(by_speed)):
        hit = which == b
        if hit.any():
            by_speed[b] = a['translation'][hit].mean(), a['rotation'][hit].mean()
    return {'length': lengths, 'by_length': by_len, 'speed': 0.5 * (edges[:-1] + edges[1:]), 'by_speed': by_speed}


def plot_kitti_errors(errors: MetricsContainer) -> matplotlib.figure.Figure:
    """The four KITTI odometry curves: translation [%] and rotation [deg/m] by path length and by speed [km/h]."""
    if len(errors) == 0:
        return plt.figure()
    c = kitti_curves(errors)
    fig, axes = _figure(2, 2, scale=2.0, sharex='row', sharey='col')
    rows = ((c['length'], c['by_length'], 'Path Length [m]'), (c['speed'] * 3.6, c['by_speed'], 'Speed [km/h]'))
    for (x, y, xlabel), (left, right) in zip(rows, axes):
        for ax, col, scale, ylabel, name in ((left, 0, 100.0, 'Translation Error [%]', 'Translation Error'),
                                             (right, 1, np.rad2deg(1.0), 'Rotation Error [deg/m]', 'Rotation Error')):
            keep = ~np.isnan(y[:, col])
            ax.plot(x[keep], y[keep, col] * scale, 'o-', label=name)
            ax.set_xlabel(xlabel), ax.set_ylabel(ylabel)
            ax.legend()
    fig.set_size_inches(10, 8)
    fig.subplots_adjust(hspace=0.3, wspace=0.3)
    return fig


def plot_segment_error_bars(segment_errors: Dict[str, MetricsContainer]) -> matplotlib.figure.Figure:
    """Mean +- std KITTI translation [%] and rotation [deg/m] per sequence, on twin axes."""
    names = list(segment_errors)
    t_mean = [e.mean.translation.kitti * 100 for e in segment_errors.values()]
    t_std = [e.std.translation.kitti * 100 for e in segment_errors.values()]
    r_mean = [np.rad2deg(e.mean.rotation.kitti) for e in segment_errors.values()]
    r_std = [np.rad2deg(e.std.rotation.kitti) for e in segment_errors.values()]
    fig, left = _figure()
    right = left.twinx()
    at, width = np.arange(len(names)), 0.35
    left.bar(at, t_mean, width, yerr=t_std, color='tab:blue')
    right.bar(at + width, r_mean, width, yerr=r_std, color='tab:orange')
    left.set_title('Errors by Dataset')
    left.set_xticks(at + width / 2)
    left.set_xticklabels(names)
    left.set_ylabel('Translation [%]', color='tab:blue'), left.tick_params(axis='y', labelcolor='tab:blue')
    right.set_ylabel('Rotation [deg/m]', color='tab:orange'), right.tick_params(axis='y', labelcolor='tab:orange')
    return fig
