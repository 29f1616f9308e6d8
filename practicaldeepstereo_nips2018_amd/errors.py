"""Evaluation metrics on MI355X.

Drop-in mirror of reference practical_deep_stereo/errors.py:9-74 (``compute_absolute_error``,
``compute_n_pixels_error``; called per example by pds_trainer.py:48-58): same arguments, same return values
(pixel-wise map on the device of the inputs, average as a Python float), ground truth ``inf`` marks unknown
pixels.  Both run through ``pds_disparity_errors_fwd``: one streaming pass that writes the pixel-wise map and
the three fp64 sums (absolute error, known pixels, pixels beyond ``n``), so the disparity map is never copied
to the host; ``compute_errors`` returns both metrics from a single pass.
"""
import ctypes

import torch

from practicaldeepstereo_nips2018_amd import _lib


def _run(estimated_disparity, ground_truth_disparity, n, want_absolute, want_n_pixels):
    est = _lib.require_gpu_tensor(estimated_disparity.detach(), 'estimated_disparity')
    gt = _lib.require_gpu_tensor(ground_truth_disparity.detach(), 'ground_truth_disparity')
    if est.shape != gt.shape:
        raise ValueError('estimated disparity of shape %s does not match the ground truth %s' %
                         (tuple(est.shape), tuple(gt.shape)))
    lib = _lib.load()
    count = est.numel()
    absolute = torch.empty_like(est) if want_absolute else None
    n_pixels = torch.empty_like(est) if want_n_pixels else None
    stats = torch.zeros(3, dtype=torch.float64, device=est.device)
    if count:
        ws = torch.empty(lib.pds_disparity_errors_workspace_bytes(count), dtype=torch.uint8, device=est.device)
        with torch.cuda.device(est.device):
            _lib.check(lib.pds_disparity_errors_fwd(
                _lib.ptr(est), _lib.ptr(gt), count, float(n),
                _lib.ptr(absolute) if want_absolute else None, _lib.ptr(n_pixels) if want_n_pixels else None,
                _lib.ptr(stats), _lib.ptr(ws), ws.numel(), _lib.stream_handle(est.device)),
                'pds_disparity_errors_fwd')
    return absolute, n_pixels, stats


def compute_absolute_error(estimated_disparity, ground_truth_disparity, use_mean=True):
    """(pixel-wise absolute error, mean | median absolute error over pixels with ground truth) -- errors.py:9-44.
    Without any known pixel the average is 0.0."""
    absolute, _, stats = _run(estimated_disparity, ground_truth_disparity, 0.0, True, False)
    total, known, _ = stats.tolist()
    if known == 0:
        return absolute, 0.0
    if use_mean:
        return absolute, total / known
    # the median is an order statistic, not a streaming sum: selected by torch on the device (errors.py:41-43)
    return absolute, absolute[~torch.isinf(ground_truth_disparity)].median().item()


def compute_n_pixels_error(estimated_disparity, ground_truth_disparity, n=3.0):
    """(pixel-wise n-pixels error, % of pixels with ground truth whose error exceeds n) -- errors.py:47-74."""
    _, n_pixels, stats = _run(estimated_disparity, ground_truth_disparity, n, False, True)
    _, known, bad = stats.tolist()
    return n_pixels, (100.0 * bad / known if known else 0.0)


def compute_errors(estimated_disparity, ground_truth_disparity, n=3.0):
    """Both metrics of pds_trainer.py:48-58 from ONE pass: returns (binary error map,
    {'three_pixels_error': %, 'mean_absolute_error': px})."""
    _, n_pixels, stats = _run(estimated_disparity, ground_truth_disparity, n, False, True)
    total, known, bad = stats.tolist()
    return n_pixels, {'three_pixels_error': 100.0 * bad / known if known else 0.0,
                      'mean_absolute_error': total / known if known else 0.0}
