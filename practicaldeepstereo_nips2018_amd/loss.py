"""Sub-pixel cross-entropy loss on MI355X.

Drop-in mirror of reference practical_deep_stereo/loss.py:16-78 (``SubpixelCrossEntropy``): same constructor
(``diversity``, ``disparity_step``) and ``forward(similarities, ground_truth_disparities, weights=None)``;
ground truth ``inf`` marks unknown pixels.  Value and gradient come from two streaming HIP kernels
(``pds_subpixel_cross_entropy_fwd`` / ``_bwd``): one pass over the similarity volume forward, one read + one
write backward, instead of the reference's log-softmax copy plus a Python loop over the planes.
The gradient flows to ``similarities`` and, when ``weights`` requires it, to ``weights``
(``pds_subpixel_cross_entropy_weights_bwd``: d loss / d w = (entropy - loss) / (sum w + 1e-15) at known pixels,
loss.py:74-77 under autograd); the ground truth is a constant, as in the reference (its ``.data`` mask, loss.py:52).
"""
import ctypes

import torch
from torch import nn

from practicaldeepstereo_nips2018_amd import _lib


class SubpixelCrossEntropy(nn.Module):
    def __init__(self, diversity=1.0, disparity_step=2):
        super(SubpixelCrossEntropy, self).__init__()
        self._diversity = diversity
        self._disparity_step = disparity_step

    def forward(self, similarities, ground_truth_disparities, weights=None):
        """similarities [example, disparity_index, y, x]; ground truth [example, y, x] -> scalar loss."""
        sim = _lib.require_gpu_tensor(similarities, 'similarities', 4)
        gt = _lib.require_gpu_tensor(ground_truth_disparities.detach(), 'ground_truth_disparities', 3)
        if tuple(gt.shape) != (sim.size(0), sim.size(2), sim.size(3)):
            raise ValueError('ground truth of shape %s does not match similarities %s' %
                             (tuple(gt.shape), tuple(sim.shape)))
        w = None
        if weights is not None:
            w = _lib.require_gpu_tensor(weights, 'weights', 3)
            if w.shape != gt.shape:
                raise ValueError('weights of shape %s do not match the ground truth %s' %
                                 (tuple(w.shape), tuple(gt.shape)))
        return _SubpixelCrossEntropyFunction.apply(sim, gt, w, float(self._diversity), int(self._disparity_step))


class _SubpixelCrossEntropyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sim, gt, weights, diversity, step):
        lib = _lib.load()
        n, planes, h, w = sim.shape
        loss = torch.empty((), dtype=torch.float32, device=sim.device)
        lse = torch.empty((n, h, w), dtype=torch.float32, device=sim.device)
        stats = torch.empty(2, dtype=torch.float32, device=sim.device)
        ws = torch.empty(lib.pds_subpixel_cross_entropy_workspace_bytes(n, h, w), dtype=torch.uint8,
                         device=sim.device)
        with torch.cuda.device(sim.device):
            _lib.check(lib.pds_subpixel_cross_entropy_fwd(
                _lib.ptr(sim), _lib.ptr(gt), _lib.ptr(weights) if weights is not None else None,
                _lib.ptr(loss), _lib.ptr(lse), _lib.ptr(stats), n, planes, h, w, diversity, step,
                _lib.ptr(ws), ws.numel(), _lib.stream_handle(sim.device)), 'pds_subpixel_cross_entropy_fwd')
        ctx.save_for_backward(sim, gt, lse, stats)
        ctx.weights = weights.detach() if weights is not None else None
        ctx.config = (diversity, step)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        sim, gt, lse, stats = ctx.saved_tensors
        weights = ctx.weights
        diversity, step = ctx.config
        n, planes, h, w = sim.shape
        grad_loss = grad_loss.to(torch.float32).contiguous().view(1)
        grad_sim = torch.empty_like(sim)
        with torch.cuda.device(sim.device):
            _lib.check(lib.pds_subpixel_cross_entropy_bwd(
                _lib.ptr(sim), _lib.ptr(gt), _lib.ptr(weights) if weights is not None else None,
                _lib.ptr(lse), _lib.ptr(stats), _lib.ptr(grad_loss), _lib.ptr(grad_sim),
                n, planes, h, w, diversity, step, _lib.stream_handle(sim.device)),
                'pds_subpixel_cross_entropy_bwd')
        grad_weights = None
        if weights is not None and ctx.needs_input_grad[2]:
            grad_weights = torch.empty_like(weights)
            with torch.cuda.device(sim.device):
                _lib.check(lib.pds_subpixel_cross_entropy_weights_bwd(
                    _lib.ptr(sim), _lib.ptr(gt), _lib.ptr(lse), _lib.ptr(stats), _lib.ptr(grad_loss),
                    _lib.ptr(grad_weights), n, planes, h, w, diversity, step, _lib.stream_handle(sim.device)),
                    'pds_subpixel_cross_entropy_weights_bwd')
        return grad_sim, None, grad_weights, None, None
