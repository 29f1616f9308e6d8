"""Sub-pixel MAP disparity estimator on MI355X.

Drop-in mirror of reference practical_deep_stereo/estimator.py:10-91 (``SubpixelMap``): a plain
callable object (not an nn.Module), the same three ``ValueError`` checks, inference only.  The
computation is one streaming HIP kernel (``pds_subpixel_map_fwd``) that reads the similarity
volume exactly once.
"""
import torch

from practicaldeepstereo_nips2018_amd import _lib


class SubpixelMap(object):
    """Approximate sub-pixel MAP: softmax-weighted mean of the disparities within
    ``half_support_window`` pixels of the arg-max (estimator.py:10-20)."""

    def __init__(self, half_support_window=4, disparity_step=2):
        if disparity_step < 1:
            raise ValueError('"disparity_step" should be positive integer.')
        if half_support_window < 1:
            raise ValueError('"half_support_window" should be positive integer.')
        if half_support_window % disparity_step != 0:
            raise ValueError('"half_support_window" should be multiple of the'
                             '"disparity_step"')
        self._disparity_step = disparity_step
        self._half_support_window = half_support_window

    def __call__(self, similarities):
        """similarities [batch, disparity_index, y, x] -> disparities [batch, y, x]."""
        sim = _lib.require_gpu_tensor(similarities.detach(), 'similarities', 4)
        lib = _lib.load()
        batch, planes, height, width = sim.shape
        out = torch.empty((batch, height, width), dtype=torch.float32, device=sim.device)
        with torch.cuda.device(sim.device):
            _lib.check(lib.pds_subpixel_map_fwd(
                _lib.ptr(sim), _lib.ptr(out), batch, planes, height, width,
                self._half_support_window, self._disparity_step,
                _lib.stream_handle(sim.device)), 'pds_subpixel_map_fwd')
        return out
