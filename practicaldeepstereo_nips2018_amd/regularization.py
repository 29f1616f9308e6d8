"""3-D hourglass regularization of the matching cost volume on MI355X.

Drop-in mirrors of reference practical_deep_stereo/regularization.py: ``ContractionBlock3d``
(:11-31), ``ExpansionBlock3d`` (:34-57) and ``Regularization`` (:60-126) with the same constructor
arguments, call signatures and state-dict keys (``_smoothing``, ``_contraction_blocks.{0-3}.
{_downsampling_2x,_smoothing}``, ``_expansion_blocks.{0-3}.{_upsampling_2x,_smoothing}``,
``_upsample_to_halfsize``, ``_upsample_to_fullsize``).  All arithmetic runs in libpds_hip.so: every
layer stores LeakyReLU(conv) once plus partial InstanceNorm sums; the normalisation itself, the
skip additions and the broadcast of the left-image shortcut are folded into the loads of the
consuming layer, so no normalised tensor is ever written inside the hourglass.
"""
import ctypes

import torch
from torch import nn

from practicaldeepstereo_nips2018_amd import _lib
from practicaldeepstereo_nips2018_amd import network_blocks


def _block_params(block, tensor_of=None):
    return _lib.conv_block_params(block.conv, block.norm, tensor_of)


class ContractionBlock3d(nn.Module):
    """2x "downsampling" convolution followed by a "smoothing" convolution (regularization.py:11-31)."""

    def __init__(self, number_of_features):
        super(ContractionBlock3d, self).__init__()
        self._downsampling_2x = network_blocks.convolutional_block_3x3x3_stride_2(
            number_of_features, 2 * number_of_features)
        self._smoothing = network_blocks.convolutional_block_3x3x3(
            2 * number_of_features, 2 * number_of_features)
        self._workspace = _lib.Workspace()

    def forward(self, block_input):
        x = _lib.require_gpu_tensor(block_input, 'block_input', 5)
        if x.size(1) != self._downsampling_2x.conv.in_channels:
            raise ValueError('expected %d input features, got %d' %
                             (self._downsampling_2x.conv.in_channels, x.size(1)))
        return _ContractionFunction.apply(self, x, *self.parameters())


class _ContractionFunction(torch.autograd.Function):
    """pds_contraction_block_fwd / _bwd."""

    @staticmethod
    def forward(ctx, module, x, *unused_parameters):
        lib = _lib.load()
        batch, c, d, h, w = x.shape
        shape = (batch, 2 * c, (d + 1) // 2, (h + 1) // 2, (w + 1) // 2)
        down = torch.empty(shape, dtype=torch.float32, device=x.device)
        smooth = torch.empty(shape, dtype=torch.float32, device=x.device)
        pd, ps = _block_params(module._downsampling_2x), _block_params(module._smoothing)
        nbytes = lib.pds_contraction_block_workspace_bytes(batch, c, d, h, w)
        training = any(ctx.needs_input_grad)
        ws = (torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device) if training
              else module._workspace.get(nbytes, x.device))
        with torch.cuda.device(x.device):
            _lib.check(lib.pds_contraction_block_fwd(
                ctypes.byref(pd), ctypes.byref(ps), _lib.ptr(x), _lib.ptr(down), _lib.ptr(smooth),
                batch, c, d, h, w, _lib.ptr(ws), ws.numel(), _lib.stream_handle(x.device)),
                'pds_contraction_block_fwd')
        if training:
            ctx.module = module
            ctx.forward_workspace = ws
            ctx.save_for_backward(x)
        return down, smooth

    @staticmethod
    def backward(ctx, grad_down, grad_smooth):
        lib = _lib.load()
        module = ctx.module
        x, = ctx.saved_tensors
        batch, c, d, h, w = x.shape
        shape = (batch, 2 * c, (d + 1) // 2, (h + 1) // 2, (w + 1) // 2)
        grad_down = (torch.zeros(shape, dtype=torch.float32, device=x.device) if grad_down is None
                     else grad_down.contiguous())
        grad_smooth = (torch.zeros(shape, dtype=torch.float32, device=x.device) if grad_smooth is None
                       else grad_smooth.contiguous())
        grads, tensor_of = _lib.gradient_buffers(module)
        pd, ps = _block_params(module._downsampling_2x), _block_params(module._smoothing)
        gd, gs = _block_params(module._downsampling_2x, tensor_of), _block_params(module._smoothing, tensor_of)
        grad_x = torch.empty_like(x)
        ws = torch.empty(max(int(lib.pds_contraction_block_bwd_workspace_bytes(batch, c, d, h, w)), 256),
                         dtype=torch.uint8, device=x.device)
        fws = _lib.saved_workspace(ctx, 'regularization')
        with torch.cuda.device(x.device):
            _lib.check(lib.pds_contraction_block_bwd(
                ctypes.byref(pd), ctypes.byref(ps), ctypes.byref(gd), ctypes.byref(gs), _lib.ptr(x),
                _lib.ptr(grad_down), _lib.ptr(grad_smooth), _lib.ptr(grad_x), batch, c, d, h, w,
                _lib.ptr(fws), fws.numel(), _lib.ptr(ws), ws.numel(), _lib.stream_handle(x.device)),
                'pds_contraction_block_bwd')
        ctx.forward_workspace = None
        return (None, grad_x) + tuple(grads[id(p)] for p in module.parameters())


class ExpansionBlock3d(nn.Module):
    """2x "upsampling" transposed convolution, skip sum, "smoothing" convolution
    (regularization.py:34-57)."""

    def __init__(self, number_of_features):
        super(ExpansionBlock3d, self).__init__()
        self._upsampling_2x = network_blocks.transposed_convolutional_block_4x4x4_stride_2(
            number_of_features, number_of_features // 2)
        self._smoothing = network_blocks.convolutional_block_3x3x3(
            number_of_features // 2, number_of_features // 2)
        self._workspace = _lib.Workspace()

    def forward(self, block_input, shortcut_from_contraction):
        x = _lib.require_gpu_tensor(block_input, 'block_input', 5)
        shortcut = _lib.require_gpu_tensor(shortcut_from_contraction, 'shortcut_from_contraction', 5)
        batch, c, d, h, w = x.shape
        if c != self._upsampling_2x.conv.in_channels:
            raise ValueError('expected %d input features, got %d' % (self._upsampling_2x.conv.in_channels, c))
        if tuple(shortcut.shape) != (batch, c // 2, 2 * d, 2 * h, 2 * w):
            raise ValueError('shortcut of shape %s does not match upsampled input %s' %
                             (tuple(shortcut.shape), (batch, c // 2, 2 * d, 2 * h, 2 * w)))
        return _ExpansionFunction.apply(self, x, shortcut, *self.parameters())


class _ExpansionFunction(torch.autograd.Function):
    """pds_expansion_block_fwd / _bwd."""

    @staticmethod
    def forward(ctx, module, x, shortcut, *unused_parameters):
        lib = _lib.load()
        batch, c, d, h, w = x.shape
        out = torch.empty_like(shortcut)
        pu, ps = _block_params(module._upsampling_2x), _block_params(module._smoothing)
        nbytes = lib.pds_expansion_block_workspace_bytes(batch, c, d, h, w)
        training = any(ctx.needs_input_grad)
        ws = (torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device) if training
              else module._workspace.get(nbytes, x.device))
        with torch.cuda.device(x.device):
            _lib.check(lib.pds_expansion_block_fwd(
                ctypes.byref(pu), ctypes.byref(ps), _lib.ptr(x), _lib.ptr(shortcut), _lib.ptr(out),
                batch, c, d, h, w, _lib.ptr(ws), ws.numel(), _lib.stream_handle(x.device)),
                'pds_expansion_block_fwd')
        if training:
            ctx.module = module
            ctx.forward_workspace = ws
            ctx.save_for_backward(x, shortcut)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        module = ctx.module
        x, shortcut = ctx.saved_tensors
        batch, c, d, h, w = x.shape
        grad_out = grad_out.contiguous()
        grads, tensor_of = _lib.gradient_buffers(module)
        pu, ps = _block_params(module._upsampling_2x), _block_params(module._smoothing)
        gu, gs = _block_params(module._upsampling_2x, tensor_of), _block_params(module._smoothing, tensor_of)
        grad_x, grad_shortcut = torch.empty_like(x), torch.empty_like(shortcut)
        ws = torch.empty(max(int(lib.pds_expansion_block_bwd_workspace_bytes(batch, c, d, h, w)), 256),
                         dtype=torch.uint8, device=x.device)
        fws = _lib.saved_workspace(ctx, 'regularization')
        with torch.cuda.device(x.device):
            _lib.check(lib.pds_expansion_block_bwd(
                ctypes.byref(pu), ctypes.byref(ps), ctypes.byref(gu), ctypes.byref(gs), _lib.ptr(x),
                _lib.ptr(shortcut), _lib.ptr(grad_out), _lib.ptr(grad_x), _lib.ptr(grad_shortcut),
                batch, c, d, h, w, _lib.ptr(fws), fws.numel(), _lib.ptr(ws), ws.numel(),
                _lib.stream_handle(x.device)), 'pds_expansion_block_bwd')
        ctx.forward_workspace = None
        return (None, grad_x, grad_shortcut) + tuple(grads[id(p)] for p in module.parameters())


class Regularization(_lib.FrozenWeightsMixin, nn.Module):
    """Hourglass 3-D network: /16 contraction, expansion back, then x2 (D, H, W) and x2 (H, W)
    up-sampling; returns matching cost for even disparities (regularization.py:60-126)."""

    def __init__(self, number_of_features=8):
        super(Regularization, self).__init__()
        self._smoothing = network_blocks.convolutional_block_3x3x3(
            number_of_features, number_of_features)
        self._contraction_blocks = nn.ModuleList(
            [ContractionBlock3d(number_of_features * scale) for scale in (1, 2, 4, 8)])
        self._expansion_blocks = nn.ModuleList(
            [ExpansionBlock3d(number_of_features * scale) for scale in (16, 8, 4, 2)])
        self._upsample_to_halfsize = network_blocks.transposed_convolutional_block_4x4x4_stride_2(
            number_of_features, number_of_features // 2)
        self._upsample_to_fullsize = network_blocks.transposed_convolution_3x4x4_stride_122(
            number_of_features // 2, 1)
        self._workspace = _lib.Workspace()

    @property
    def number_of_features(self):
        return self._smoothing.conv.in_channels

    def native_params(self, tensor_of=None):
        """PdsRegularizationParams of the parameters, or of the tensors ``tensor_of`` maps them to."""
        params = _lib.RegularizationParams()
        params.features = self.number_of_features
        params.smoothing = _block_params(self._smoothing, tensor_of)
        for level in range(4):
            contraction, expansion = self._contraction_blocks[level], self._expansion_blocks[level]
            params.contraction[level][0] = _block_params(contraction._downsampling_2x, tensor_of)
            params.contraction[level][1] = _block_params(contraction._smoothing, tensor_of)
            params.expansion[level][0] = _block_params(expansion._upsampling_2x, tensor_of)
            params.expansion[level][1] = _block_params(expansion._smoothing, tensor_of)
        params.upsample_half = _block_params(self._upsample_to_halfsize, tensor_of)
        params.upsample_full = _lib.conv_block_params(self._upsample_to_fullsize, None, tensor_of)
        return params

    def can_fold_crop(self, estimator):
        """The fused kernel (4 half-resolution features, at most 4 taps per side) is the one that folds the crop."""
        taps = -(-estimator._half_support_window // estimator._disparity_step)
        return self.number_of_features == 8 and 1 <= taps <= 4

    def _check_inputs(self, matching_signatures, shortcut_from_left_image):
        ms = _lib.require_gpu_tensor(matching_signatures, 'matching_signatures', 5)
        shortcut = _lib.require_gpu_tensor(shortcut_from_left_image, 'shortcut_from_left_image', 4)
        batch, c, d, h, w = ms.shape
        if c != self.number_of_features:
            raise ValueError('expected %d matching signature features, got %d' % (self.number_of_features, c))
        if tuple(shortcut.shape) != (batch, c, h, w):
            raise ValueError('shortcut of shape %s does not match signatures %s' %
                             (tuple(shortcut.shape), tuple(ms.shape)))
        return ms, shortcut

    def forward(self, matching_signatures, shortcut_from_left_image):
        """[batch, 8, D, h, w] + [batch, 8, h, w] -> matching cost [batch, 2D, 4h, 4w]."""
        ms, shortcut = self._check_inputs(matching_signatures, shortcut_from_left_image)
        return _RegularizationFunction.apply(self, ms, shortcut, None, *self.parameters())

    def forward_with_estimator(self, matching_signatures, shortcut_from_left_image, estimator, crop=(0, 0)):
        """Eval-mode fusion used by PdsNetwork: Regularization followed by SubpixelMap without
        materialising the full-resolution cost volume (network.py:50-51).  ``crop`` = (rows, columns)
        SizeAdapter.pad added on top / left (size_adapter.py:29-43): the crop of ``unpad`` (:45-52) is folded
        into the store.  -> contiguous [batch, 4h - rows, 4w - columns]."""
        ms, shortcut = self._check_inputs(matching_signatures, shortcut_from_left_image)
        window = (estimator._half_support_window, estimator._disparity_step, int(crop[0]), int(crop[1]))
        return _RegularizationFunction.apply(self, ms, shortcut, window, *self.parameters())


class _RegularizationFunction(torch.autograd.Function):
    """pds_regularization_fwd / _bwd (and the eval-only fusion with the estimator).  When a gradient is
    needed the forward runs in a workspace of its own that is kept, with the inputs, until backward."""

    @staticmethod
    def forward(ctx, module, ms, shortcut, estimator_window, *unused_parameters):
        lib = _lib.load()
        batch, _, d, h, w = ms.shape
        params = module.native_params()
        nbytes = lib.pds_regularization_workspace_bytes(ctypes.byref(params), batch, d, h, w)
        if nbytes == 0:
            raise ValueError(lib.pds_last_error().decode())
        training = any(ctx.needs_input_grad) and estimator_window is None
        # outputs first: nothing may fail between taking the workspace and the native call
        if estimator_window is None:
            out = torch.empty((batch, 2 * d, 4 * h, 4 * w), dtype=torch.float32, device=ms.device)
        else:
            crop_top, crop_left = estimator_window[2], estimator_window[3]
            out = torch.empty((batch, 4 * h - crop_top, 4 * w - crop_left), dtype=torch.float32, device=ms.device)
        token = None
        if training:
            ws, resident = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=ms.device), False
        else:
            # a frozen module's workspace keeps the re-laid-out weights: skipped when it last completed a call with
            # these shapes, this entry point / estimator window (the arena layout of the fused tail depends on the
            # window: at most 4 taps per side takes the fused trunk) and these parameter values
            window = None if estimator_window is None else (estimator_window[0], estimator_window[1])
            ws, resident, token = module._workspace.get_resident(
                nbytes, ms.device, _lib.resident_key(module, module, (batch, d, h, w, window)))
        with torch.cuda.device(ms.device):
            if estimator_window is None:
                _lib.check(lib.pds_regularization_fwd(
                    ctypes.byref(params), _lib.ptr(ms), _lib.ptr(shortcut), _lib.ptr(out),
                    batch, d, h, w, _lib.ptr(ws), ws.numel(), int(resident), _lib.stream_handle(ms.device)),
                    'pds_regularization_fwd')
            else:
                _lib.check(lib.pds_regularization_subpixel_map_fwd(
                    ctypes.byref(params), _lib.ptr(ms), _lib.ptr(shortcut), _lib.ptr(out),
                    batch, d, h, w, estimator_window[0], estimator_window[1], crop_top, crop_left,
                    _lib.ptr(ws), ws.numel(), int(resident), _lib.stream_handle(ms.device)),
                    'pds_regularization_subpixel_map_fwd')
        if token is not None:
            module._workspace.commit(token)
        if training:
            ctx.module = module
            ctx.forward_workspace = ws
            ctx.save_for_backward(ms, shortcut)
        else:
            ctx.module = None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        module = ctx.module
        if module is None:
            _lib.not_differentiable('Regularization fused with SubpixelMap (inference only, estimator.py:19)')
        lib = _lib.load()
        ms, shortcut = ctx.saved_tensors
        batch, _, d, h, w = ms.shape
        grad_out = grad_out.contiguous()
        params = module.native_params()
        grads, tensor_of = _lib.gradient_buffers(module)
        grad_params = module.native_params(tensor_of)
        grad_ms = torch.empty_like(ms)
        grad_shortcut = torch.empty_like(shortcut)
        nbytes = lib.pds_regularization_bwd_workspace_bytes(ctypes.byref(params), batch, d, h, w)
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=ms.device)
        fws = _lib.saved_workspace(ctx, 'regularization')
        with torch.cuda.device(ms.device):
            _lib.check(lib.pds_regularization_bwd(
                ctypes.byref(params), ctypes.byref(grad_params), _lib.ptr(ms), _lib.ptr(shortcut),
                _lib.ptr(grad_out), _lib.ptr(grad_ms), _lib.ptr(grad_shortcut), batch, d, h, w,
                _lib.ptr(fws), fws.numel(), _lib.ptr(ws), ws.numel(), _lib.stream_handle(ms.device)),
                'pds_regularization_bwd')
        ctx.forward_workspace = None
        return (None, grad_ms, grad_shortcut, None) + tuple(grads[id(p)] for p in module.parameters())
