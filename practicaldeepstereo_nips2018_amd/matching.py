"""Matching and MatchingOperation on MI355X.

Drop-in mirrors of reference practical_deep_stereo/matching.py:16-63 (``Matching``) and :66-112
(``MatchingOperation``): same constructor arguments, ``set_maximum_disparity``, ``forward``
signature, output layout ``[batch, features, disparity, y, x]`` and state-dict keys
(``_operation._matching_operation_modules.{0..3}...``).  The arithmetic runs in libpds_hip.so:

* ``Matching`` with a ``MatchingOperation`` takes the fused path ``pds_matching_fwd``: the linear
  first convolution is factorised into conv_L(left) + shift_d(conv_R(right)) so the right
  descriptor is convolved once for all disparities; the 64->64 convolutions run as implicit GEMMs
  over all disparity planes at once on the 16-bit matrix pipe -- every fp32 operand is split into
  an fp16 high and low part (22 of 24 significand bits, scaled by exact powers of two taken from
  range certificates of the data), three partial products per multiply, fp32 accumulation
  (csrc/conv2d_x3.hip; NOT exact-fp32 products: measured 2-5e-7 of the output scale per layer
  against fp64, the fp32 CPU reference itself is 3e-7) -- with LeakyReLU and the per-plane
  InstanceNorm statistics fused, and the last convolution writes straight into the stacked layout.
  Tensors without a certificate take a three-way bf16 split (six products) or exact-fp32 MFMAs.
* ``Matching`` with any other callable builds cat([left, S_d(right)]) for all disparities with one
  HIP kernel (``pds_shift_concat_fwd``) and applies the callable per plane, like the reference.
"""
import ctypes

import torch
from torch import nn

from practicaldeepstereo_nips2018_amd import _lib
from practicaldeepstereo_nips2018_amd import network_blocks


class MatchingOperation(nn.Module):
    """Operation applied to concatenated left / right descriptors (matching.py:66-112)."""

    def __init__(self,
                 number_of_concatenated_descriptor_features=128,
                 number_of_features=64,
                 number_of_compact_matching_signature_features=8,
                 number_of_residual_blocks=2):
        super(MatchingOperation, self).__init__()
        if number_of_concatenated_descriptor_features % 2 != 0:
            raise ValueError('"number_of_concatenated_descriptor_features" should be even.')
        self._number_of_residual_blocks = number_of_residual_blocks
        layers = [network_blocks.convolution_3x3(number_of_concatenated_descriptor_features,
                                                 number_of_features)]
        for _ in range(number_of_residual_blocks):
            layers.append(network_blocks.ResidualBlock(number_of_features))
        layers.append(network_blocks.convolution_3x3(
            number_of_features, number_of_compact_matching_signature_features))
        self._matching_operation_modules = nn.ModuleList(layers)
        self._workspace = _lib.Workspace()

    # -- geometry ------------------------------------------------------------------------------
    @property
    def number_of_features(self):
        return self._matching_operation_modules[0].out_channels

    @property
    def number_of_descriptor_features(self):
        return self._matching_operation_modules[0].in_channels // 2

    @property
    def number_of_signature_features(self):
        return self._matching_operation_modules[-1].out_channels

    def supports_fused_matching(self):
        """The factorised first layer needs the descriptor width to equal the feature width
        (128 -> 64 in the reference: two 64-channel descriptors)."""
        return self.number_of_descriptor_features == self.number_of_features

    def supports_native_training(self):
        """pds_matching_train_fwd / pds_matching_bwd differentiate the factorised first layer with the matrix-pipe weight
        gradient kernels, which take 64 or at most 16 output channels (csrc/wgrad2d_mfma.hip: wgrad2d_mfma_supported);
        other widths train through the generic shift / concat route, whose operation has a backward for any width."""
        return self.supports_fused_matching() and (self.number_of_features == 64 or self.number_of_features <= 16)

    def native_params(self, tensor_of=None):
        """(PdsMatchingParams, keep-alive list) pointing at this module's parameters, or, with
        ``tensor_of``, at the tensors it maps them to (gradient buffers)."""
        mods = self._matching_operation_modules
        blocks = []
        for residual in mods[1:-1]:
            for block in residual.convolutions:
                blocks.append(_lib.conv_block_params(block.conv, block.norm, tensor_of))
        array = (_lib.ConvBlockParams * max(len(blocks), 1))(*blocks)
        params = _lib.MatchingParams()
        params.features = self.number_of_features
        params.signature_features = self.number_of_signature_features
        params.residual_blocks = self._number_of_residual_blocks
        params.first = _lib.conv_block_params(mods[0], None, tensor_of)
        params.blocks = ctypes.cast(array, ctypes.POINTER(_lib.ConvBlockParams))
        params.last = _lib.conv_block_params(mods[-1], None, tensor_of)
        return params, array

    def forward(self, concatenated_descriptors):
        """[batch, 128, h, w] -> compact matching signature [batch, 8, h, w] (matching.py:97-112)."""
        x = _lib.require_gpu_tensor(concatenated_descriptors, 'concatenated_descriptors', 4)
        if x.size(1) != self._matching_operation_modules[0].in_channels:
            raise ValueError('expected %d concatenated descriptor features, got %d' %
                             (self._matching_operation_modules[0].in_channels, x.size(1)))
        return _MatchingOperationFunction.apply(self, x, *self.parameters())


class _MatchingOperationFunction(torch.autograd.Function):
    """pds_matching_operation_fwd / _bwd.  When a gradient is needed the forward runs in a workspace of its
    own, kept (with the input) until backward: the backward entry point re-derives every intermediate
    from it."""

    @staticmethod
    def forward(ctx, module, x, *unused_parameters):
        lib = _lib.load()
        n, _, h, w = x.shape
        params, keep = module.native_params()
        out = torch.empty((n, module.number_of_signature_features, h, w), dtype=torch.float32,
                          device=x.device)
        nbytes = lib.pds_matching_operation_workspace_bytes(ctypes.byref(params), n, h, w)
        training = any(ctx.needs_input_grad)
        if training:
            ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device)
        else:
            ws = module._workspace.get(nbytes, x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.pds_matching_operation_fwd(
                ctypes.byref(params), _lib.ptr(x), _lib.ptr(out), n, h, w,
                _lib.ptr(ws), ws.numel(), _lib.stream_handle(x.device)), 'pds_matching_operation_fwd')
        del keep
        if training:
            ctx.module = module
            ctx.forward_workspace = ws
            ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        module = ctx.module
        x, = ctx.saved_tensors
        n, _, h, w = x.shape
        grad_out = grad_out.contiguous()
        params, keep = module.native_params()
        grads, tensor_of = _lib.gradient_buffers(module)
        grad_params, keep_grads = module.native_params(tensor_of)
        grad_x = torch.empty_like(x)
        nbytes = lib.pds_matching_operation_bwd_workspace_bytes(ctypes.byref(params), n, h, w)
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device)
        fws = _lib.saved_workspace(ctx, 'matching')
        with torch.cuda.device(x.device):
            _lib.check(lib.pds_matching_operation_bwd(
                ctypes.byref(params), ctypes.byref(grad_params), _lib.ptr(x), _lib.ptr(grad_out),
                _lib.ptr(grad_x), n, h, w, _lib.ptr(fws), fws.numel(), _lib.ptr(ws), ws.numel(),
                _lib.stream_handle(x.device)), 'pds_matching_operation_bwd')
        del keep, keep_grads
        ctx.forward_workspace = None
        return (None, grad_x) + tuple(grads[id(p)] for p in module.parameters())


class _ShiftConcatFunction(torch.autograd.Function):
    """cat([left, S_d(right)], 1) for a range of disparities (matching.py:50-61) and its adjoint."""

    @staticmethod
    def forward(ctx, left, right, begin, count):
        lib = _lib.load()
        batch, channels, h, w = left.shape
        out = torch.empty((count, batch, 2 * channels, h, w), dtype=torch.float32, device=left.device)
        with torch.cuda.device(left.device):
            _lib.check(lib.pds_shift_concat_fwd(
                _lib.ptr(left), _lib.ptr(right), _lib.ptr(out), batch, channels, h, w,
                begin, count, _lib.stream_handle(left.device)), 'pds_shift_concat_fwd')
        ctx.geometry = (batch, channels, h, w, begin, count)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        batch, channels, h, w, begin, count = ctx.geometry
        grad_out = grad_out.contiguous()
        grad_left = torch.empty((batch, channels, h, w), dtype=torch.float32, device=grad_out.device)
        grad_right = torch.empty_like(grad_left)
        with torch.cuda.device(grad_out.device):
            _lib.check(lib.pds_shift_concat_bwd(
                _lib.ptr(grad_out), _lib.ptr(grad_left), _lib.ptr(grad_right), batch, channels, h, w,
                begin, count, _lib.stream_handle(grad_out.device)), 'pds_shift_concat_bwd')
        return grad_left, grad_right, None, None


class Matching(_lib.FrozenWeightsMixin, nn.Module):
    """Applies ``operation`` to cat([left, right shifted by d]) for d in [0, maximum_disparity]
    and stacks the results on dim 2 (matching.py:16-63)."""

    def __init__(self, maximum_disparity, operation):
        super(Matching, self).__init__()
        self._maximum_disparity = maximum_disparity
        self._operation = operation
        self._workspace = _lib.Workspace()
        # [begin, count) of the disparity planes this process computes; None = all
        self._disparity_shard = None

    def set_maximum_disparity(self, maximum_disparity):
        self._maximum_disparity = maximum_disparity

    def set_disparity_shard(self, shard):
        """shard = (first_plane, number_of_planes) or None; used by the multi-GPU wrapper."""
        self._disparity_shard = shard

    def _plane_range(self):
        if self._disparity_shard is None:
            return 0, self._maximum_disparity + 1
        begin, count = self._disparity_shard
        if begin < 0 or count < 1 or begin + count > self._maximum_disparity + 1:
            raise ValueError('disparity shard (%d, %d) outside [0, %d]' %
                             (begin, count, self._maximum_disparity))
        return begin, count

    def forward(self, left_embedding, right_embedding):
        """left/right [batch, features, y, x] -> [batch, op features, disparity, y, x]."""
        left = _lib.require_gpu_tensor(left_embedding, 'left_embedding', 4)
        right = _lib.require_gpu_tensor(right_embedding, 'right_embedding', 4)
        if left.shape != right.shape:
            raise ValueError('left and right embeddings differ in shape: %s vs %s' %
                             (tuple(left.shape), tuple(right.shape)))
        begin, count = self._plane_range()
        operation = self._operation
        needs_grad = torch.is_grad_enabled() and (
            left.requires_grad or right.requires_grad or
            (isinstance(operation, nn.Module) and any(p.requires_grad for p in operation.parameters())))
        if (isinstance(operation, MatchingOperation) and operation.supports_fused_matching()
                and left.size(1) == operation.number_of_descriptor_features):
            if not needs_grad:
                return _FusedMatchingFunction.apply(self, left, right, begin, count,
                                                    *operation.parameters())
            _lib.warn_eval_with_grad(self)
            if operation.supports_native_training():
                # Training: the differentiable route keeps the factorised first layer (no [D', B, 128, h, w] concat) and
                # every layer output for the backward pass (pds_matching_train_fwd / pds_matching_bwd).
                return _TrainMatchingFunction.apply(self, left, right, begin, count, *operation.parameters())
        return self._forward_generic(left, right, begin, count)

    def _forward_generic(self, left, right, begin, count):
        concatenated = _ShiftConcatFunction.apply(left, right, begin, count)
        return torch.stack([self._operation(plane) for plane in concatenated.unbind(0)], dim=2)


class _TrainMatchingFunction(torch.autograd.Function):
    """pds_matching_train_fwd / pds_matching_bwd: Matching + MatchingOperation under autograd (matching.py:34-63 as
    pds_trainer.py:40-46 drives it).  The forward runs in a workspace of its own, kept until backward; the gradient
    of the first (linear) layer is taken through its factorisation: one reduction of d loss / d x0 over the disparity
    planes, then single-plane convolution gradients."""

    @staticmethod
    def forward(ctx, module, left, right, begin, count, *unused_parameters):
        lib = _lib.load()
        operation = module._operation
        left, right = left.contiguous(), right.contiguous()
        batch, _, h, w = left.shape
        params, keep = operation.native_params()
        out = torch.empty((batch, operation.number_of_signature_features, count, h, w),
                          dtype=torch.float32, device=left.device)
        nbytes = _lib.planned_bytes(lib.pds_matching_train_workspace_bytes(ctypes.byref(params), batch, h, w, count),
                                    'pds_matching_train_workspace_bytes')
        ws = torch.empty(nbytes, dtype=torch.uint8, device=left.device)
        with torch.cuda.device(left.device):
            _lib.check(lib.pds_matching_train_fwd(
                ctypes.byref(params), _lib.ptr(left), _lib.ptr(right), _lib.ptr(out), batch, h, w, begin, count,
                _lib.ptr(ws), ws.numel(), _lib.stream_handle(left.device)), 'pds_matching_train_fwd')
        del keep
        ctx.module = module
        ctx.geometry = (begin, count)
        ctx.forward_workspace = ws
        ctx.save_for_backward(left, right)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        operation = ctx.module._operation
        left, right = ctx.saved_tensors
        begin, count = ctx.geometry
        batch, _, h, w = left.shape
        grad_out = grad_out.contiguous()
        params, keep = operation.native_params()
        grads, tensor_of = _lib.gradient_buffers(operation)
        grad_params, keep_grads = operation.native_params(tensor_of)
        grad_left, grad_right = torch.empty_like(left), torch.empty_like(right)
        nbytes = _lib.planned_bytes(lib.pds_matching_bwd_workspace_bytes(ctypes.byref(params), batch, h, w, count),
                                    'pds_matching_bwd_workspace_bytes')
        ws = torch.empty(nbytes, dtype=torch.uint8, device=left.device)
        fws = _lib.saved_workspace(ctx, 'matching')
        with torch.cuda.device(left.device):
            _lib.check(lib.pds_matching_bwd(
                ctypes.byref(params), ctypes.byref(grad_params), _lib.ptr(left), _lib.ptr(right), _lib.ptr(grad_out),
                _lib.ptr(grad_left), _lib.ptr(grad_right), batch, h, w, begin, count, _lib.ptr(fws), fws.numel(),
                _lib.ptr(ws), ws.numel(), _lib.stream_handle(left.device)), 'pds_matching_bwd')
        del keep, keep_grads
        ctx.forward_workspace = None
        return (None, grad_left, grad_right, None, None) + tuple(grads[id(p)] for p in operation.parameters())


class _FusedMatchingFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, left, right, begin, count, *unused_parameters):
        lib = _lib.load()
        operation = module._operation
        batch, _, h, w = left.shape
        params, keep = operation.native_params()
        out = torch.empty((batch, operation.number_of_signature_features, count, h, w),
                          dtype=torch.float32, device=left.device)
        nbytes = lib.pds_matching_workspace_bytes(ctypes.byref(params), batch, h, w, count)
        # a frozen module's workspace keeps the re-laid-out weights: skipped when it last completed a call with these
        # shapes and parameter values (the key is committed only after the native call succeeded)
        ws, resident, token = module._workspace.get_resident(
            nbytes, left.device, _lib.resident_key(module, operation, (batch, h, w, count)))
        with torch.cuda.device(left.device):
            _lib.check(lib.pds_matching_fwd(
                ctypes.byref(params), _lib.ptr(left), _lib.ptr(right), _lib.ptr(out),
                batch, h, w, begin, count, _lib.ptr(ws), ws.numel(), int(resident),
                _lib.stream_handle(left.device)), 'pds_matching_fwd')
        module._workspace.commit(token)
        del keep
        return out

    @staticmethod
    def backward(ctx, *grads):
        _lib.not_differentiable('Matching (fused inference path)')
