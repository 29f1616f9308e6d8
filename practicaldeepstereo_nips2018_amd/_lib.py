"""ctypes binding of libpds_hip.so (C ABI: include/pds_hip.h).

The library is built in-tree by ``build_library()`` (``hipcc --offload-arch=gfx950``) and loaded
from this package directory.  There is NO fallback: if the shared object is missing or a tensor is
not a contiguous fp32 GPU tensor, the call raises.
"""
import ctypes
import os
import subprocess
import threading

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG_DIR, 'csrc')
# PDS_HIP_LIB overrides the library path (A/B experiments with alternative builds)
LIB_PATH = os.environ.get('PDS_HIP_LIB') or os.path.join(_PKG_DIR, 'libpds_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), 'include', 'pds_hip.h')

_lock = threading.Lock()
_lib = None


class ConvBlockParams(ctypes.Structure):
    """struct PdsConvBlockParams"""
    _fields_ = [('weight', ctypes.c_void_p), ('bias', ctypes.c_void_p),
                ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p)]


class MatchingParams(ctypes.Structure):
    """struct PdsMatchingParams"""
    _fields_ = [('features', ctypes.c_int), ('signature_features', ctypes.c_int),
                ('residual_blocks', ctypes.c_int),
                ('first', ConvBlockParams),
                ('blocks', ctypes.POINTER(ConvBlockParams)),
                ('last', ConvBlockParams)]


class EmbeddingParams(ctypes.Structure):
    """struct PdsEmbeddingParams"""
    _fields_ = [('input_features', ctypes.c_int), ('features', ctypes.c_int),
                ('shortcut_features', ctypes.c_int), ('residual_blocks', ctypes.c_int),
                ('downsampling', ConvBlockParams * 2),
                ('blocks', ctypes.POINTER(ConvBlockParams)),
                ('shortcut', ConvBlockParams)]


class RegularizationParams(ctypes.Structure):
    """struct PdsRegularizationParams"""
    _fields_ = [('features', ctypes.c_int),
                ('smoothing', ConvBlockParams),
                ('contraction', (ConvBlockParams * 2) * 4),
                ('expansion', (ConvBlockParams * 2) * 4),
                ('upsample_half', ConvBlockParams),
                ('upsample_full', ConvBlockParams)]


def planned_bytes(nbytes, what):
    """Result of a ``pds_*_workspace_bytes`` planning walk of a training route: zero means the walk FAILED (unsupported
    shape, arena overflow ...), never "no workspace needed" -- surface the library's message instead of running the real
    walk into a 256-byte arena."""
    nbytes = int(nbytes)
    if nbytes <= 0:
        raise RuntimeError('%s: planning failed: %s' % (what, load().pds_last_error().decode(errors='replace')))
    return nbytes


def sources():
    return sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith('.hip'))


BUILD_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def _source_digest():
    """sha256 over the compiler flags and every source / header the library is built from."""
    import hashlib
    h = hashlib.sha256(' '.join(BUILD_FLAGS).encode())
    deps = sources() + sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith('.hpp')) + [HEADER_PATH]
    for path in deps:
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into libpds_hip.so (cross-compiles without a GPU): one object per source,
    compiled in parallel, then one link.  The library is rebuilt when the digest of flags + sources recorded next to
    it differs from the tree's (a shipped .so with a stale or missing stamp is rebuilt, not trusted), or on ``force``."""
    from concurrent.futures import ThreadPoolExecutor
    if os.environ.get('PDS_HIP_LIB'):
        # an alternative build selected for an A/B run is used as it is: rebuilding it from the tree's sources would
        # silently turn the experiment into "tree vs tree"
        return LIB_PATH
    srcs = sources()
    stamp = LIB_PATH + '.sha256'
    digest = _source_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return LIB_PATH
    objdir = os.path.join(os.path.dirname(_PKG_DIR), 'build', 'obj')
    os.makedirs(objdir, exist_ok=True)

    import hashlib
    headers = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith('.hpp')) + [HEADER_PATH]
    base = hashlib.sha256(' '.join(BUILD_FLAGS).encode())
    for path in headers:
        with open(path, 'rb') as f:
            base.update(f.read())

    def compile_one(src):
        # an object is reused when the digest of flags + headers + its own source recorded beside it still matches
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        h = base.copy()
        with open(src, 'rb') as f:
            h.update(f.read())
        mark = obj + '.sha256'
        if not force and os.path.exists(obj) and os.path.exists(mark):
            with open(mark) as f:
                if f.read().strip() == h.hexdigest():
                    return obj
        cmd = ['hipcc'] + BUILD_FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
        with open(mark, 'w') as f:
            f.write(h.hexdigest() + '\n')
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    with open(stamp, 'w') as f:
        f.write(digest + '\n')
    return LIB_PATH


_VP = ctypes.c_void_p
_I = ctypes.c_int
_SZ = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/pds_hip.h declares
SIGNATURES = {
    'pds_abi_version': (_I, []),
    'pds_last_error': (ctypes.c_char_p, []),
    'pds_nonfinite_statistics': (ctypes.c_longlong, [_I]),
    'pds_probe_begin': (_I, [ctypes.c_char_p, _I]),
    'pds_probe_end': (_I, [_VP, _VP, _I]),
    'pds_debug_chain_stamps': (_I, [_VP, _I]),
    'pds_subpixel_map_fwd': (_I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _VP]),
    'pds_shift_concat_fwd': (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP]),
    'pds_matching_workspace_bytes': (_SZ, [ctypes.POINTER(MatchingParams), _I, _I, _I, _I]),
    'pds_matching_fwd': (_I, [ctypes.POINTER(MatchingParams), _VP, _VP, _VP, _I, _I, _I, _I, _I,
                              _VP, _SZ, _I, _VP]),
    'pds_matching_train_workspace_bytes': (_SZ, [ctypes.POINTER(MatchingParams), _I, _I, _I, _I]),
    'pds_matching_train_fwd': (_I, [ctypes.POINTER(MatchingParams), _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _SZ, _VP]),
    'pds_matching_bwd_workspace_bytes': (_SZ, [ctypes.POINTER(MatchingParams), _I, _I, _I, _I]),
    'pds_matching_bwd': (_I, [ctypes.POINTER(MatchingParams), ctypes.POINTER(MatchingParams), _VP, _VP, _VP, _VP, _VP,
                              _I, _I, _I, _I, _I, _VP, _SZ, _VP, _SZ, _VP]),
    'pds_matching_operation_workspace_bytes': (_SZ, [ctypes.POINTER(MatchingParams), _I, _I, _I]),
    'pds_matching_operation_fwd': (_I, [ctypes.POINTER(MatchingParams), _VP, _VP, _I, _I, _I,
                                        _VP, _SZ, _VP]),
    'pds_regularization_workspace_bytes': (_SZ, [ctypes.POINTER(RegularizationParams), _I, _I, _I, _I]),
    'pds_regularization_fwd': (_I, [ctypes.POINTER(RegularizationParams), _VP, _VP, _VP,
                                    _I, _I, _I, _I, _VP, _SZ, _I, _VP]),
    'pds_regularization_subpixel_map_fwd': (_I, [ctypes.POINTER(RegularizationParams), _VP, _VP, _VP,
                                                 _I, _I, _I, _I, _I, _I, _I, _I, _VP, _SZ, _I, _VP]),
    'pds_conv_block_workspace_bytes': (_SZ, [_I] * 9),
    'pds_conv_block_fwd': (_I, [ctypes.POINTER(ConvBlockParams), _VP, _VP, _VP, _VP,
                                _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _SZ, _VP]),
    'pds_conv_block_chained_fwd': (_I, [ctypes.POINTER(ConvBlockParams), _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP,
                                        _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _SZ, _VP]),
    'pds_regularization_bwd_workspace_bytes': (_SZ, [ctypes.POINTER(RegularizationParams), _I, _I, _I, _I]),
    'pds_regularization_bwd': (_I, [ctypes.POINTER(RegularizationParams), ctypes.POINTER(RegularizationParams),
                                    _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _SZ, _VP, _SZ, _VP]),
    'pds_matching_operation_bwd_workspace_bytes': (_SZ, [ctypes.POINTER(MatchingParams), _I, _I, _I]),
    'pds_matching_operation_bwd': (_I, [ctypes.POINTER(MatchingParams), ctypes.POINTER(MatchingParams),
                                        _VP, _VP, _VP, _I, _I, _I, _VP, _SZ, _VP, _SZ, _VP]),
    'pds_contraction_block_bwd_workspace_bytes': (_SZ, [_I, _I, _I, _I, _I]),
    'pds_contraction_block_bwd': (_I, [ctypes.POINTER(ConvBlockParams)] * 4 + [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I,
                                       _VP, _SZ, _VP, _SZ, _VP]),
    'pds_expansion_block_bwd_workspace_bytes': (_SZ, [_I, _I, _I, _I, _I]),
    'pds_expansion_block_bwd': (_I, [ctypes.POINTER(ConvBlockParams)] * 4 + [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I,
                                     _I, _VP, _SZ, _VP, _SZ, _VP]),
    'pds_embedding_workspace_bytes': (_SZ, [ctypes.POINTER(EmbeddingParams), _I, _I, _I, _I, _I]),
    'pds_embedding_fwd': (_I, [ctypes.POINTER(EmbeddingParams), _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _SZ, _I,
                               _VP]),
    'pds_embedding_bwd_workspace_bytes': (_SZ, [ctypes.POINTER(EmbeddingParams), _I, _I, _I, _I, _I]),
    'pds_embedding_bwd': (_I, [ctypes.POINTER(EmbeddingParams)] * 2 + [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I,
                               _VP, _SZ, _VP, _SZ, _VP]),
    'pds_embedding_image_bwd_workspace_bytes': (_SZ, [ctypes.POINTER(EmbeddingParams), _I, _I, _I, _I, _I]),
    'pds_embedding_image_bwd': (_I, [ctypes.POINTER(EmbeddingParams)] * 2 + [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I,
                                     _VP, _SZ, _VP, _SZ, _VP]),
    'pds_disparity_errors_workspace_bytes': (_SZ, [_SZ]),
    'pds_disparity_errors_fwd': (_I, [_VP, _VP, _SZ, ctypes.c_float, _VP, _VP, _VP, _VP, _SZ, _VP]),
    'pds_subpixel_cross_entropy_workspace_bytes': (_SZ, [_I, _I, _I]),
    'pds_subpixel_cross_entropy_fwd': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, ctypes.c_float, _I,
                                            _VP, _SZ, _VP]),
    'pds_subpixel_cross_entropy_bwd': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, ctypes.c_float, _I,
                                            _VP]),
    'pds_subpixel_cross_entropy_weights_bwd': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, ctypes.c_float, _I,
                                                    _VP]),
    'pds_shift_concat_bwd': (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP]),
    'pds_contraction_block_workspace_bytes': (_SZ, [_I, _I, _I, _I, _I]),
    'pds_contraction_block_fwd': (_I, [ctypes.POINTER(ConvBlockParams), ctypes.POINTER(ConvBlockParams),
                                       _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _SZ, _VP]),
    'pds_expansion_block_workspace_bytes': (_SZ, [_I, _I, _I, _I, _I]),
    'pds_expansion_block_fwd': (_I, [ctypes.POINTER(ConvBlockParams), ctypes.POINTER(ConvBlockParams),
                                     _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _SZ, _VP]),
}


ABI_VERSION = 6   # include/pds_hip.h PDS_ABI_VERSION: the argument lists in SIGNATURES are those of this version


def load():
    """Returns the loaded library; raises if it has not been built, or if the file on disk is a build of another ABI
    version (a stale libpds_hip.so would read shifted arguments -- e.g. the ``weights_resident`` flag as the stream)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    'libpds_hip.so is missing (%s). Build it with '
                    '`python -c "import __graft_entry__ as g; g.build()"`; there is no CPU fallback.'
                    % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            lib.pds_abi_version.restype = ctypes.c_int
            lib.pds_abi_version.argtypes = []
            found = lib.pds_abi_version()
            if found != ABI_VERSION:
                raise RuntimeError(
                    '%s implements ABI version %d, this package binds version %d: rebuild it with '
                    '`python -c "import __graft_entry__ as g; g.build()"`' % (LIB_PATH, found, ABI_VERSION))
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, load().pds_last_error().decode()))


def require_gpu_tensor(t, name, dims=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must live on an MI355X (cuda) device: the HIP path has no CPU fallback' % name)
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    if dims is not None and t.dim() != dims:
        raise ValueError('%s must have %d dimensions, got %d' % (name, dims, t.dim()))
    return t.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def stream_handle(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def conv_block_params(conv, norm=None, tensor_of=None):
    """PdsConvBlockParams from a conv module (and its InstanceNorm, if any).  ``tensor_of`` maps each
    parameter to the tensor whose address is stored (identity for the values, the gradient buffers
    for the ``grads`` structs of the backward entry points)."""
    f = tensor_of if tensor_of is not None else (lambda t: t)
    p = ConvBlockParams()
    p.weight = f(conv.weight).data_ptr()
    p.bias = f(conv.bias).data_ptr()
    p.gamma = f(norm.weight).data_ptr() if norm is not None else None
    p.beta = f(norm.bias).data_ptr() if norm is not None else None
    return p


def gradient_buffers(module):
    """One gradient tensor per parameter (written, not accumulated, by the backward kernels) and the
    lookup ``tensor_of`` for conv_block_params."""
    grads = {id(p): torch.empty_like(p) for p in module.parameters()}
    return grads, (lambda p: grads[id(p)])


def parameter_signature(module):
    """Identity and version of every parameter, or None when that cannot be established (inference tensors carry no
    version counter).  Equal signatures mean: same storage, and no in-place update THROUGH THE PARAMETER since (an
    optimizer step or load_state_dict bumps ``_version``; re-assignment changes ``data_ptr``).  Edits through
    ``p.data`` (``p.data.copy_()``, EMA swaps, old-style init) do NOT bump it -- which is why the signature is only
    trusted for modules whose weights the user froze explicitly (``resident_key``)."""
    signature = []
    for p in module.parameters():
        if p.is_inference():
            return None
        signature.append((p.data_ptr(), p._version))
    return tuple(signature)


def resident_key(module, parameter_owner, geometry):
    """Key under which a workspace may keep this module's re-laid-out weights between calls, or None (re-layout on every
    call -- the default).  Residency is opt-in: ``module.freeze_weights()`` promises that the parameters are not
    edited behind autograd's back until ``thaw_weights()`` / ``invalidate_weights()``.  ``train()`` thaws;
    ``.to()`` / ``.cuda()`` / ``load_state_dict`` only INVALIDATE (the next call re-lays out once, the module stays
    frozen and goes on trusting (data_ptr, _version) afterwards) -- FrozenWeightsMixin."""
    if not getattr(module, '_weights_frozen', False):
        return None
    signature = parameter_signature(parameter_owner)
    if signature is None:
        return None
    return (geometry, signature)


class FrozenWeightsMixin(object):
    """Opt-in weight residency of a module that owns a ``Workspace`` (``self._workspace``).

    By default every call re-lays out the weights (a handful of microsecond launches), which is always correct.
    ``freeze_weights()`` lets the inference entry points skip that while the parameters stay untouched: the key still
    carries (data_ptr, _version) of every parameter, so optimizer steps, ``load_state_dict`` and re-assignment are
    seen; writes through ``p.data`` are not, and need ``invalidate_weights()``."""

    _weights_frozen = False

    def freeze_weights(self):
        self._weights_frozen = True
        return self

    def thaw_weights(self):
        self._weights_frozen = False
        self._workspace.invalidate()
        return self

    def invalidate_weights(self):
        """Forget the re-laid-out weights (call after editing parameters through ``.data``); stays frozen."""
        self._workspace.invalidate()
        return self

    # nn.Module entry points that replace or rewrite parameters
    def _apply(self, fn, *args, **kwargs):
        result = super(FrozenWeightsMixin, self)._apply(fn, *args, **kwargs)
        self._workspace.invalidate()
        return result

    def _load_from_state_dict(self, *args, **kwargs):
        # called for every module of the hierarchy a load_state_dict walks (load_state_dict itself only on the root)
        self._workspace.invalidate()
        return super(FrozenWeightsMixin, self)._load_from_state_dict(*args, **kwargs)

    def train(self, mode=True):
        if mode and self._weights_frozen:
            self.thaw_weights()
        return super(FrozenWeightsMixin, self).train(mode)


class Workspace(object):
    """Grow-only device scratch buffers of one module, one per (device, stream): calls on the same stream reuse the
    buffer (stream order makes that safe), calls on different streams -- two pairs in flight -- never share one.

    The buffer also holds the module's re-laid-out weights (the arena of an entry point is deterministic), so it
    can remember what they were made from: ``get_resident(..., key)`` reports ``resident = True`` when the same
    buffer last COMPLETED a call with the same ``key`` (call geometry + parameter signature), which lets the entry point
    skip its weight re-layout launches (``weights_resident`` of include/pds_hip.h).  The key is recorded by
    ``commit(token)`` only after the native call returned success: a call that raised in between (allocation failure,
    non-zero return code) leaves the slot without a key, so the next call re-lays out."""

    MAX_STREAMS = 8   # buffers kept (least recently used first out): streams come and go in a long-lived process

    def __init__(self):
        self._buffers = {}
        self._keys = {}

    def _buffer(self, nbytes, device):
        slot = (device.index, torch.cuda.current_stream(device).cuda_stream)
        buf = self._buffers.pop(slot, None)
        if buf is None or buf.numel() < nbytes:
            buf = None
            self._keys.pop(slot, None)
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        self._buffers[slot] = buf               # most recently used last
        while len(self._buffers) > self.MAX_STREAMS:
            old = next(iter(self._buffers))
            self._buffers.pop(old)
            self._keys.pop(old, None)
        return slot, buf

    def get(self, nbytes, device):
        """Scratch only: whatever weights the buffer held are forgotten."""
        slot, buf = self._buffer(nbytes, device)
        self._keys.pop(slot, None)
        return buf

    def get_resident(self, nbytes, device, key):
        """-> (buffer, resident, token); pass ``token`` to ``commit`` once the call that used the buffer succeeded."""
        slot, buf = self._buffer(nbytes, device)
        previous = self._keys.pop(slot, None)   # recorded again by commit(): a failed call leaves no key behind
        resident = key is not None and previous == key
        return buf, resident, (slot, key, buf.data_ptr())

    def commit(self, token):
        slot, key, address = token
        buf = self._buffers.get(slot)
        if key is not None and buf is not None and buf.data_ptr() == address:
            self._keys[slot] = key

    def invalidate(self):
        self._keys.clear()


def not_differentiable(name):
    raise NotImplementedError(
        '%s: backward of this entry point is not built yet; run it under torch.no_grad()' % name)


def saved_workspace(ctx, name):
    """The forward workspace a backward needs; a second backward through the same node is refused (the workspace is
    released after the first one, and the backward kernels use the forward's intermediates in place)."""
    ws = getattr(ctx, 'forward_workspace', None)
    if ws is None:
        raise RuntimeError('%s: backward through this node a second time is not supported (retain_graph / repeated '
                           'torch.autograd.grad): run the forward again' % name)
    return ws


_warned_eval_with_grad = set()


def warn_eval_with_grad(module):
    """Inference is meant to run under torch.no_grad() (as the reference's trainer.py:232-243 does): with gradients
    enabled the modules take the training route -- every layer output is kept and the fused inference kernels are
    skipped -- whatever module.training says.  Said once per module class."""
    key = type(module).__name__
    if not module.training and key not in _warned_eval_with_grad:
        _warned_eval_with_grad.add(key)
        import warnings
        warnings.warn('%s is in eval() mode but gradients are enabled: taking the (slower, memory-hungry) training '
                      'route; wrap inference in torch.no_grad()' % key, stacklevel=3)

