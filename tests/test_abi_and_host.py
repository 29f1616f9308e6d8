"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares;
host-side mirrors keep the reference's API surface (names, arguments, errors, state-dict keys)."""
import ctypes
import re

import pytest
import torch

from tests import helpers
import practicaldeepstereo_nips2018_amd as pds
from practicaldeepstereo_nips2018_amd import _lib, size_adapter


def header_symbols():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pds_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol(hip_library):
    names = header_symbols()
    assert len(names) >= 15
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), 'libpds_hip.so does not export %s' % name
        assert name in _lib.SIGNATURES, 'python binding lacks %s' % name
    assert sorted(_lib.SIGNATURES) == names
    assert hip_library.pds_abi_version() == 2


def test_argument_validation_needs_no_gpu(hip_library):
    rc = hip_library.pds_subpixel_map_fwd(None, None, 1, 5, 1, 1, 2, 1, None)
    assert rc != 0 and b'null' in hip_library.pds_last_error()
    params = pds.Regularization().native_params()
    assert hip_library.pds_regularization_workspace_bytes(ctypes.byref(params), 1, 12, 16, 32) == 0
    assert b'multiples of 16' in hip_library.pds_last_error()
    assert hip_library.pds_regularization_workspace_bytes(ctypes.byref(params), 1, 16, 16, 16) == 0
    assert hip_library.pds_regularization_workspace_bytes(ctypes.byref(params), 1, 16, 16, 32) > 0
    op = pds.MatchingOperation()
    mp, keep = op.native_params()
    assert hip_library.pds_matching_workspace_bytes(ctypes.byref(mp), 1, 16, 32, 16) > 0


def test_cpu_tensors_are_refused_loudly():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pds.SubpixelMap()(torch.zeros(1, 5, 2, 2))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pds.Matching(3, pds.MatchingOperation())(torch.zeros(1, 64, 4, 4), torch.zeros(1, 64, 4, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pds.Regularization()(torch.zeros(1, 8, 16, 16, 32), torch.zeros(1, 8, 16, 32))


@pytest.mark.parametrize('bad', [dict(half_support_window=4, disparity_step=0),
                                 dict(half_support_window=0, disparity_step=2),
                                 dict(half_support_window=3, disparity_step=2)])
def test_subpixel_map_value_errors(bad):
    # reference estimator.py:34-41
    with pytest.raises(ValueError):
        pds.SubpixelMap(**bad)


def test_network_maximum_disparity_rule():
    net = pds.PdsNetwork.default(63)
    with pytest.raises(ValueError):
        net.set_maximum_disparity(100)  # network.py:28-31
    net.set_maximum_disparity(191)
    assert net._matching._maximum_disparity == 47


def test_state_dict_keys_match_reference_layout():
    net = helpers.seeded(lambda: pds.PdsNetwork.default(191))
    keys = list(net.state_dict().keys())
    assert len(keys) == 122 + 0 or len(keys) > 100
    for k in ['_matching._operation._matching_operation_modules.0.weight',
              '_matching._operation._matching_operation_modules.1.convolutions.0.0.weight',
              '_matching._operation._matching_operation_modules.2.convolutions.1.2.bias',
              '_matching._operation._matching_operation_modules.3.bias',
              '_regularization._smoothing.0.weight', '_regularization._smoothing.2.weight',
              '_regularization._contraction_blocks.3._downsampling_2x.0.weight',
              '_regularization._expansion_blocks.0._upsampling_2x.0.weight',
              '_regularization._upsample_to_halfsize.2.bias',
              '_regularization._upsample_to_fullsize.weight',
              '_embedding._embedding_modules.1.0.weight', '_embedding._shortcut.2.weight']:
        assert k in keys, k
    # seed 0 + default(191) -> 2 217 717 parameters, fp64 sum 1398.3613765379
    assert sum(v.numel() for v in net.state_dict().values()) == 2217717
    assert abs(helpers.checksum(net.state_dict()) - 1398.3613765379) < 1e-6
    assert net._regularization._expansion_blocks[0]._upsampling_2x[0].weight.shape == (128, 64, 4, 4, 4)
    assert net._regularization._upsample_to_fullsize.weight.shape == (4, 1, 3, 4, 4)


def test_size_adapter_round_trip():
    # reference test/test_size_adapter.py: pad (1,10,63,100) -> (1,10,64,128), unpad identity
    adapter = size_adapter.SizeAdapter()
    x = torch.rand(1, 10, 63, 100)
    padded = adapter.pad(x)
    assert padded.shape == (1, 10, 64, 128)
    assert torch.equal(padded[..., :1, :], torch.zeros(1, 10, 1, 128))
    assert torch.equal(padded[..., :, :28], torch.zeros(1, 10, 64, 28))
    assert torch.equal(adapter.unpad(padded), x)
    assert adapter.pad(torch.rand(1, 3, 64, 128)).shape == (1, 3, 64, 128)


def test_embedding_has_no_cpu_fallback():
    # the descriptor network runs on the HIP library only (shapes of reference test/test_embedding.py are
    # checked on the GPU in tests/test_gpu_embedding.py); here: same state-dict keys, loud failure on CPU
    from practicaldeepstereo_nips2018_amd.embedding import Embedding
    emb = helpers.seeded(Embedding)
    keys = set(emb.state_dict().keys())
    assert '_embedding_modules.1.0.weight' in keys and '_shortcut.2.bias' in keys
    assert emb.state_dict()['_embedding_modules.1.0.weight'].shape == (64, 3, 5, 5)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        emb(torch.rand(2, 3, 100, 100))


def test_backward_workspace_planning_succeeds():
    # the planning walks run on the host: every *_bwd_workspace_bytes must cover at least the gradient of its
    # largest activation (a failed planning walk used to return a few hundred bytes)
    import ctypes
    from practicaldeepstereo_nips2018_amd.embedding import Embedding
    lib = _lib.load()
    params, keep = Embedding().native_params()
    fwd = lib.pds_embedding_workspace_bytes(ctypes.byref(params), 2, 40, 56, 0, 0)
    bwd = lib.pds_embedding_bwd_workspace_bytes(ctypes.byref(params), 2, 40, 56, 0, 0)
    assert fwd > 2 * 64 * 20 * 28 * 4 and bwd > 2 * 64 * 20 * 28 * 4
    assert lib.pds_contraction_block_bwd_workspace_bytes(1, 8, 16, 16, 16) > 16 * 8 * 8 * 8 * 4
    assert lib.pds_expansion_block_bwd_workspace_bytes(1, 16, 8, 8, 8) > 8 * 16 * 16 * 16 * 4
    del keep
