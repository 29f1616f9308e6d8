"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares;
host-side mirrors keep the reference's API surface (names, arguments, errors, state-dict keys)."""
import ctypes
import os
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
    assert hip_library.pds_abi_version() == 6


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
    # exactly the reference's 122 tensors, in its order: the list is pinned by the reference's own training step (G11)
    assert len(keys) == 122
    import numpy as np
    with np.load(os.path.join(helpers.GOLDEN, 'g11_training_step.npz')) as z:
        assert keys == [str(k) for k in z['parameter_names']]
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


def test_stale_library_of_another_abi_version_is_refused(hip_library, monkeypatch):
    # load() itself must check pds_abi_version() (a stale git-ignored .so would read shifted arguments)
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match='ABI version'):
        _lib.load()
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION - 1)
    assert _lib.load().pds_abi_version() == _lib.ABI_VERSION


def test_weight_residency_is_opt_in_and_guarded():
    # not frozen: no key, i.e. the weights are re-laid out on every call (p.data edits are then always seen)
    net = pds.PdsNetwork.default(63)
    geometry = (1, 8, 16, 16)
    assert _lib.resident_key(net._matching, net._matching._operation, geometry) is None
    net.freeze_weights()
    assert net._matching._weights_frozen and net._regularization._weights_frozen and net._embedding._weights_frozen
    key = _lib.resident_key(net._matching, net._matching._operation, geometry)
    assert key is not None and key == _lib.resident_key(net._matching, net._matching._operation, geometry)
    with torch.no_grad():
        next(net._matching.parameters()).mul_(2.0)          # bumps _version: a different key
    assert _lib.resident_key(net._matching, net._matching._operation, geometry) != key
    assert _lib.resident_key(net._matching, net._matching._operation, (2, 8, 16, 16)) != key
    # inference tensors carry no version counter: never resident (and no crash)
    with torch.inference_mode():
        frozen = pds.MatchingOperation()
    holder = pds.Matching(3, frozen).freeze_weights()
    assert _lib.parameter_signature(frozen) is None
    assert _lib.resident_key(holder, frozen, geometry) is None
    # train() thaws; load_state_dict / _apply forget what the workspaces held
    net.train()
    assert not net._matching._weights_frozen and not net._regularization._weights_frozen
    net.eval().freeze_weights()
    net._regularization._workspace._keys[(0, 0)] = 'stale'
    net._matching._workspace._keys[(0, 0)] = 'stale'
    net.load_state_dict(net.state_dict())
    assert not net._regularization._workspace._keys and not net._matching._workspace._keys
    net._embedding._workspace._keys[(0, 0)] = 'stale'
    net.double().float()
    assert not net._embedding._workspace._keys


def test_workspace_commits_its_key_only_after_success(monkeypatch):
    ws = _lib.Workspace()
    buffers = {}

    def fake_buffer(nbytes, device):
        slot = (0, 7)
        buf = buffers.get(slot)
        if buf is None or buf.numel() < nbytes:
            ws._keys.pop(slot, None)
            buf = buffers[slot] = torch.empty(int(nbytes), dtype=torch.uint8)
        ws._buffers[slot] = buf
        return slot, buf
    monkeypatch.setattr(ws, '_buffer', fake_buffer)
    buf, resident, token = ws.get_resident(1024, None, 'k1')
    assert not resident and not ws._keys            # nothing recorded before commit
    buf, resident, token = ws.get_resident(1024, None, 'k1')
    assert not resident                              # the first call never committed (it "failed")
    ws.commit(token)
    buf, resident, token = ws.get_resident(1024, None, 'k1')
    assert resident and not ws._keys                 # resident now; the key is out until THIS call commits
    ws.commit(token)
    assert ws.get_resident(1024, None, 'k2')[1] is False   # other key
    buf, resident, token = ws.get_resident(1024, None, 'k1')
    assert not resident                              # ... and the k2 call never committed
    ws.commit(token)
    buf, resident, token = ws.get_resident(4096, None, 'k1')
    assert not resident                              # a grown buffer holds nothing
    ws.commit(token)
    ws.get(1024, None)                               # scratch use forgets the weights
    assert ws.get_resident(1024, None, 'k1')[1] is False
    assert ws.get_resident(1024, None, None)[1] is False   # no key: never resident


def test_bench_host_helpers(tmp_path, monkeypatch):
    """bench.py's host-side helpers that need no GPU: the kernel-switch gate (PDS_X3 must be ignored without
    PDS_DEBUG_SWITCHES=1, as the library ignores it) and the shader-clock sampler's sysfs parser."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    monkeypatch.setenv('PDS_X3', '0')
    monkeypatch.delenv('PDS_DEBUG_SWITCHES', raising=False)
    sys.modules.pop('bench', None)
    bench = importlib.import_module('bench')
    assert bench.X3 is True and bench.CONV64_EXECUTED_PEAK == bench.BF16_MFMA_PEAK_TFLOPS   # the variable is ignored
    monkeypatch.setenv('PDS_DEBUG_SWITCHES', '1')
    sys.modules.pop('bench', None)
    bench = importlib.import_module('bench')
    assert bench.X3 is False and bench.CONV64_EXECUTED_PEAK == bench.FP32_MFMA_PEAK_TFLOPS     # armed: honoured
    sys.modules.pop('bench', None)
    # pp_dpm_sclk: the level marked '*' is the current clock; several cards -> the busiest (highest) one
    a, b = tmp_path / 'a', tmp_path / 'b'
    a.write_text('0: 132Mhz\n1: 2100Mhz *\n')
    b.write_text('0: 132Mhz *\n1: 2400Mhz\n')
    sampler = bench.ClockSampler()
    sampler._files = [str(a), str(b)]
    assert sampler._read_sysfs() == 2100.0
    sampler._files = [str(tmp_path / 'missing')]
    assert sampler._read_sysfs() is None
    sampler.samples = [2100.0, 2400.0, 2000.0]
    sampler.source = 'test'
    summary = sampler.summary()
    assert summary['sclk_mhz_min'] == 2000.0 and summary['sclk_mhz_median'] == 2100.0 and summary['sclk_mhz_max'] == 2400.0
