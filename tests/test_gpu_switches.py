"""GPU (-m gpu): the alternative kernel paths behind the runtime switches of DESIGN.md 8.1 stay correct.

The switches are read once per process (static initialisers in libpds_hip.so), so every configuration runs the
single-layer suite (tests/test_gpu_conv_block.py) and the small fused-Matching parity cases in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {'PDS_X3_FP16': '0'},                    # conv2d_x3 on its range-safe form everywhere (three bf16 parts, six products)
    {'PDS_X3': '0'},                         # exact-fp32 MFMA kernels (Winograd domain) instead of the split-operand kernel
    {'PDS_X3': '0', 'PDS_WINOGRAD': '0'},    # ... and the direct exact-fp32 MFMA kernel
    {'PDS_X3': '0', 'PDS_WINO_TILE16': '0'},   # ... with wide (4 x 64) Winograd tiles everywhere
    {'PDS_MATCHING_FUSED': '0'},    # Matching without the factorisation glue
    {'PDS_MATCHING_COLUMNS': '0'},  # whole-plane form of the layer-0 / layer-1 factorisation (3 + 5 planes)
    {'PDS_CONV3D_XCD_MAP': '0'},
    {'PDS_CONV3D_T8': '0', 'PDS_DECONV_CELL': '0', 'PDS_CONV3D_KS': '0'},   # generic MFMA kernels for all hourglass layers
    {'PDS_CONV3D_T8X': '0'},        # exact-fp32 kernel for the 8 -> 8 full-resolution layers (round 3)
    {'PDS_CONV3D_T8X': '2'},        # ... the split-operand kernel also for the un-certified first layer (bf16 x 3 form)
    {'PDS_DECONV_CELL_X': '0'},     # exact-fp32 MFMAs in the dense-cell transposed convolutions
    {'PDS_CONV2D_T8': '0'},         # generic kernel for the 64 -> 8 signature convolution
    {'PDS_CONV2D_T8W': '0'},        # ... its 16 x 32-tile form instead of the full-width one
    {'PDS_CONV2D_T8W_ROWS': '8'},   # ... 8-row tiles (two staging items per thread) where round 6 takes 6-row tiles
    {'PDS_WINO_ROWS6': '0'},        # 4-row tiles in every Winograd launch (before the 6-row form of the small launches)
    {'PDS_CONV3D_KSX': '0'},        # exact-fp32 MFMAs in every K-split layer of the deep hourglass levels (round 5)
    {'PDS_CONV3D_KS_LIMIT': '1000'},   # the K-split kernel only for the two deepest levels of the hourglass
    {'PDS_CONV3D_NX': '0'},         # exact-fp32 kernel for the 16-channel quarter-resolution layers (before round 6)
]


@pytest.mark.parametrize('switch', SWITCHES, ids=lambda s: '_'.join('%s=%s' % kv for kv in s.items()))
def test_alternative_paths(hip_library, switch):
    env = dict(os.environ)
    env.update(switch)
    env['PDS_DEBUG_SWITCHES'] = '1'   # the library ignores kernel-selection variables without it (csrc/common.hpp)
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
           'tests/test_gpu_conv_block.py',
           'tests/test_gpu_parity.py::test_subpixel_map_random_vs_oracle',
           'tests/test_gpu_parity.py::test_fused_matching_shapes_vs_oracle',
           'tests/test_gpu_parity.py::test_fused_matching_golden',
           'tests/test_gpu_parity.py::test_config1_hot_path_vs_golden',
           'tests/test_gpu_parity.py::test_regularization_golden']
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = out.stdout.decode(errors='replace')[-2000:]
    assert out.returncode == 0, tail


# the kernel choices of the backward pass: every alternative must reproduce the oracle's gradients
BACKWARD_SWITCHES = [
    {'PDS_WGRAD2D_X3': '0'},          # 2-D weight gradients on the exact-fp32 MFMA kernel
    {'PDS_BWD_DATA_V2': '0'},         # stride-2 data gradients on the one-position-per-thread kernels of round 2
    {'PDS_IN_BWD_PLANE': '0'},        # InstanceNorm backward of per-plane groups on the two-pass kernels
    {'PDS_WGRAD3D_S2_ROLLING': '2'},  # the rolling stride-2 weight-gradient kernel also for small layers
    {'PDS_WGRAD3D_S2_ROLLING': '0'},  # ... and never
    {'PDS_WGRAD3D_MFMA': '0', 'PDS_WGRAD3D_S2_MFMA': '0'},   # 3-D weight gradients on the VALU kernels
]


@pytest.mark.parametrize('switch', BACKWARD_SWITCHES, ids=lambda s: '_'.join('%s=%s' % kv for kv in s.items()))
def test_alternative_backward_paths(hip_library, switch):
    env = dict(os.environ)
    env.update(switch)
    env['PDS_DEBUG_SWITCHES'] = '1'
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
           'tests/test_gpu_backward.py::test_matching_operation_backward',
           'tests/test_gpu_backward.py::test_matching_training_route_backward',
           'tests/test_gpu_backward.py::test_regularization_backward',
           'tests/test_gpu_backward.py::test_standalone_blocks_backward',
           'tests/test_gpu_backward.py::test_standalone_blocks_backward_eight_features_odd_sizes']
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = out.stdout.decode(errors='replace')[-2000:]
    assert out.returncode == 0, tail


def test_switches_are_ignored_without_the_debug_gate(hip_library):
    """A stray kernel-selection variable must not move a production process off the default paths: without
    PDS_DEBUG_SWITCHES=1 the library does not read them.  PDS_X3=0 would send the 64-channel layers to the exact-fp32
    Winograd kernel, which the launch probe would show; here the probe must still see conv2d_x3 launches."""
    code = (
        "import ctypes, torch, practicaldeepstereo_nips2018_amd as pds\n"
        "from practicaldeepstereo_nips2018_amd import _lib\n"
        "lib = _lib.load(); dev = torch.device('cuda:0'); torch.manual_seed(0)\n"
        "m = pds.Matching(7, pds.MatchingOperation()).to(dev).eval()\n"
        "l, r = torch.randn(1, 64, 32, 64, device=dev), torch.randn(1, 64, 32, 64, device=dev)\n"
        "_lib.check(lib.pds_probe_begin(b'conv2d_x3', 64), 'pds_probe_begin')\n"
        "with torch.no_grad(): m(l, r)\n"
        "torch.cuda.synchronize()\n"
        "print('PROBE', lib.pds_probe_end(None, None, 64))\n")
    for gate, expect_x3 in ((None, True), ('1', False)):
        env = dict(os.environ)
        env['PDS_X3'] = '0'
        env.pop('PDS_DEBUG_SWITCHES', None)
        if gate:
            env['PDS_DEBUG_SWITCHES'] = gate
        out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=600)
        text = out.stdout.decode(errors='replace')
        assert out.returncode == 0, text[-2000:]
        launches = int([t for t in text.splitlines() if t.startswith('PROBE')][-1].split()[1])
        assert (launches > 0) == expect_x3, (gate, launches)


def test_channel_blocked_levels_are_bit_identical(hip_library):
    """PDS_MATCHING_CB8 = 0 (planar), 1 (default: blocked tensor inside a residual block), 2 (+ the first 64 -> 64 launch fed
    from the blocked layer-1 planes): the same products in the same order, so the signatures are equal bit for bit at
    config-2, config-1 (batch 2) and config-4 plane shapes (tools/cb8_check.py spawns one process per level)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'cb8_check.py'), '0', '1', '2'], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode(errors='replace')
    assert out.returncode == 0 and 'IDENTICAL' in text, text[-2000:]
