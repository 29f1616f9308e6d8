"""GPU (-m gpu): the persistent chain kernel of the inner hourglass levels (csrc/conv3d_ks.hip: conv3d_ks_chain_kernel,
regularization.py:22-26, 48-52) -- eleven layers + their InstanceNorm folds as ONE launch, workgroups drawing tickets and
handing results to each other inside the launch -- gives the bits of the per-launch path, also under uneven load.

Opt-in (measured slower than the launches it replaces: docs/LAB_NOTES.md, round 6), so the tests select it through
PDS_DEBUG_SWITCHES=1 PDS_CONV3D_KS_CHAIN=1 in child processes (tools/chain_check.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, 'tools', 'chain_check.py')


def test_chain_kernel_is_bit_identical_to_the_per_launch_path(hip_library):
    """config-1, config-2, config-4 (batch 2) and an odd small shape: cost volume and fused disparity, sha256 of the bytes."""
    out = subprocess.run([sys.executable, TOOL], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode(errors='replace')
    assert out.returncode == 0 and 'IDENTICAL' in text and 'DIFFERENT' not in text, text[-2000:]


def test_chain_kernel_under_stream_skew_and_competing_kernels(hip_library):
    """300 passes at config 2 dealt to three HIP streams with random delays and a competing GEMM stream (two chain kernels
    may be partly resident at once: the ticket scheme needs no co-residency): every result equals the first, no poll timed
    out.  (tools/chain_check.py stress 1000 is the long form.)"""
    out = subprocess.run([sys.executable, TOOL, 'stress', '300'], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode(errors='replace')
    assert out.returncode == 0 and 'mismatches 0 nonfinite/timeouts 0' in text, text[-2000:]
