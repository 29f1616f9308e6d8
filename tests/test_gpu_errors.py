"""GPU (-m gpu): evaluation metrics on the HIP library against the reference's known answers and the golden
fixture (reference practical_deep_stereo/errors.py:9-74, test/test_errors.py:13-66).

Tolerances (stated): pixel-wise maps bit-exact (one fp32 subtraction + abs / one comparison); averages within
1e-6 relative (the library sums in fp64, the reference in fp32)."""
import math

import pytest
import torch

from oracle import pds_oracle as oracle
from tests import helpers
import practicaldeepstereo_nips2018_amd as pds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev(hip_library):
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def test_reference_known_answers(dev):
    g = helpers.golden('g10_errors')
    est, gt = g['ref_est'].to(dev), g['ref_gt'].to(dev)
    pixelwise, mean = pds.errors.compute_absolute_error(est, gt, use_mean=True)
    assert torch.equal(pixelwise.cpu(), torch.tensor([[1.0, 0.0], [0.0, 3.0]]))
    assert math.isclose(mean, 4.0 / 3.0, rel_tol=1e-3)
    pixelwise, median = pds.errors.compute_absolute_error(est, gt, use_mean=False)
    assert math.isclose(median, 1.0, rel_tol=1e-3)
    assert torch.equal(pixelwise.cpu(), torch.tensor([[1.0, 0.0], [0.0, 3.0]]))
    bad, percent = pds.errors.compute_n_pixels_error(est, gt, n=1.0)
    assert torch.equal(bad.cpu(), torch.tensor([[0.0, 0.0], [0.0, 1.0]]))
    assert math.isclose(percent, 100.0 / 3.0, rel_tol=1e-3)
    nothing = torch.full((2, 2), float('inf'), device=dev)
    assert pds.errors.compute_absolute_error(est, nothing)[1] == 0.0
    assert pds.errors.compute_n_pixels_error(est, nothing, n=1.0)[1] == 0.0


def test_random_case_with_unknown_band(dev):
    g = helpers.golden('g10_errors')
    est, gt = g['random_est'].to(dev), g['random_gt'].to(dev)
    pixelwise, mean = pds.errors.compute_absolute_error(est, gt)
    assert torch.equal(pixelwise.cpu(), g['random_pixelwise'])
    assert math.isclose(mean, g['random_mean'].item(), rel_tol=1e-6)
    assert math.isclose(pds.errors.compute_absolute_error(est, gt, use_mean=False)[1], g['random_median'].item(),
                        rel_tol=1e-6)
    bad, percent = pds.errors.compute_n_pixels_error(est, gt)
    assert torch.equal(bad.cpu(), g['random_bad'])
    assert math.isclose(percent, g['random_percent'].item(), rel_tol=1e-6)
    bad2, both = pds.errors.compute_errors(est, gt)
    assert torch.equal(bad2, bad)
    assert math.isclose(both['three_pixels_error'], percent) and math.isclose(both['mean_absolute_error'], mean)


def test_full_size_map_against_oracle(dev):
    """960x540 disparity map (config 2 output size), ragged against the 1024-element blocks."""
    g = torch.Generator().manual_seed(41)
    est = torch.rand(1, 540, 960, generator=g) * 190
    gt = est + torch.randn(1, 540, 960, generator=g) * 3
    gt[:, :36, :] = float('inf')
    pixelwise, mean = pds.errors.compute_absolute_error(est.to(dev), gt.to(dev))
    bad, percent = pds.errors.compute_n_pixels_error(est.to(dev), gt.to(dev))
    pixelwise_o, mean_o = oracle.absolute_error(est, gt)
    bad_o, percent_o = oracle.n_pixels_error(est, gt)
    assert torch.equal(pixelwise.cpu(), pixelwise_o) and torch.equal(bad.cpu(), bad_o)
    assert math.isclose(mean, mean_o, rel_tol=1e-5) and math.isclose(percent, percent_o, rel_tol=1e-5)


def test_argument_checks(dev):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pds.errors.compute_absolute_error(torch.zeros(2, 2), torch.zeros(2, 2))
    with pytest.raises(ValueError):
        pds.errors.compute_n_pixels_error(torch.zeros(2, 2, device=dev), torch.zeros(2, 3, device=dev))
