"""GPU (-m gpu): the time-per-image harness follows the reference's test loop (trainer.py:141-148, 229-252;
pds_trainer.py:48-64): host examples in, whole network inside the synchronize/time bracket, metrics on the device."""
import pytest
import torch

import practicaldeepstereo_nips2018_amd as pds
from practicaldeepstereo_nips2018_amd import errors, timing
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev(hip_library):
    return torch.device('cuda:0')


def test_time_per_image_protocol(dev):
    net = helpers.seeded(lambda: pds.PdsNetwork.default(63)).to(dev)
    g = torch.Generator().manual_seed(4)
    examples = [{'left': torch.rand(1, 3, 100, 156, generator=g) * 255,
                 'right': torch.rand(1, 3, 100, 156, generator=g) * 255,
                 'disparity': torch.rand(1, 100, 156, generator=g) * 60} for _ in range(5)]
    examples[0]['disparity'][:, :10] = float('inf')   # pixels without ground truth are skipped by the metrics
    result = timing.time_per_image(net, examples, dev, warmup=2)
    assert result['examples'] == 3                      # the warm-up examples are run but not timed
    assert 0.0 < result['time_per_image_ms'] <= result['time_per_image_with_host_copy_ms']
    assert 'trainer.py:141-148' in result['protocol']
    assert not net.training                             # the test loop runs in eval mode (trainer.py:231)

    # the reported metrics are the reference's (pds_trainer.py:48-58) averaged over ALL examples
    maes, bad = [], []
    with torch.no_grad():
        for e in examples:
            out = net(e['left'].to(dev), e['right'].to(dev))
            assert out.shape == (1, 100, 156) and out.is_contiguous()   # cropped back to the input size
            maes.append(float(errors.compute_absolute_error(out, e['disparity'].to(dev))[1]))
            bad.append(float(errors.compute_n_pixels_error(out, e['disparity'].to(dev))[1]))
    assert result['mean_absolute_error'] == pytest.approx(sum(maes) / 5, rel=1e-6)
    assert result['three_pixels_error'] == pytest.approx(sum(bad) / 5, rel=1e-6)


def test_run_network_and_measure_time_brackets_the_call(dev):
    calls = []

    def network(left, right):
        calls.append((left.device.type, right.device.type))
        return left.sum() + right.sum()

    left = torch.ones(4, device=dev)
    out, seconds = timing.run_network_and_measure_time(network, left, left)
    assert calls == [('cuda', 'cuda')] and float(out) == 8.0 and seconds >= 0.0
