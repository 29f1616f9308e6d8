"""CPU: the oracle restatement against the committed golden vectors (generated from the reference
by tests/golden/make_golden.py) and against the reference's own known-answer tests."""
import numpy as np
import pytest
import torch

from oracle import pds_oracle as oracle
from tests import helpers
import practicaldeepstereo_nips2018_amd as pds


def mock_operation(x):
    return torch.max(x, dim=1, keepdim=True)[0]


def test_matching_shift_semantics_known_answer():
    # reference test/test_matching.py:17-32
    left = torch.tensor([0., 2., 1., 2.]).view(1, 1, 1, 4)
    right = torch.tensor([3., 4., 2., 4.]).view(1, 1, 1, 4)
    out = oracle.matching(left, right, 2, mock_operation)
    assert np.array_equal(out.numpy().reshape(3, 4), [[3, 4, 2, 4], [0, 3, 4, 2], [0, 2, 3, 4]])
    out = oracle.matching(left, right, 1, mock_operation)
    assert np.array_equal(out.numpy().reshape(2, 4), [[3, 4, 2, 4], [0, 3, 4, 2]])
    g = helpers.golden('g1_matching_mock')
    assert torch.equal(oracle.matching(g['left2'], g['right2'], 5, mock_operation), g['out2_max5'])


def test_subpixel_map_known_answers():
    # reference test/test_estimator.py:14-27
    sim = torch.tensor([0.1, 0.4, 0.3, 0.2, 0.3]).view(1, 5, 1, 1)
    assert abs(oracle.subpixel_map(sim, 2, 1).item() - 1.52) < 1e-4
    assert abs(oracle.subpixel_map(sim, 2, 2).item() - 2.124) < 1e-4


def test_subpixel_map_golden_cases():
    g = helpers.golden('g5_subpixel_map')
    names = sorted(k[:-3] for k in g if k.endswith('_in'))
    assert len(names) >= 10
    for name in names:
        hw, step = [int(v) for v in g[name + '_cfg']]
        out = oracle.subpixel_map(g[name + '_in'], hw, step)
        assert helpers.maxdiff(out, g[name + '_out']) <= 1e-6, name
    # probes of SURVEY.md 7.3
    assert abs(g['tie_first_out'].item() - oracle.subpixel_map(g['tie_first_in']).item()) < 1e-6
    assert abs(g['all_equal_out'].item() - 2.0) < 1e-6
    assert abs(g['edge_low_out'].item() - 1.2053844) < 1e-5
    assert abs(g['edge_high_out'].item() - 9.6050835) < 1e-5


@pytest.mark.parametrize('bad', [(4, 0), (0, 2), (3, 2)])
def test_subpixel_map_value_errors(bad):
    with pytest.raises(ValueError):
        oracle.check_subpixel_map_arguments(*bad)


def test_matching_operation_golden():
    g = helpers.golden('g2_matching')
    op = helpers.seeded(pds.MatchingOperation)
    assert abs(helpers.checksum(op.state_dict()) - g['weight_checksum'].item()) < 1e-9
    p = helpers.prefixed(op.state_dict(), '_m._operation')
    out = oracle.matching_with_operation(p, '_m', g['left'], g['right'], 15)
    assert out.shape == (1, 8, 16, 16, 32)
    assert helpers.maxdiff(out, g['signatures']) <= 1e-6
    assert helpers.maxdiff(oracle.matching_operation(p, '_m._operation', g['concatenated']),
                           g['operation_out']) <= 1e-6


def test_regularization_golden():
    g = helpers.golden('g3_regularization')
    reg = helpers.seeded(pds.Regularization)
    assert abs(helpers.checksum(reg.state_dict()) - g['weight_checksum'].item()) < 1e-9
    p = helpers.prefixed(reg.state_dict(), '_r')
    out = oracle.regularization(p, '_r', g['signatures'], g['shortcut'])
    assert out.shape == (1, 32, 64, 128)
    assert helpers.maxdiff(out, g['cost']) <= 1e-6


def test_blocks_golden():
    g = helpers.golden('g4_blocks')
    con = helpers.seeded(lambda: (torch.rand(2, 6, 10, 14, 16), pds.ContractionBlock3d(6))[1])
    assert abs(helpers.checksum(con.state_dict()) - g['contraction_checksum'].item()) < 1e-9
    down, smooth = oracle.contraction_block_3d(helpers.prefixed(con.state_dict(), '_c'), '_c',
                                               g['contraction_in'])
    assert down.shape == (2, 12, 5, 7, 8) and smooth.shape == (2, 12, 5, 7, 8)
    assert helpers.maxdiff(down, g['contraction_down']) <= 1e-6
    assert helpers.maxdiff(smooth, g['contraction_smooth']) <= 1e-6

    def make_expansion():
        torch.rand(2, 6, 10, 14, 16)
        torch.rand(2, 3, 20, 28, 32)
        return pds.ExpansionBlock3d(6)
    exp = helpers.seeded(make_expansion)
    assert abs(helpers.checksum(exp.state_dict()) - g['expansion_checksum'].item()) < 1e-9
    out = oracle.expansion_block_3d(helpers.prefixed(exp.state_dict(), '_e'), '_e',
                                    g['expansion_in'], g['expansion_shortcut'])
    assert out.shape == (2, 3, 20, 28, 32)
    assert helpers.maxdiff(out, g['expansion_out']) <= 1e-6


def test_config1_full_network_golden():
    """Config 1 (BASELINE.json configs[0]): 128x256, D=64, the whole pipeline on CPU with this
    oracle's embedding feeding the oracle hot path."""
    g = helpers.golden('g6_config1')
    net = helpers.seeded(lambda: pds.PdsNetwork.default(63)).eval()
    assert abs(helpers.checksum(net.state_dict()) - g['weight_checksum'].item()) < 1e-9
    left, right = helpers.images(1, 128, 256)
    with torch.no_grad():
        p = {k: v for k, v in net.state_dict().items()}
        ld, shortcut = oracle.embedding(p, '_embedding', oracle.pad_to_multiple(left)[0])
        rd = oracle.embedding(p, '_embedding', oracle.pad_to_multiple(right)[0])[0]
        ms, cost, disparity = oracle.hot_path(p, ld, rd, shortcut, 63, return_stages=True)
    assert helpers.maxdiff(ms[:, :, ::2, ::4, ::4], g['signatures_sub']) <= 2e-5
    assert helpers.maxdiff(cost[:, ::4, ::8, ::8], g['cost_sub']) <= 1e-4
    rep = helpers.disparity_report(disparity, g['disparity'])
    assert rep['mae'] <= 1e-3, rep


def test_embedding_golden():
    """Embedding after SizeAdapter.pad on a 100x150 image (28 rows / 42 columns of padding) and a bare odd-sized
    call (reference embedding.py:11-65, size_adapter.py:29-43)."""
    g = helpers.golden('g9_embedding')
    emb = helpers.seeded(pds.Embedding)
    assert abs(helpers.checksum(emb.state_dict()) - g['weight_checksum'].item()) < 1e-9
    p = helpers.prefixed(emb.state_dict(), '_embedding')
    with torch.no_grad():
        padded, rows, columns = oracle.pad_to_multiple(g['image'])
        assert (rows, columns) == (28, 42) and padded.shape[-2:] == (128, 192)
        assert float(padded[..., :28, :].abs().max()) == 0.0 and float(padded[..., :, :42].abs().max()) == 0.0
        descriptor, shortcut = oracle.embedding(p, '_embedding', padded)
        odd_descriptor, odd_shortcut = oracle.embedding(p, '_embedding', g['odd_image'])
    assert descriptor.shape == (2, 64, 32, 48) and shortcut.shape == (2, 8, 32, 48)
    assert odd_descriptor.shape == (1, 64, 10, 13)
    assert helpers.maxdiff(descriptor, g['descriptor']) <= 1e-5
    assert helpers.maxdiff(shortcut, g['shortcut']) <= 1e-5
    assert helpers.maxdiff(odd_descriptor, g['odd_descriptor']) <= 1e-5
    assert helpers.maxdiff(odd_shortcut, g['odd_shortcut']) <= 1e-5


def test_subpixel_cross_entropy_golden():
    """reference test/test_loss.py:12-37 (1.3654 + gradient table) and a seeded case with an inf band."""
    g = helpers.golden('g8_loss')
    sim = g['ref_sim'].clone().requires_grad_(True)
    value = oracle.subpixel_cross_entropy(sim, g['ref_gt'], g['ref_weights'], diversity=2.0, disparity_step=1)
    value.backward()
    assert abs(value.item() - 1.3654) < 1e-3
    expected = torch.tensor([[0.0262, -0.0567, -0.0219, 0.0524], [0.0, 0.0, 0.0, 0.0],
                             [0.0011, -0.0002, -0.0007, -0.0002]]).t().reshape(1, 4, 3, 1)
    assert torch.allclose(sim.grad, expected, atol=1e-3)
    assert helpers.maxdiff(sim.grad, g['ref_grad']) <= 1e-7
    for name, weights in (('plain', None), ('weighted', g['random_weights'].clone().requires_grad_(True))):
        s2 = g['random_sim'].clone().requires_grad_(True)
        v = oracle.subpixel_cross_entropy(s2, g['random_gt'], weights)
        v.backward()
        assert abs(v.item() - g['random_%s_value' % name].item()) < 1e-5
        assert helpers.maxdiff(s2.grad, g['random_%s_grad' % name]) <= 1e-7
        if weights is not None:   # loss.py:73-77: the weights receive a gradient from the reference's autograd
            assert helpers.maxdiff(weights.grad, g['random_weighted_weights_grad']) <= 1e-7


def test_image_gradient_golden():
    """embedding.py:32,46-65 behind size_adapter.py:29-43 under autograd: the oracle's fp32 image gradient against the
    reference's fp64 one (G12), no further away than the reference's own fp32 run."""
    g = helpers.golden('g12_image_gradient')
    torch.manual_seed(0)
    import practicaldeepstereo_nips2018_amd as pds
    emb = pds.Embedding()
    assert abs(float(sum(v.double().sum() for v in emb.state_dict().values())) - g['weight_checksum'].item()) < 1e-9
    p = helpers.prefixed(emb.state_dict(), '_embedding')
    leaf = g['image'].clone().requires_grad_(True)
    padded, rows, columns = oracle.pad_to_multiple(leaf)
    assert (rows, columns) == (27, 13)
    d, s = oracle.embedding(p, '_embedding', padded)
    ((d * g['wd']).sum() + (s * g['ws']).sum()).backward()
    scale = float(g['grad_image_fp64'].abs().max())
    floor = helpers.maxdiff(g['grad_image_fp32'], g['grad_image_fp64']) / scale
    assert helpers.maxdiff(leaf.grad, g['grad_image_fp64']) / scale <= max(1e-5, 3.0 * floor)


def test_errors_golden():
    """reference test/test_errors.py:13-66 known answers and a seeded case with an inf band (errors.py:9-74)."""
    g = helpers.golden('g10_errors')
    pixelwise, mean = oracle.absolute_error(g['ref_est'], g['ref_gt'])
    assert torch.equal(pixelwise, torch.tensor([[1.0, 0.0], [0.0, 3.0]])) and abs(mean - 4.0 / 3.0) < 1e-3 * 4 / 3
    assert abs(oracle.absolute_error(g['ref_est'], g['ref_gt'], use_mean=False)[1] - 1.0) < 1e-3
    bad, percent = oracle.n_pixels_error(g['ref_est'], g['ref_gt'], n=1.0)
    assert torch.equal(bad, torch.tensor([[0.0, 0.0], [0.0, 1.0]])) and abs(percent - 100.0 / 3.0) < 1e-3 * 100 / 3
    nothing = torch.full((2, 2), float('inf'))
    assert oracle.absolute_error(g['ref_est'], nothing)[1] == 0.0
    assert oracle.n_pixels_error(g['ref_est'], nothing, n=1.0)[1] == 0.0
    pixelwise, mean = oracle.absolute_error(g['random_est'], g['random_gt'])
    assert torch.equal(pixelwise, g['random_pixelwise']) and abs(mean - g['random_mean'].item()) < 1e-6
    assert abs(oracle.absolute_error(g['random_est'], g['random_gt'], False)[1] - g['random_median'].item()) < 1e-6
    bad, percent = oracle.n_pixels_error(g['random_est'], g['random_gt'])
    assert torch.equal(bad, g['random_bad']) and abs(percent - g['random_percent'].item()) < 1e-5
