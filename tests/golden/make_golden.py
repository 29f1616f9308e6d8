"""Generates the committed golden fixtures and pins oracle/pds_oracle.py against the reference.

Run ONLY in the build container, where the reference is importable:

    python tests/golden/make_golden.py

It imports /root/reference (read-only), runs the reference modules on seeded inputs, asserts the
oracle restatement reproduces them, and writes small ``.npz`` fixtures (inputs, expected outputs,
weight checksums) next to this file plus ``pinning_report.json`` with the measured differences.
Nothing of the reference travels: fixtures are data only.  Weights are never stored; they are
re-created by ``torch.manual_seed`` + constructing this repo's own modules (identical construction
order) and guarded by an fp64 checksum.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from practical_deep_stereo import embedding as ref_embedding  # noqa: E402
from practical_deep_stereo import estimator as ref_estimator  # noqa: E402
from practical_deep_stereo import matching as ref_matching  # noqa: E402
from practical_deep_stereo import network as ref_network  # noqa: E402
from practical_deep_stereo import regularization as ref_regularization  # noqa: E402
from practical_deep_stereo import size_adapter as ref_size_adapter  # noqa: E402

from oracle import pds_oracle as oracle  # noqa: E402

REPORT = {}


def checksum(state_dict):
    return float(sum(v.double().sum() for v in state_dict.values()))


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def prefixed(state_dict, prefix):
    return {prefix + '.' + k: v for k, v in state_dict.items()}


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def mock_operation(x):
    return torch.max(x, dim=1, keepdim=True)[0]


@torch.no_grad()
def g1_matching_mock():
    left = torch.tensor([0., 2., 1., 2.]).view(1, 1, 1, 4)
    right = torch.tensor([3., 4., 2., 4.]).view(1, 1, 1, 4)
    outs = {}
    for n in (2, 1, 0):
        ref = ref_matching.Matching(maximum_disparity=n, operation=mock_operation)(left, right)
        mine = oracle.matching(left, right, n, mock_operation)
        assert torch.equal(ref, mine)
        outs['out_max%d' % n] = ref
    # the reference's own expected values, test/test_matching.py:22-31
    assert np.array_equal(outs['out_max2'].numpy().reshape(3, 4), [[3, 4, 2, 4], [0, 3, 4, 2], [0, 2, 3, 4]])
    # a wider random case with several channels and rows
    g = torch.Generator().manual_seed(11)
    l2 = torch.randn(2, 3, 4, 9, generator=g)
    r2 = torch.randn(2, 3, 4, 9, generator=g)
    ref = ref_matching.Matching(maximum_disparity=5, operation=mock_operation)(l2, r2)
    assert torch.equal(ref, oracle.matching(l2, r2, 5, mock_operation))
    save('g1_matching_mock', left=left, right=right, left2=l2, right2=r2, out2_max5=ref, **outs)
    REPORT['g1_matching_mock'] = 'bit-exact'


@torch.no_grad()
def g2_matching_operation():
    torch.manual_seed(0)
    op = ref_matching.MatchingOperation()
    net = ref_matching.Matching(maximum_disparity=15, operation=op)
    g = torch.Generator().manual_seed(2)
    left = torch.randn(1, 64, 16, 32, generator=g)
    right = torch.randn(1, 64, 16, 32, generator=g)
    ref = net(left, right)
    p = prefixed(op.state_dict(), '_m._operation')
    mine = oracle.matching_with_operation(p, '_m', left, right, 15)
    d = maxdiff(ref, mine)
    assert d <= 1e-6, d
    # standalone MatchingOperation on an explicit concatenation, batch 2, odd size
    x = torch.rand(2, 128, 25, 25, generator=g)
    ref_op = op(x)
    mine_op = oracle.matching_operation(p, '_m._operation', x)
    assert maxdiff(ref_op, mine_op) <= 1e-6
    save('g2_matching', left=left, right=right, signatures=ref, weight_checksum=checksum(op.state_dict()),
         concatenated=x, operation_out=ref_op)
    REPORT['g2_matching'] = {'oracle_vs_reference_max': d,
                             'operation_max': maxdiff(ref_op, mine_op)}


@torch.no_grad()
def g3_regularization():
    torch.manual_seed(0)
    reg = ref_regularization.Regularization()
    g = torch.Generator().manual_seed(3)
    ms = torch.randn(1, 8, 16, 16, 32, generator=g)
    shortcut = torch.randn(1, 8, 16, 32, generator=g)
    ref = reg(ms, shortcut)
    p = prefixed(reg.state_dict(), '_r')
    mine = oracle.regularization(p, '_r', ms, shortcut)
    d = maxdiff(ref, mine)
    assert d <= 1e-6, d
    save('g3_regularization', signatures=ms, shortcut=shortcut, cost=ref,
         weight_checksum=checksum(reg.state_dict()))
    REPORT['g3_regularization'] = {'oracle_vs_reference_max': d}


@torch.no_grad()
def g4_blocks():
    torch.manual_seed(0)
    x = torch.rand(2, 6, 10, 14, 16)
    con = ref_regularization.ContractionBlock3d(number_of_features=6)
    down, smooth = con(x)
    pc = prefixed(con.state_dict(), '_c')
    d1, s1 = oracle.contraction_block_3d(pc, '_c', x)
    assert maxdiff(down, d1) <= 1e-6 and maxdiff(smooth, s1) <= 1e-6
    torch.manual_seed(0)
    xi = torch.rand(2, 6, 10, 14, 16)
    sc = torch.rand(2, 3, 20, 28, 32)
    exp = ref_regularization.ExpansionBlock3d(number_of_features=6)
    out = exp(xi, sc)
    pe = prefixed(exp.state_dict(), '_e')
    o1 = oracle.expansion_block_3d(pe, '_e', xi, sc)
    assert maxdiff(out, o1) <= 1e-6
    save('g4_blocks', contraction_in=x, contraction_down=down, contraction_smooth=smooth,
         contraction_checksum=checksum(con.state_dict()),
         expansion_in=xi, expansion_shortcut=sc, expansion_out=out,
         expansion_checksum=checksum(exp.state_dict()))
    REPORT['g4_blocks'] = {'contraction_max': max(maxdiff(down, d1), maxdiff(smooth, s1)),
                           'expansion_max': maxdiff(out, o1)}


@torch.no_grad()
def g5_subpixel_map():
    cases = {}
    worst = 0.0
    sim5 = torch.tensor([0.1, 0.4, 0.3, 0.2, 0.3]).view(1, 5, 1, 1)
    for name, sim, hw, step in [
            ('ref_test_21', sim5, 2, 1),                      # test/test_estimator.py:14-21 -> 1.52
            ('ref_test_22', sim5, 2, 2),                      # test/test_estimator.py:23-27 -> 2.124
            ('tie_first', torch.tensor([1., .5, 1., .2, 1., .1]).view(1, 6, 1, 1), 4, 2),
            ('all_equal', torch.ones(1, 6, 1, 1), 4, 2),
            ('edge_low', torch.tensor([2., .5, 1., .2, 1., .1]).view(1, 6, 1, 1), 4, 2),
            ('edge_high', torch.tensor([0., .5, 1., .2, 1., 3.1]).view(1, 6, 1, 1), 4, 2),
            ('random_42', torch.randn(2, 32, 8, 8, generator=torch.Generator().manual_seed(5)), 4, 2),
            ('random_41', torch.randn(2, 9, 5, 7, generator=torch.Generator().manual_seed(6)), 4, 1),
            ('random_84', torch.randn(1, 40, 6, 12, generator=torch.Generator().manual_seed(7)), 8, 4),
            ('random_102', torch.randn(1, 24, 3, 5, generator=torch.Generator().manual_seed(8)), 10, 2),
            ('single_plane', torch.randn(1, 1, 4, 4, generator=torch.Generator().manual_seed(9)), 4, 2)]:
        ref = ref_estimator.SubpixelMap(half_support_window=hw, disparity_step=step)(sim)
        mine = oracle.subpixel_map(sim, hw, step)
        d = maxdiff(ref, mine)
        assert d <= 1e-5, (name, d)
        worst = max(worst, d)
        cases[name + '_in'] = sim
        cases[name + '_out'] = ref
        cases[name + '_cfg'] = np.array([hw, step])
    assert abs(float(cases['ref_test_21_out'].reshape(-1)[0]) - 1.52) < 1e-4
    assert abs(float(cases['ref_test_22_out'].reshape(-1)[0]) - 2.124) < 1e-4
    for bad in [(4, 0), (0, 2), (3, 2)]:
        for cls in (ref_estimator.SubpixelMap,):
            try:
                cls(half_support_window=bad[0], disparity_step=bad[1])
                raise AssertionError('expected ValueError')
            except ValueError:
                pass
        try:
            oracle.check_subpixel_map_arguments(*bad)
            raise AssertionError('expected ValueError')
        except ValueError:
            pass
    save('g5_subpixel_map', **cases)
    REPORT['g5_subpixel_map'] = {'oracle_vs_reference_max': worst}


def g8_loss():
    """SubpixelCrossEntropy: the reference's own known answer (test/test_loss.py:12-37: 1.3654 and the gradient
    table) and a seeded random case with an inf band, value + gradient."""
    from practical_deep_stereo import loss as ref_loss
    sim = torch.tensor([[0.1, 0.3, 0.2, 0.05], [0.2, 0.1, 0.4, 0.0], [0.2, 0.1, 0.4, 0.0]])
    sim = sim.t().contiguous().view(1, 4, 3, 1).requires_grad_(True)
    gt = torch.tensor([[1.3], [float('inf')], [1.9]]).view(1, 3, 1)
    w = torch.tensor([[0.9], [0.0], [0.01]]).view(1, 3, 1).requires_grad_(True)
    value = ref_loss.SubpixelCrossEntropy(diversity=2.0, disparity_step=1)(sim, gt, w)
    value.backward()
    assert abs(value.item() - 1.3654) < 1e-3
    sim_o = sim.detach().clone().requires_grad_(True)
    w_o = w.detach().clone().requires_grad_(True)
    value_o = oracle.subpixel_cross_entropy(sim_o, gt, w_o, 2.0, 1)
    value_o.backward()
    assert abs(value_o.item() - value.item()) < 1e-6 and maxdiff(sim_o.grad, sim.grad) < 1e-7
    assert maxdiff(w_o.grad, w.grad) < 1e-7
    g = torch.Generator().manual_seed(21)
    sim2 = (torch.randn(2, 32, 9, 13, generator=g) * 0.6).requires_grad_(True)
    gt2 = torch.rand(2, 9, 13, generator=g) * 62
    gt2[:, 2:4, :] = float('inf')
    w2 = torch.rand(2, 9, 13, generator=g)
    out = {}
    for name, weights in (('plain', None), ('weighted', w2)):
        s_ref = sim2.detach().clone().requires_grad_(True)
        if weights is not None:
            weights = weights.clone().requires_grad_(True)   # loss.py:73-77: the weights receive a gradient too
        v = ref_loss.SubpixelCrossEntropy()(s_ref, gt2, weights)
        v.backward()
        if weights is not None:
            out['random_weighted_weights_grad'] = weights.grad
            weights = weights.detach()
        s_or = sim2.detach().clone().requires_grad_(True)
        vo = oracle.subpixel_cross_entropy(s_or, gt2, weights)
        vo.backward()
        assert abs(v.item() - vo.item()) < 1e-5 and maxdiff(s_ref.grad, s_or.grad) < 1e-7, name
        out['random_' + name + '_value'] = v.detach()
        out['random_' + name + '_grad'] = s_ref.grad
    save('g8_loss', ref_sim=sim.detach(), ref_gt=gt, ref_weights=w.detach(), ref_weights_grad=w.grad,
         ref_value=value.detach(), ref_grad=sim.grad,
         random_sim=sim2.detach(), random_gt=gt2, random_weights=w2, **out)
    REPORT['g8_loss'] = {'reference_value': value.item()}


@torch.no_grad()
def g9_embedding():
    """Embedding (embedding.py:11-65) after SizeAdapter.pad (size_adapter.py:29-43) on an image whose size is
    not a multiple of 64 (rows on top and columns on the left are padded), plus a bare odd-sized call."""
    torch.manual_seed(0)
    emb = ref_embedding.Embedding()
    g = torch.Generator().manual_seed(3)
    image = torch.rand(2, 3, 100, 150, generator=g) * 255
    adapter = ref_size_adapter.SizeAdapter()
    padded = adapter.pad(image)
    descriptor, shortcut = emb(padded)
    p = prefixed(emb.state_dict(), '_embedding')
    padded_o, rows, columns = oracle.pad_to_multiple(image)
    assert torch.equal(padded_o, padded) and (rows, columns) == (28, 42)
    d_o, s_o = oracle.embedding(p, '_embedding', padded_o)
    odd = torch.rand(1, 3, 37, 51, generator=g) * 255
    d_odd, s_odd = emb(odd)
    d_odd_o, s_odd_o = oracle.embedding(p, '_embedding', odd)
    rep = {'descriptor_max': maxdiff(descriptor, d_o), 'shortcut_max': maxdiff(shortcut, s_o),
           'odd_max': max(maxdiff(d_odd, d_odd_o), maxdiff(s_odd, s_odd_o))}
    assert max(rep.values()) <= 1e-5, rep
    save('g9_embedding', image=image, descriptor=descriptor, shortcut=shortcut, odd_image=odd,
         odd_descriptor=d_odd, odd_shortcut=s_odd, weight_checksum=checksum(emb.state_dict()))
    REPORT['g9_embedding'] = rep


def g12_image_gradient():
    """The gradient the reference's autograd delivers to the IMAGE (embedding.py:32,46-65 behind SizeAdapter.pad,
    size_adapter.py:29-43): d/d image of sum(descriptor * wd) + sum(shortcut * ws) in the reference's own fp64 run
    (the fp32 run rides along as the noise floor), on an image whose size needs padding on top and on the left."""
    torch.manual_seed(0)
    emb = ref_embedding.Embedding()
    g = torch.Generator().manual_seed(12)
    image = torch.rand(1, 3, 37, 51, generator=g) * 255
    adapter = ref_size_adapter.SizeAdapter()
    wd = ws = None
    grads = {}
    for dtype in (torch.float64, torch.float32):
        net = ref_embedding.Embedding().to(dtype)
        net.load_state_dict({k: v.to(dtype) for k, v in emb.state_dict().items()})
        leaf = image.clone().to(dtype).requires_grad_(True)
        descriptor, shortcut = net(adapter.pad(leaf))
        if wd is None:
            wd = torch.randn(descriptor.shape, generator=g)
            ws = torch.randn(shortcut.shape, generator=g)
        ((descriptor * wd.to(dtype)).sum() + (shortcut * ws.to(dtype)).sum()).backward()
        grads[dtype] = leaf.grad
    p = {k: v.double() for k, v in prefixed(emb.state_dict(), '_embedding').items()}
    leaf = image.clone().double().requires_grad_(True)
    padded, rows, columns = oracle.pad_to_multiple(leaf)
    d_o, s_o = oracle.embedding(p, '_embedding', padded)
    ((d_o * wd.double()).sum() + (s_o * ws.double()).sum()).backward()
    scale = float(grads[torch.float64].abs().max())
    rep = {'oracle_vs_reference_fp64_rel': maxdiff(leaf.grad, grads[torch.float64]) / scale,
           'reference_fp32_vs_fp64_rel': maxdiff(grads[torch.float32], grads[torch.float64]) / scale,
           'pad': [rows, columns]}
    assert rep['oracle_vs_reference_fp64_rel'] <= 1e-9, rep
    save('g12_image_gradient', image=image, wd=wd, ws=ws, grad_image_fp64=grads[torch.float64],
         grad_image_fp32=grads[torch.float32], weight_checksum=checksum(emb.state_dict()))
    REPORT['g12_image_gradient'] = rep


def g10_errors():
    """errors.py:9-74: the reference's own known answers (test/test_errors.py:13-66) and a seeded random case with
    an inf band; pixel-wise maps, mean / median absolute error and n-pixels error."""
    from practical_deep_stereo import errors as ref_errors
    est = torch.tensor([[1.0, 2.0], [3.0, 4.0]])
    gt = torch.tensor([[2.0, 2.0], [float('inf'), 1.0]])
    pix, mean = ref_errors.compute_absolute_error(est, gt, use_mean=True)
    _, median = ref_errors.compute_absolute_error(est, gt, use_mean=False)
    bad, percent = ref_errors.compute_n_pixels_error(est, gt, n=1.0)
    assert torch.equal(pix, torch.tensor([[1.0, 0.0], [0.0, 3.0]])) and abs(mean - 4.0 / 3.0) < 1e-3
    assert median == 1.0 and torch.equal(bad, torch.tensor([[0.0, 0.0], [0.0, 1.0]]))
    assert abs(percent - 100.0 / 3.0) < 1e-3
    nothing = torch.full((2, 2), float('inf'))
    assert ref_errors.compute_absolute_error(est, nothing)[1] == 0.0
    assert ref_errors.compute_n_pixels_error(est, nothing, n=1.0)[1] == 0.0
    g = torch.Generator().manual_seed(31)
    est2 = torch.rand(2, 37, 53, generator=g) * 190
    gt2 = est2 + torch.randn(2, 37, 53, generator=g) * 4
    gt2[:, 5:9, :] = float('inf')
    gt2[0, 20, 3] = -float('inf')
    pix2, mean2 = ref_errors.compute_absolute_error(est2, gt2, use_mean=True)
    _, median2 = ref_errors.compute_absolute_error(est2, gt2, use_mean=False)
    bad2, percent2 = ref_errors.compute_n_pixels_error(est2, gt2)
    worst = 0.0
    for (e, t) in ((est, gt), (est2, gt2), (est, nothing)):
        for use_mean in (True, False):
            a_ref = ref_errors.compute_absolute_error(e, t, use_mean)
            a_or = oracle.absolute_error(e, t, use_mean)
            assert torch.equal(a_ref[0], a_or[0])
            worst = max(worst, abs(a_ref[1] - a_or[1]))
        n_ref = ref_errors.compute_n_pixels_error(e, t, n=1.0)
        n_or = oracle.n_pixels_error(e, t, n=1.0)
        assert torch.equal(n_ref[0], n_or[0])
        worst = max(worst, abs(n_ref[1] - n_or[1]))
    assert worst == 0.0
    save('g10_errors', ref_est=est, ref_gt=gt, ref_pixelwise=pix, ref_mean=mean, ref_median=median, ref_bad=bad,
         ref_percent=percent, random_est=est2, random_gt=gt2, random_pixelwise=pix2, random_mean=mean2,
         random_median=median2, random_bad=bad2, random_percent=percent2)
    REPORT['g10_errors'] = {'oracle_vs_reference_max': worst, 'reference_mean': mean, 'reference_percent': percent}


def images(batch, height, width):
    g = torch.Generator().manual_seed(1)
    left = torch.rand(batch, 3, height, width, generator=g) * 255
    right = torch.rand(batch, 3, height, width, generator=g) * 255
    return left, right


def stage_stats(t):
    d = t.double()
    return np.array([d.mean().item(), d.std().item(), d.min().item(), d.max().item(), d.sum().item(),
                     d.abs().sum().item()])


@torch.no_grad()
def g6_config1_network():
    """Config 1: full PdsNetwork.default(63) eval forward on a 128x256 pair."""
    torch.manual_seed(0)
    net = ref_network.PdsNetwork.default(63).eval()
    left, right = images(1, 128, 256)
    ld, shortcut = net._embedding(net._size_adapter.pad(left))
    rd = net._embedding(net._size_adapter.pad(right))[0]
    ms = net._matching(ld, rd)
    cost = net._regularization(ms, shortcut)
    disparity = net._estimator(cost)
    full = net(left, right)
    assert torch.equal(full, disparity)
    p = net.state_dict()
    ms_o, cost_o, disp_o = oracle.hot_path(p, ld, rd, shortcut, 63, return_stages=True)
    rep = {'ms_max': maxdiff(ms, ms_o), 'cost_max': maxdiff(cost, cost_o), 'disparity_max': maxdiff(disparity, disp_o)}
    assert rep['ms_max'] <= 1e-6 and rep['cost_max'] <= 1e-5 and rep['disparity_max'] <= 1e-3, rep
    save('g6_config1', disparity=disparity, left_descriptor_stats=stage_stats(ld),
         signatures_stats=stage_stats(ms), cost_stats=stage_stats(cost),
         cost_sub=cost[:, ::4, ::8, ::8].contiguous(), signatures_sub=ms[:, :, ::2, ::4, ::4].contiguous(),
         weight_checksum=checksum(p))
    REPORT['g6_config1'] = rep


@torch.no_grad()
def g7_config2_statistics():
    """Config 2 (960x540, D=192): per-stage statistics and a sub-sample, not the 212 MB tensors."""
    torch.manual_seed(0)
    net = ref_network.PdsNetwork.default(191).eval()
    left, right = images(1, 540, 960)
    ld, shortcut = net._embedding(net._size_adapter.pad(left))
    rd = net._embedding(net._size_adapter.pad(right))[0]
    ms = net._matching(ld, rd)
    cost = net._regularization(ms, shortcut)
    disparity = net._estimator(cost)
    p = net.state_dict()
    ms_o, cost_o, disp_o = oracle.hot_path(p, ld, rd, shortcut, 191, return_stages=True)
    delta = (disparity - disp_o).abs()
    rep = {'ms_max': maxdiff(ms, ms_o), 'cost_max': maxdiff(cost, cost_o),
           'disparity_mae': float(delta.mean()), 'disparity_flips': float((delta > 0.5).double().mean())}
    assert rep['ms_max'] <= 2e-5 and rep['cost_max'] <= 1e-4 and rep['disparity_mae'] <= 1e-3, rep
    save('g7_config2_stats', signatures_stats=stage_stats(ms), cost_stats=stage_stats(cost),
         disparity_stats=stage_stats(disparity), disparity_sub=disparity[:, ::16, ::16].contiguous(),
         cost_sub=cost[:, ::8, ::32, ::32].contiguous(),
         signatures_sub=ms[:, :, ::4, ::16, ::16].contiguous(), weight_checksum=checksum(p))
    REPORT['g7_config2'] = rep


def g11_training_step():
    """SURVEY.md 8c fixture "G9" (the file name g9 was already taken by the descriptor network): ONE training step of
    the reference -- train-mode PdsNetwork.default(63) on the 128x256 pair of config 1, SubpixelCrossEntropy against a
    seeded ground truth with an unknown (inf) band, backward (pds_trainer.py:35-46, loss.py:30-78).  Stored: the loss,
    the norm of the gradient of every parameter tensor, and a sub-sample of dL/dcost."""
    from practical_deep_stereo import loss as ref_loss
    torch.manual_seed(0)
    net = ref_network.PdsNetwork.default(63).train()
    left, right = images(1, 128, 256)
    g = torch.Generator().manual_seed(31)
    gt = torch.rand(1, 128, 256, generator=g) * 62.0
    gt[:, :, :24] = float('inf')
    cost = net(left, right)
    cost.retain_grad()
    value = ref_loss.SubpixelCrossEntropy()(cost, gt)
    value.backward()
    # the oracle's training-mode network + loss reproduce the reference (value, cost volume, every gradient)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    cost_o = oracle.network_training_output(p, left, right, 63)
    cost_o.retain_grad()
    value_o = oracle.subpixel_cross_entropy(cost_o, gt)
    value_o.backward()
    names = [n for n, _ in net.named_parameters()]
    grads = {n: q.grad for n, q in net.named_parameters()}
    # the same step of the reference in fp64: the arbiter.  Weight gradients sum up to millions of sign-cancelling
    # terms, so two fp32 evaluations that only differ in summation order (the reference loops over the disparities,
    # the oracle batches them) are a few percent of the largest entry apart in the worst tensors; what a third
    # implementation can be held to is its distance from fp64 relative to the reference's own.
    import copy
    net64 = copy.deepcopy(net).double()
    net64.zero_grad()
    cost64 = net64(left.double(), right.double())
    ref_loss.SubpixelCrossEntropy()(cost64, gt.double()).backward()
    grads64 = {n: q.grad for n, q in net64.named_parameters()}

    def rel(a, b):
        return maxdiff(a, b) / (float(b.abs().max()) + 1e-30)
    oracle_rel = np.array([rel(p[n].grad, grads64[n]) for n in names])
    reference_rel = np.array([rel(grads[n], grads64[n]) for n in names])
    rep = {'loss': value.item(), 'loss_oracle_diff': abs(value.item() - value_o.item()),
           'cost_oracle_max': maxdiff(cost, cost_o), 'dcost_oracle_max': maxdiff(cost.grad, cost_o.grad),
           'param_grad_reference_fp32_vs_fp64_rel_max': float(reference_rel.max()),
           'param_grad_oracle_fp32_vs_fp64_rel_max': float(oracle_rel.max())}
    assert rep['loss_oracle_diff'] <= 1e-6 and rep['cost_oracle_max'] <= 1e-6 and rep['dcost_oracle_max'] <= 1e-9, rep
    # the oracle is no further from fp64 than the reference is (both are fp32 evaluations of the same formulas);
    # tensors whose true gradient is zero (the bias in front of the soft-max: 1e-17 in fp64) carry only rounding noise
    live = np.array([grads64[n].norm().item() > 1e-9 for n in names])
    rep['param_grad_reference_fp32_vs_fp64_rel_max'] = float(reference_rel[live].max())
    rep['param_grad_oracle_fp32_vs_fp64_rel_max'] = float(oracle_rel[live].max())
    assert float(oracle_rel[live].max()) <= max(3.0 * float(reference_rel[live].max()), 1e-4), rep
    # element-wise evidence: a strided sub-sample (at most 512 entries) of every parameter gradient, from the
    # reference's fp32 run and from its fp64 run, concatenated; tensor i owns [grad_sub_offsets[i], grad_sub_offsets[i+1])
    def sub(t):
        flat = t.detach().flatten()
        return flat[::max(1, -(-flat.numel() // 512))]
    subs32 = [sub(grads[n]).double() for n in names]
    subs64 = [sub(grads64[n]) for n in names]
    offsets = np.cumsum([0] + [t.numel() for t in subs64])
    save('g11_training_step', loss=value.detach(), ground_truth=gt,
         grad_sub_offsets=offsets, grad_sub=torch.cat(subs32).float(), grad_sub_fp64=torch.cat(subs64),
         grad_norms=np.array([grads[n].double().norm().item() for n in names]),
         grad_norms_fp64=np.array([grads64[n].norm().item() for n in names]),
         grad_abs_max=np.array([grads[n].abs().max().item() for n in names]),
         grad_reference_vs_fp64_rel=reference_rel,
         parameter_names=np.array(names), dcost_sub=cost.grad[:, ::4, ::8, ::8].contiguous(),
         cost_sub=cost.detach()[:, ::4, ::8, ::8].contiguous(), weight_checksum=checksum(net.state_dict()))
    REPORT['g11_training_step'] = rep


def g13_config5_gradients():
    """BASELINE.json configs[4] at FULL size: ONE training step of the reference -- train-mode
    PdsNetwork.default(191) on the 540x960 pair of the input recipe, SubpixelCrossEntropy against a seeded ground truth
    with an unknown (inf) band, backward (pds_trainer.py:35-46, loss.py:30-78) -- in fp32 AND in fp64.  Stored per
    parameter tensor: the gradient's norm and largest entry of both runs, the reference's own fp32-vs-fp64 distance, and a
    strided sub-sample of at most 512 entries of both (a few hundred KB in all).  The weight gradients of the 64-channel
    layers sum 1.6 M sign-cancelling terms per entry here -- the regime the small fixture G11 cannot reach."""
    import copy
    import gc
    from practical_deep_stereo import loss as ref_loss
    torch.manual_seed(0)
    net = ref_network.PdsNetwork.default(191).train()
    left, right = images(1, 540, 960)
    g = torch.Generator().manual_seed(41)
    gt = torch.rand(1, 540, 960, generator=g) * 190.0
    gt[:, :17] = float('inf')
    names = [n for n, _ in net.named_parameters()]

    def sub(t):
        flat = t.detach().flatten()
        return flat[::max(1, -(-flat.numel() // 512))]

    def run(network, dtype):
        network.zero_grad()
        cost = network(left.to(dtype), right.to(dtype))
        value = ref_loss.SubpixelCrossEntropy()(cost, gt.to(dtype))
        value.backward()
        grads = {n: q.grad.detach().clone() for n, q in network.named_parameters()}
        out = (float(value), grads, cost.detach()[:, ::8, ::36, ::64].contiguous().clone())
        del cost, value
        gc.collect()
        return out
    loss32, grads32, cost_sub32 = run(net, torch.float32)
    net64 = copy.deepcopy(net).double()
    net64.zero_grad()
    loss64, grads64, cost_sub64 = run(net64, torch.float64)
    del net64
    gc.collect()

    def rel(a, b):
        return maxdiff(a, b) / (float(b.abs().max()) + 1e-300)
    reference_rel = np.array([rel(grads32[n], grads64[n]) for n in names])
    live = np.array([grads64[n].norm().item() > 1e-9 for n in names])
    subs32 = [sub(grads32[n]).double() for n in names]
    subs64 = [sub(grads64[n]) for n in names]
    offsets = np.cumsum([0] + [t.numel() for t in subs64])
    save('g13_config5_gradients', loss=np.array([loss32]), loss_fp64=np.array([loss64]), ground_truth_seed=np.array([41]),
         grad_sub_offsets=offsets, grad_sub=torch.cat(subs32).float(), grad_sub_fp64=torch.cat(subs64),
         grad_norms=np.array([grads32[n].double().norm().item() for n in names]),
         grad_norms_fp64=np.array([grads64[n].norm().item() for n in names]),
         grad_abs_max_fp64=np.array([grads64[n].abs().max().item() for n in names]),
         grad_reference_vs_fp64_rel=reference_rel, live=live,
         parameter_names=np.array(names), cost_sub=cost_sub32, cost_sub_fp64=cost_sub64,
         weight_checksum=checksum(net.state_dict()))
    REPORT['g13_config5_gradients'] = {
        'loss_fp32': loss32, 'loss_fp64': loss64,
        'param_grad_reference_fp32_vs_fp64_rel_max': float(reference_rel[live].max()),
        'param_grad_reference_fp32_vs_fp64_rel_median': float(np.median(reference_rel[live])),
        'tensors': len(names), 'tensors_with_nonzero_gradient': int(live.sum())}


def injection_into_reference():
    """The reference-side half of the drop-in claim (BASELINE.json north_star: "network.py/pds_trainer.py drop them in
    unchanged"): the REFERENCE's own PdsNetwork (network.py:17-24) and PdsTrainer (pds_trainer.py:35-46, trainer.py:87-122)
    are constructed around THIS package's modules, exactly as INTEGRATION.md section 1 shows, and exercised as far as a
    GPU-less container allows: registration, state-dict keys in order, set_maximum_disparity reaching the injected
    Matching, load_state_dict of a reference checkpoint, train()/eval() propagation, and a th.save / th.load round trip
    in trainer.py:110-122's checkpoint format through the reference trainer's own _save_checkpoint / load_checkpoint.
    No forward pass runs here (the modules have no CPU path by design); the GPU side of the same construction is
    tests/test_gpu_parity.py.  The outcome goes to pinning_report.json."""
    import tempfile
    import practicaldeepstereo_nips2018_amd as pds_amd
    from practical_deep_stereo import pds_trainer as ref_pds_trainer
    rep = {}
    torch.manual_seed(0)
    reference = ref_network.PdsNetwork.default(191)
    reference_state = {k: v.clone() for k, v in reference.state_dict().items()}
    torch.manual_seed(123)   # different initial weights: load_state_dict below has to overwrite every tensor
    injected = ref_network.PdsNetwork(
        ref_size_adapter.SizeAdapter(), pds_amd.Embedding(),
        pds_amd.Matching(0, pds_amd.MatchingOperation()), pds_amd.Regularization(), pds_amd.SubpixelMap())
    assert type(injected).__module__ == 'practical_deep_stereo.network'
    # (1) module registration and state-dict surface: same keys, same order, same shapes
    keys, ref_keys = list(injected.state_dict().keys()), list(reference_state.keys())
    assert keys == ref_keys, [k for k in keys if k not in ref_keys][:5]
    assert all(injected.state_dict()[k].shape == reference_state[k].shape for k in keys)
    assert [n for n, _ in injected.named_parameters()] == [n for n, _ in reference.named_parameters()]
    rep['state_dict_keys'] = len(keys)
    rep['state_dict_keys_equal_in_order'] = True
    # (2) set_maximum_disparity (network.py:26-36) reaches the injected Matching
    injected.set_maximum_disparity(191)
    assert injected._matching._maximum_disparity == 47 and injected._maximum_disparity == 191
    try:
        injected.set_maximum_disparity(100)
        raise AssertionError('ValueError expected')
    except ValueError:
        pass
    rep['set_maximum_disparity_reaches_matching'] = True
    # (3) a reference checkpoint loads (strict) and round-trips bit-exactly
    differed = sum(int(not torch.equal(injected.state_dict()[k], reference_state[k])) for k in keys)
    assert differed >= 60, differed   # every convolution weight (InstanceNorm affine terms start at 1 / 0 in both)
    result = injected.load_state_dict(reference_state)
    assert not result.missing_keys and not result.unexpected_keys
    assert all(torch.equal(injected.state_dict()[k], reference_state[k]) for k in keys)
    assert abs(checksum(injected.state_dict()) - checksum(reference_state)) == 0.0
    rep['load_state_dict_round_trip_bit_exact'] = True
    rep['parameter_checksum'] = checksum(injected.state_dict())
    # (4) train() / eval() propagate to the injected modules (network.py:50 branches on .training)
    injected.eval()
    assert not any(m.training for m in injected.modules())
    injected.train()
    assert all(m.training for m in injected.modules())
    rep['train_eval_propagate'] = True
    # (5) the reference trainer around the injected network: its own checkpoint writer and reader
    # (trainer.py:87-122), optimizer and scheduler as train_on_flyingthings3d.py builds them
    with tempfile.TemporaryDirectory() as folder:
        def make_trainer(network):
            optimizer = torch.optim.RMSprop(network.parameters(), lr=1e-2)
            scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=1, gamma=0.5)
            return ref_pds_trainer.PdsTrainer({
                'network': network, 'optimizer': optimizer, 'learning_rate_scheduler': scheduler,
                'criterion': pds_amd.SubpixelCrossEntropy(), 'training_set_loader': [], 'test_set_loader': [],
                'experiment_folder': folder, 'end_epoch': 2})
        trainer = make_trainer(injected)
        trainer._initialize_filenames()
        trainer._training_losses, trainer._test_errors = [1.25], [7.5]
        trainer._save_checkpoint()
        filename = trainer._checkpoint_template.format(1)
        assert os.path.exists(filename)
        checkpoint = torch.load(filename)
        assert sorted(checkpoint.keys()) == ['learning_rate_scheduler', 'network', 'optimizer', 'test_errors',
                                             'training_losses']
        assert list(checkpoint['network'].keys()) == ref_keys
        # ... read back into a FRESH injected network by the reference trainer, and into the reference's own network
        torch.manual_seed(7)
        fresh = ref_network.PdsNetwork(
            ref_size_adapter.SizeAdapter(), pds_amd.Embedding(),
            pds_amd.Matching(0, pds_amd.MatchingOperation()), pds_amd.Regularization(), pds_amd.SubpixelMap())
        fresh.set_maximum_disparity(191)
        second = make_trainer(fresh)
        second.load_checkpoint(filename)
        assert second._current_epoch == 1 and second._training_losses == [1.25] and second._test_errors == [7.5]
        assert all(torch.equal(fresh.state_dict()[k], reference_state[k]) for k in keys)
        torch.manual_seed(9)
        plain = ref_network.PdsNetwork.default(191)
        make_trainer(plain).load_checkpoint(filename, load_only_network=True)
        assert all(torch.equal(plain.state_dict()[k], reference_state[k]) for k in keys)
    rep['reference_trainer_checkpoint_round_trip'] = True
    # (6) this package's own PdsNetwork carries the same surface (what INTEGRATION.md section 1 offers as the short form)
    torch.manual_seed(0)
    own = pds_amd.PdsNetwork.default(191)
    assert list(own.state_dict().keys()) == ref_keys
    assert all(torch.equal(own.state_dict()[k], reference_state[k]) for k in keys)   # identical construction order
    rep['own_network_seed0_equals_reference_seed0'] = True
    REPORT['injection_into_reference'] = rep


if __name__ == '__main__':
    torch.set_num_threads(8)
    if '--only' in sys.argv:  # regenerate one fixture, keep the rest of the report
        name = sys.argv[sys.argv.index('--only') + 1]
        with open(os.path.join(HERE, 'pinning_report.json')) as f:
            REPORT.update(json.load(f))
        globals()[name]()
        with open(os.path.join(HERE, 'pinning_report.json'), 'w') as f:
            json.dump(REPORT, f, indent=2, sort_keys=True)
        print(json.dumps(REPORT[name], indent=2, sort_keys=True))
        sys.exit(0)
    g1_matching_mock()
    g2_matching_operation()
    g3_regularization()
    g4_blocks()
    g5_subpixel_map()
    g6_config1_network()
    g8_loss()
    g9_embedding()
    g10_errors()
    g11_training_step()
    g12_image_gradient()
    injection_into_reference()
    if '--skip-config2' not in sys.argv:
        g7_config2_statistics()
        g13_config5_gradients()
    REPORT['torch'] = torch.__version__
    with open(os.path.join(HERE, 'pinning_report.json'), 'w') as f:
        json.dump(REPORT, f, indent=2, sort_keys=True)
    print(json.dumps(REPORT, indent=2, sort_keys=True))
