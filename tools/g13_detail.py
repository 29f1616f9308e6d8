import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import helpers
import practicaldeepstereo_nips2018_amd as pds
g = dict(np.load('tests/golden/g13_config5_gradients.npz'))
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = pds.PdsNetwork.default(191)
names = [n for n, _ in net.named_parameters()]
left, right = helpers.images(1, 540, 960)
gen = torch.Generator().manual_seed(41)
gt = torch.rand(1, 540, 960, generator=gen) * 190.0
gt[:, :17] = float('inf')
net = net.to(dev).train()
pds.SubpixelCrossEntropy()(net(left.to(dev), right.to(dev)), gt.to(dev)).backward()
off = g['grad_sub_offsets']
for key in ('_regularization._contraction_blocks.2._smoothing.0.bias', '_regularization._contraction_blocks.2._smoothing.2.bias', '_regularization._contraction_blocks.2._smoothing.2.weight'):
    i = names.index(key)
    p = dict(net.named_parameters())[key]
    got = p.grad.flatten().double().cpu().numpy()
    want = g['grad_sub_fp64'][off[i]:off[i + 1]]
    ref32 = g['grad_sub'][off[i]:off[i + 1]].astype(np.float64)
    e = (got - want) / np.abs(want).max()
    r = (ref32 - want) / np.abs(want).max()
    print(key, 'max|want| %.3e' % np.abs(want).max())
    print('  gpu err  :', np.array2string(e[:24], precision=4, suppress_small=True, max_line_width=220))
    print('  ref32 err:', np.array2string(r[:24], precision=4, suppress_small=True, max_line_width=220))
    print('  corr(gpu err, want) %.3f  mean gpu err %.3e  std %.3e ; mean ref err %.3e std %.3e' % (np.corrcoef(e, want)[0, 1], e.mean(), e.std(), r.mean(), r.std()))
