import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests import helpers
import practicaldeepstereo_nips2018_amd as pds
dev = torch.device('cuda:0')
g = helpers.golden('g11_training_step')
net = helpers.seeded(lambda: pds.PdsNetwork.default(63)).to(dev).train()
left, right = helpers.images(1, 128, 256)
cost = net(left.to(dev), right.to(dev))
loss = pds.SubpixelCrossEntropy()(cost, g['ground_truth'].to(dev)); loss.backward()
offsets = [int(v) for v in g['grad_sub_offsets']]
rows = []
for i, (name, p) in enumerate(net.named_parameters()):
    if float(g['grad_norms_fp64'][i]) < 1e-9: continue
    flat = p.grad.detach().flatten()
    mine = flat[::max(1, -(-flat.numel() // 512))].double().cpu()
    w64 = g['grad_sub_fp64'][offsets[i]:offsets[i+1]].double(); w32 = g['grad_sub'][offsets[i]:offsets[i+1]].double()
    top = float(w64.abs().max()) + 1e-30
    rows.append((float((mine-w64).abs().max())/top, float((w32-w64).abs().max())/top, float(g['grad_reference_vs_fp64_rel'][i]), name))
rows.sort(reverse=True)
print('PDS_X3', os.environ.get('PDS_X3'), '(honoured only with PDS_DEBUG_SWITCHES=1:', os.environ.get('PDS_DEBUG_SWITCHES'), ')')
for r in rows[:12]: print('%.2e mine  %.2e ref  %.2e ref(full)  %s' % r)
q = [r for r in rows if r[2] < 1e-3]
print('qualifying', len(q), 'worst mine %.2e' % max(r[0] for r in q), 'ratio max %.1f' % max(r[0]/max(r[1],1e-9) for r in q))
