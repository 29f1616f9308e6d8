"""Diagnostic (not a test): stage-wise distance of the HIP path (train-mode route and eval route) and of the fp32 CPU
oracle from the fp64 oracle -- where does the forward noise that the image gradient amplifies come from?"""
import sys, torch
sys.path.insert(0, '.')
from oracle import pds_oracle as oracle
from tests import helpers
import practicaldeepstereo_nips2018_amd as pds
dev = torch.device('cuda:0')
H, W = 128, 192
net = helpers.seeded(lambda: pds.PdsNetwork.default(63)).to(dev)
params = {k: v.detach().cpu() for k, v in net.state_dict().items()}
left, right = helpers.images(1, H, W)

def cpu(dtype):
    p = oracle.cast_params(params, dtype)
    with torch.no_grad():
        ld, sc = oracle.embedding(p, '_embedding', left.to(dtype))
        rd = oracle.embedding(p, '_embedding', right.to(dtype))[0]
        ms = oracle.matching_with_operation(p, '_matching', ld, rd, 15)
        cost = oracle.regularization(p, '_regularization', ms, sc)
    return ld, sc, ms, cost

r64, r32 = cpu(torch.float64), cpu(torch.float32)

def hip(train):
    net.train(train)
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        l = left.to(dev).requires_grad_(train)
        ld, sc = net._embedding(l)
        rd = net._embedding(right.to(dev))[0]
        ms = net._matching(ld, rd)
        cost = net._regularization(ms, sc)
    return [t.detach() for t in (ld, sc, ms, cost)]

for label, got in (('fp32 CPU oracle', r32), ('HIP eval route', hip(False)), ('HIP train route', hip(True))):
    print(label)
    for name, a, b in zip(('descriptor', 'shortcut', 'signatures', 'cost'), got, r64):
        e = (a.double().cpu() - b).abs()
        print('   %-11s max %.3g mean %.3g   (|ref| max %.3g)' % (name, float(e.max()), float(e.mean()), float(b.abs().max())))
