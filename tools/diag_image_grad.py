"""Diagnostic (not a test): where does the noise of the whole-network image gradient come from?
Gradients reaching the descriptors / shortcut (Matching + Regularization + loss backward) on the HIP path against the
fp64 oracle, with the fp32 CPU oracle as the noise floor."""
import sys, torch
sys.path.insert(0, '.')
from oracle import pds_oracle as oracle
from tests import helpers
import practicaldeepstereo_nips2018_amd as pds
dev = torch.device('cuda:0')
H, W = 128, 192
net = helpers.seeded(lambda: pds.PdsNetwork.default(63)).to(dev).train()
params = {k: v.detach().cpu() for k, v in net.state_dict().items()}
left, right = helpers.images(1, H, W)
gt = torch.rand(1, H, W, generator=torch.Generator().manual_seed(6)) * 60
gt[:, :8] = float('inf')
with torch.no_grad():
    ld0, sc0 = oracle.embedding(params, '_embedding', left)
    rd0 = oracle.embedding(params, '_embedding', right)[0]

def cpu(dtype):
    p = oracle.cast_params(params, dtype)
    t = [x.clone().to(dtype).requires_grad_(True) for x in (ld0, rd0, sc0)]
    ms = oracle.matching_with_operation(p, '_matching', t[0], t[1], 15)
    ms.retain_grad()
    cost = oracle.regularization(p, '_regularization', ms, t[2])
    oracle.subpixel_cross_entropy(cost, gt.to(dtype)).backward()
    return [x.grad for x in t] + [ms.grad]

g64, g32 = cpu(torch.float64), cpu(torch.float32)
t = [x.clone().to(dev).requires_grad_(True) for x in (ld0, rd0, sc0)]
ms = net._matching(t[0], t[1])
ms.retain_grad()
cost = net._regularization(ms, t[2])
pds.SubpixelCrossEntropy()(cost, gt.to(dev)).backward()
got = [x.grad for x in t] + [ms.grad]
for name, a, b, c in zip(('d left descriptor', 'd right descriptor', 'd shortcut', 'd signatures'), got, g64, g32):
    e = (a.double().cpu() - b).abs()
    f = (c.double() - b).abs()
    print('%-20s HIP: max %.3g mean %.3g | fp32 CPU: max %.3g mean %.3g | max |want| %.3g mean |want| %.3g' % (
        name, float(e.max()), float(e.mean()), float(f.max()), float(f.mean()), float(b.abs().max()), float(b.abs().mean())))

# the descriptor network alone, driven by the REALISTIC upstream gradients above (fp64 values), not random ones
def emb_cpu(dtype):
    p = oracle.cast_params(params, dtype)
    leaf = left.clone().to(dtype).requires_grad_(True)
    d, s = oracle.embedding(p, '_embedding', leaf)
    ((d * g64[0].to(dtype)).sum() + (s * g64[2].to(dtype)).sum()).backward()
    return leaf.grad
e64, e32 = emb_cpu(torch.float64), emb_cpu(torch.float32)
leaf = left.clone().to(dev).requires_grad_(True)
d, s = net._embedding(leaf)
((d * g64[0].float().to(dev)).sum() + (s * g64[2].float().to(dev)).sum()).backward()
e = (leaf.grad.double().cpu() - e64).abs()
f = (e32.double() - e64).abs()
print('embedding alone, realistic upstream: HIP max %.3g mean %.3g | fp32 CPU max %.3g mean %.3g | max |want| %.3g mean %.3g' % (
    float(e.max()), float(e.mean()), float(f.max()), float(f.mean()), float(e64.abs().max()), float(e64.abs().mean())))
# per-layer: gradient reaching the first InstanceNorm's output, i.e. before the image head
