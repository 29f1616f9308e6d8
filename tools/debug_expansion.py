"""ExpansionBlock3d on the GPU vs the oracle (debugging aid for the transposed-convolution kernels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds
from practicaldeepstereo_nips2018_amd import regularization
from oracle import pds_oracle as oracle

dev = torch.device('cuda:0')
for feats, shape in [(8, (1, 8, 6, 8, 20)), (16, (1, 16, 4, 6, 36)), (8, (2, 8, 16, 16, 32)), (16, (1, 16, 8, 8, 16))]:
    torch.manual_seed(0)
    block = regularization.ExpansionBlock3d(feats)
    x = torch.randn(*shape)
    sc = torch.randn(shape[0], feats // 2, 2 * shape[2], 2 * shape[3], 2 * shape[4])
    params = {'b.' + k: v.detach().clone() for k, v in block.state_dict().items()}
    with torch.no_grad():
        ref = oracle.expansion_block_3d(params, 'b', x, sc)
        out = block.to(dev)(x.to(dev), sc.to(dev)).cpu()
    d = (out - ref).abs()
    print(feats, shape, 'max diff %.3e' % d.max(), 'at', [int(i) for i in (d == d.max()).nonzero()[0]])
