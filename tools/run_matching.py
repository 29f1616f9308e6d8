"""Runs only Matching (fused inference chain) a few times at config 2 -- the workload of kernel-trace A/B runs of the
Matching kernels (tools/time_variants.sh with RUNNER=tools/run_matching.py).    python tools/run_matching.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = pds.PdsNetwork.default(191).eval().to(dev)
g = torch.Generator().manual_seed(1)
ld = torch.randn(1, 64, 144, 240, generator=g).to(dev)
rd = torch.randn(1, 64, 144, 240, generator=g).to(dev)
with torch.no_grad():
    for _ in range(2):
        net._matching(ld, rd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = net._matching(ld, rd)
    torch.cuda.synchronize()
print('matching: %.3f ms per pair, checksum %.6f' % ((time.perf_counter() - t0) / reps * 1e3, float(out.double().abs().mean())))
