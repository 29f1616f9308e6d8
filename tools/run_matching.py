"""Runs the fused Matching (inference) a few times at config 2 -- the workload of the Matching-side debug builds
(e.g. PDS_HIP_LIB=build/variants/libpds_c2w_TIMING.so).   python tools/run_matching.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = pds.Matching(47, pds.MatchingOperation()).to(dev).eval()
g = torch.Generator().manual_seed(1)
left = torch.randn(1, 64, 144, 240, generator=g).to(dev)
right = torch.randn(1, 64, 144, 240, generator=g).to(dev)
with torch.no_grad():
    m(left, right)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = m(left, right)
    torch.cuda.synchronize()
print('matching: %.3f ms per pair, checksum %.6f' % ((time.perf_counter() - t0) / reps * 1e3, float(out.double().mean())))
