"""Per-rank Matching time of the disparity-sharded schedule on ONE GPU: the shard of rank 0 for N = 1, 2, 4, 8 at config 2
(the other ranks do the same work on their planes), and the unshardable tail.  Usage: python tools/time_shards.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = pds.PdsNetwork.default(191).eval().to(dev).freeze_weights()
g = torch.Generator().manual_seed(1)
left = (torch.rand(1, 3, 540, 960, generator=g) * 255).to(dev)
right = (torch.rand(1, 3, 540, 960, generator=g) * 255).to(dev)
with torch.no_grad():
    ld, sc = net._embedding(net._size_adapter.pad(left))
    rd = net._embedding(net._size_adapter.pad(right))[0]

def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

with torch.no_grad():
    ms = net._matching(ld, rd)
    tail = timed(lambda: net._regularization.forward_with_estimator(ms, sc, net._estimator))
    print('tail (Regularization + estimator): %.3f ms' % tail)
    for n in (1, 2, 4, 8):
        net._matching.set_disparity_shard((0, 48 // n))
        t = timed(lambda: net._matching(ld, rd))
        print('N = %d: Matching of one rank (%d planes): %.3f ms   -> ideal pair time with the tail dealt round-robin: %.3f ms'
              % (n, 48 // n, t, t + tail / n))
    net._matching.set_disparity_shard(None)
