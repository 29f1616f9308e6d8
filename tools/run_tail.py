"""Runs only the Regularization + SubpixelMap tail (eval fusion) a few times at config 2 -- the workload profiled by
tools/prof_tail.sh (rocprofv3 kernel trace / PMC passes of the hourglass kernels).

    python tools/run_tail.py [reps] [--train]     (--train: pds_regularization_fwd, the cost volume is written)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    train = '--train' in sys.argv
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = pds.PdsNetwork.default(191).eval().to(dev)
    g = torch.Generator().manual_seed(1)
    ms = torch.randn(1, 8, 48, 144, 240, generator=g).to(dev)
    sc = torch.randn(1, 8, 144, 240, generator=g).to(dev)
    reg, est = net._regularization, net._estimator
    for a in sys.argv:
        if a.startswith('--window='):   # e.g. --window=2: SubpixelMap(half_support_window=2) -> one tap on either side
            est = pds.SubpixelMap(half_support_window=int(a.split('=')[1]), disparity_step=2)
    with torch.no_grad():
        def step():
            return reg(ms, sc) if train else reg.forward_with_estimator(ms, sc, est)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print('tail (%s): %.3f ms per pair, checksum %.6f' % ('train' if train else 'eval', dt * 1e3,
                                                        float(out.double().mean())))


if __name__ == '__main__':
    main()
