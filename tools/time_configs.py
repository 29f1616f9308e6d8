"""Times the hot path (Matching -> Regularization + SubpixelMap, eval) at the BASELINE.json shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds

dev = torch.device('cuda:0')
for name, batch, h, w, maxd in [('config1 128x256 D=64', 1, 32, 64, 63), ('config2 540x960 D=192', 1, 144, 240, 191),
                                ('config4 375x1242 D=256 B=4', 4, 96, 320, 255)]:
    torch.manual_seed(0)
    net = pds.PdsNetwork.default(maxd).eval().to(dev)
    g = torch.Generator().manual_seed(1)
    ld = torch.randn(batch, 64, h, w, generator=g).to(dev)
    rd = torch.randn(batch, 64, h, w, generator=g).to(dev)
    sc = torch.randn(batch, 8, h, w, generator=g).to(dev)
    with torch.no_grad():
        def step():
            return net._regularization.forward_with_estimator(net._matching(ld, rd), sc, net._estimator)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        tm = time.perf_counter()
        for _ in range(n):
            net._matching(ld, rd)
        torch.cuda.synchronize()
        dm = (time.perf_counter() - tm) / n
    print('%-30s %.2f ms per batch (%.1f pairs/s), matching %.2f ms, peak mem %.1f GB' %
          (name, dt * 1e3, batch / dt, dm * 1e3, torch.cuda.max_memory_allocated() / 2**30))
