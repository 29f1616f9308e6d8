"""Times one full-size training step (BASELINE configs[4] shape on one GPU): train-mode PdsNetwork forward at
960x540 / D=192 -> SubpixelCrossEntropy (HIP) on the cost volume -> backward through the HIP modules."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = pds.PdsNetwork.default(191).to(dev).train()
g = torch.Generator().manual_seed(1)
left = (torch.rand(1, 3, 540, 960, generator=g) * 255).to(dev)
right = (torch.rand(1, 3, 540, 960, generator=g) * 255).to(dev)
gt = (torch.rand(1, 540, 960, generator=g) * 190).to(dev)
gt[:, :16] = float('inf')  # a band without ground truth
criterion = pds.SubpixelCrossEntropy()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cost = net(left, right)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = criterion(cost, gt)
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('step %d: forward %.1f ms, loss+backward %.1f ms, loss %.4f, peak mem %.1f GB' %
          (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, float(loss), torch.cuda.max_memory_allocated() / 2**30))
    net.zero_grad(set_to_none=True)
