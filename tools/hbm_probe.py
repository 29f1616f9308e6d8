"""Calibration of streaming rates on this chip with plain torch kernels: write-only (fill), read+write (copy),
read-only (sum) over a 425 MB fp32 tensor (the size of one 64-channel activation volume at config 2)."""
import torch
dev = torch.device('cuda:0')
x = torch.randn(1, 64, 48, 144, 240, device=dev)
y = torch.empty_like(x)
mb = x.numel() * 4 / 1e6


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


t = timeit(lambda: y.fill_(1.0)); print('fill  %.3f ms  %.2f TB/s written' % (t, mb / t / 1e3))
t = timeit(lambda: y.copy_(x)); print('copy  %.3f ms  %.2f TB/s read+written' % (t, 2 * mb / t / 1e3))
t = timeit(lambda: x.sum()); print('sum   %.3f ms  %.2f TB/s read' % (t, mb / t / 1e3))
t = timeit(lambda: torch.add(x, 1.0, out=y)); print('add   %.3f ms  %.2f TB/s read+written' % (t, 2 * mb / t / 1e3))
