"""Debug aid: one Regularization pass at a small shape through the chain kernel; while it runs (or hangs) the synchronisation
words -- and, in a -DPDS_KS_CHAIN_MARK build, every workgroup's progress word -- are read from a side stream."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import practicaldeepstereo_nips2018_amd as pds
from practicaldeepstereo_nips2018_amd import _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = pds.PdsNetwork.default(63).eval().to(dev)
shape = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [1, 16, 32, 64]
g = torch.Generator().manual_seed(7)
ms = torch.randn(shape[0], 8, shape[1], shape[2], shape[3], generator=g).to(dev)
sc = torch.randn(shape[0], 8, shape[2], shape[3], generator=g).to(dev)
lib = _lib.load()
captured = []
_original_buffer = _lib.Workspace._buffer


def _capturing_buffer(self, nbytes, device):
    slot, buf = _original_buffer(self, nbytes, device)
    captured.append(buf)
    return slot, buf


_lib.Workspace._buffer = _capturing_buffer
t0 = time.time()
with torch.no_grad():
    cost = net._regularization.forward_with_estimator(ms, sc, net._estimator)
side = torch.cuda.Stream(dev)


def peek(label):
    with torch.cuda.stream(side):
        words = captured[-1][:8192].clone().cpu().view(torch.int32)
    side.synchronize()
    n = int(words[64])
    first = words[65:65 + 12].tolist()
    print(label, 'phases', n, 'first tickets', first[:n + 1])
    print('  head', int(words[0]), 'done', words[16:16 + n].tolist(), 'ready', words[32:32 + n].tolist())
    marks = words[1024:1024 + 256].tolist()
    hist = {}
    for m in marks:
        hist[(m & 255)] = hist.get(m & 255, 0) + 1
    print('  stage histogram (stage: workgroups)', {hex(k): v for k, v in sorted(hist.items())})
    print('  sample marks (ticket, stage):', [(m >> 8, hex(m & 255)) for m in marks[:24]])
    sys.stdout.flush()


for k in range(3):
    time.sleep(2.0)
    peek('after %.0f s:' % (time.time() - t0))
if '--nowait' in sys.argv:
    os._exit(0)
torch.cuda.synchronize()
print('pass took %.2f s, |cost| mean %.6f, nonfinite/timeouts %d' % (time.time() - t0, float(cost.abs().mean()), lib.pds_nonfinite_statistics(0)))
