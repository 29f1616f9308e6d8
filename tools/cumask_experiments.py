"""CU-masked streams (hipExtStreamCreateWithCUMask): Matching of pair i on one part of the chip, the latency-bound tail
(Regularization + estimator) of pair i - 1 on the rest, concurrently.   python tools/cumask_experiments.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from practicaldeepstereo_nips2018_amd.distributed import PairStreams

dev = torch.device('cuda:0')
hip = ctypes.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) enabled on the stream"""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


net, descriptors, images = bench.make_inputs(dev)
reg, est, matching = net._regularization, net._estimator, net._matching
STEPS = 40


def timed(run, drain, label):
    with torch.no_grad():
        for i in range(8):
            run(i)
        drain()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(STEPS):
                run(i)
            drain()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / STEPS)
    print('%-78s %.3f ms/pair  %.1f pairs/s' % (label, best * 1e3, 1.0 / best), flush=True)


p = PairStreams(lambda a, b, c: reg.forward_with_estimator(matching(a, b), c, est), streams=3)
timed(lambda i: p.submit(*descriptors[i % bench.PAIRS]), p.drain, 'baseline: whole pairs over 3 streams')


class Split(object):
    def __init__(self, mstreams, tstreams):
        self.ms, self.ts = mstreams, tstreams
        self.n = 0
        self.pending = []

    def submit(self, ld, rd, sc):
        i = self.n
        self.n += 1
        ms, ts = self.ms[i % len(self.ms)], self.ts[i % len(self.ts)]
        if len(self.pending) >= 6:
            self.pending.pop(0).synchronize()
        with torch.cuda.stream(ms):
            sig = matching(ld, rd)
            ready = torch.cuda.Event()
            ready.record(ms)
        ts.wait_event(ready)
        with torch.cuda.stream(ts):
            out = reg.forward_with_estimator(sig, sc, est)
            done = torch.cuda.Event()
            done.record(ts)
        sig.record_stream(ts)
        self.pending.append(done)
        return out

    def drain(self):
        for s in self.ms + self.ts:
            s.synchronize()
        self.pending = []


def contiguous(lo, hi):
    return list(range(lo, hi))


def strided(keep_mod, of):
    return [i for i in range(256) if (i % of) in keep_mod]


configs = [
    ('Matching CUs 0-191, tail CUs 192-255', contiguous(0, 192), contiguous(192, 256)),
    ('Matching CUs 0-223, tail CUs 224-255', contiguous(0, 224), contiguous(224, 256)),
    ('Matching CUs 0-159, tail CUs 160-255', contiguous(0, 160), contiguous(160, 256)),
    ('Matching 3 of every 4 CUs, tail the 4th', strided({0, 1, 2}, 4), strided({3}, 4)),
    ('Matching 6 of every 8 CUs, tail the other 2', strided({0, 1, 2, 3, 4, 5}, 8), strided({6, 7}, 8)),
    ('Matching 7 of every 8 CUs, tail the 8th', strided({0, 1, 2, 3, 4, 5, 6}, 8), strided({7}, 8)),
    ('Matching all CUs, tail CUs 192-255', contiguous(0, 256), contiguous(192, 256)),
]
for label, mbits, tbits in configs:
    for nm, nt in ((1, 1), (2, 2)):
        sp = Split([masked_stream(mbits) for _ in range(nm)], [masked_stream(tbits) for _ in range(nt)])
        timed(lambda i: sp.submit(*descriptors[i % bench.PAIRS]), sp.drain, '%s (%d + %d streams)' % (label, nm, nt))
