"""Single-GPU schedules for a stream of pairs: whole pairs round-robin over S streams (bench.py's default) against
Matching and the tail on separate streams (tail at high priority).   python tools/sched_experiments.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from practicaldeepstereo_nips2018_amd.distributed import PairStreams

dev = torch.device('cuda:0')
net, descriptors, images = bench.make_inputs(dev)
reg, est, matching = net._regularization, net._estimator, net._matching
STEPS = 40


def whole(ld, rd, sc):
    return reg.forward_with_estimator(matching(ld, rd), sc, est)


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
    print('%-60s %.3f ms/pair  %.1f pairs/s' % (label, best * 1e3, 1.0 / best))


for s in (1, 2, 3, 4):
    p = PairStreams(whole, streams=s)
    timed(lambda i: p.submit(*descriptors[i % bench.PAIRS]), p.drain, 'whole pairs over %d streams' % s)


class Split(object):
    """Matching of pair i on one of M streams, its tail on one of T (high-priority) streams."""

    def __init__(self, m, t, high):
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
        self.ms = [torch.cuda.Stream(dev) for _ in range(m)]
        self.ts = [torch.cuda.Stream(dev, priority=-1 if high else 0) for _ in range(t)]
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


for m, t, high in ((1, 1, False), (1, 1, True), (2, 1, True), (2, 2, True), (2, 2, False), (3, 2, True)):
    sp = Split(m, t, high)
    timed(lambda i: sp.submit(*descriptors[i % bench.PAIRS]), sp.drain,
          'Matching on %d stream(s), tail on %d %s-priority stream(s)' % (m, t, 'high' if high else 'normal'))
