"""Micro-benchmark of the stand-alone SubpixelMap kernel on a [1, 96, 576, 960] volume (the eval-mode cost volume of
BASELINE configs[1]); usage: python tools/bench_estimator.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
torch.manual_seed(0)
vol = torch.randn(1, 96, 576, 960, device=dev)
est = pds.SubpixelMap()
for _ in range(3):
    est(vol)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); est(vol); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
nbytes = vol.numel() * 4 + 576 * 960 * 4
print('subpixel_map: min %.1f us  median %.1f us  -> %.2f TB/s (median)' % (ts[0], ts[len(ts) // 2], nbytes / ts[len(ts) // 2] / 1e6))
