"""In-situ duration of the 48-plane conv2d_x3 launches inside sequential pairs of the hot path (launch probe), for A/B
runs of library variants (PDS_HIP_LIB=build/variants/libpds_NAME.so):   python tools/x3_in_situ.py [pairs]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from practicaldeepstereo_nips2018_amd import _lib

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
net, descriptors, images = bench.make_inputs(dev)
reg, est = net._regularization, net._estimator


def run_pair(i):
    ld, rd, sc = descriptors[i % bench.PAIRS]
    return reg.forward_with_estimator(net._matching(ld, rd), sc, est)


with torch.no_grad():
    rec = bench.time_dominant_kernel_in_situ(run_pair, dev, pairs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        run_pair(i)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / 20 * 1e3
print('%-28s x3 in situ %.1f us (min %.1f, max %.1f, %d launches)   sequential pair %.3f ms' % (
    os.path.basename(os.environ.get('PDS_HIP_LIB', 'tree')), rec['launch_ms'] * 1e3, rec['min_ms'] * 1e3, rec['max_ms'] * 1e3,
    rec['launches'], seq))
