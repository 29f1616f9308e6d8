"""Diagnostic (GPU box): stage-wise error of the GPU path and of the fp32 CPU oracle against an fp64
run of the oracle on the same config-2 inputs (the arbiter of SURVEY.md 8c)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds
from oracle import pds_oracle as oracle

def rep(a, b):
    d = (a.double().cpu() - b.double().cpu()).abs()
    return {'max': float(d.max()), 'mean': float(d.mean())}

def drep(a, b):
    d = (a.double().cpu() - b.double().cpu()).abs()
    return {'mae': float(d.mean()), 'flips': int((d > 0.5).sum()), 'mae_noflip': float(d[d <= 0.5].mean())}

torch.set_num_threads(32)
H, W, MD = (540, 960, 191) if len(sys.argv) < 2 else (128, 256, 63)
torch.manual_seed(0)
net = pds.PdsNetwork.default(MD).eval()
g = torch.Generator().manual_seed(1)
left = torch.rand(1, 3, H, W, generator=g) * 255
right = torch.rand(1, 3, H, W, generator=g) * 255
with torch.no_grad():
    p32 = {k: v.clone() for k, v in net.state_dict().items()}
    ld, sc = oracle.embedding(p32, '_embedding', oracle.pad_to_multiple(left)[0])
    rd = oracle.embedding(p32, '_embedding', oracle.pad_to_multiple(right)[0])[0]
    p64 = oracle.cast_params(p32, torch.float64)
    t = time.time(); ms32, c32, d32 = oracle.hot_path(p32, ld, rd, sc, MD, return_stages=True); t32 = time.time() - t
    t = time.time(); ms64, c64, d64 = oracle.hot_path(p64, ld.double(), rd.double(), sc.double(), MD, return_stages=True); t64 = time.time() - t
    dev = torch.device('cuda:0')
    net = net.to(dev)
    msg = net._matching(ld.to(dev), rd.to(dev))
    cg = net._regularization(msg, sc.to(dev))
    dg = net._estimator(cg)
    # hybrids: GPU matching -> CPU fp32 regularization ; CPU matching -> GPU regularization
    c_h1 = oracle.regularization(p32, '_regularization', msg.cpu(), sc)
    c_h2 = net._regularization(ms32.to(dev), sc.to(dev))
    out = {
      'cpu_seconds_fp32': t32, 'cpu_seconds_fp64': t64,
      'signatures': {'cpu32_vs_64': rep(ms32, ms64), 'gpu_vs_64': rep(msg, ms64), 'gpu_vs_cpu32': rep(msg, ms32)},
      'cost': {'cpu32_vs_64': rep(c32, c64), 'gpu_vs_64': rep(cg, c64), 'gpu_vs_cpu32': rep(cg, c32),
               'gpuM_cpuR_vs_64': rep(c_h1, c64), 'cpuM_gpuR_vs_64': rep(c_h2, c64)},
      'disparity': {'cpu32_vs_64': drep(d32, d64), 'gpu_vs_64': drep(dg, d64), 'gpu_vs_cpu32': drep(dg, d32),
                    'gpuM_cpuR_vs_64': drep(oracle.subpixel_map(c_h1), d64),
                    'cpuM_gpuR_vs_64': drep(net._estimator(c_h2), d64)},
    }
print(json.dumps(out, indent=1))
