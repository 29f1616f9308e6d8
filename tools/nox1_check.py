"""The fused Matching chain without a materialised first residual sum (PDS_MATCHING_NOX1, api_matching.hip) against the
default chain: the 64 -> 8 layer receives the share of x0 = A + shift_d(G) factorised (exact fp32) instead of inside the
fp16-split products, so the signatures agree to rounding, not bit for bit.  Prints the largest absolute difference per case
(whole range and disparity shards with d_begin > 0) and fails above 2e-5 (the parity tolerance of the fp16-split kernels).

    python tools/nox1_check.py [levels ...]     (default: 0 1; one process per level, the switch is read once per process)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'nox1')
CHILD = r'''
import sys, numpy as np, torch, time
sys.path.insert(0, %r)
import practicaldeepstereo_nips2018_amd as pds
dev = torch.device('cuda:0')
k = 0
for (maxd, b, h, w, shard) in ((191, 1, 144, 240, None), (63, 2, 32, 64, None), (255, 1, 96, 320, None),
                               (191, 1, 144, 240, (24, 24)), (63, 1, 32, 64, (3, 5)), (63, 1, 16, 128, (0, 1))):
    torch.manual_seed(0)
    m = pds.Matching((maxd + 1) // 4 - 1, pds.MatchingOperation()).to(dev).eval()
    m.set_disparity_shard(shard)
    g = torch.Generator().manual_seed(1)
    l = torch.randn(b, 64, h, w, generator=g).to(dev)
    r = torch.randn(b, 64, h, w, generator=g).to(dev)
    with torch.no_grad():
        out = m(l, r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = m(l, r)
        torch.cuda.synchronize()
    print('CASE', k, maxd, b, h, w, shard, '%%.3f ms' %% ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
    np.save(sys.argv[1] + '_%%d.npy' %% k, out.cpu().numpy())
    k += 1
''' % ROOT


def main():
    os.makedirs(OUT, exist_ok=True)
    levels = sys.argv[1:] or ['0', '1']
    for level in levels:
        env = dict(os.environ, PDS_DEBUG_SWITCHES='1', PDS_MATCHING_NOX1=level)
        out = subprocess.run([sys.executable, '-c', CHILD, os.path.join(OUT, 'l' + level)], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT)
        text = out.stdout.decode(errors='replace')
        print('--- level', level)
        print(text[-3000:])
        if out.returncode != 0:
            raise SystemExit('level %s failed' % level)
    worst = 0.0
    for level in levels[1:]:
        k = 0
        while os.path.exists(os.path.join(OUT, 'l%s_%d.npy' % (levels[0], k))):
            a = np.load(os.path.join(OUT, 'l%s_%d.npy' % (levels[0], k))).astype(np.float64)
            b = np.load(os.path.join(OUT, 'l%s_%d.npy' % (level, k))).astype(np.float64)
            d = np.abs(a - b)
            at = np.unravel_index(int(d.argmax()), d.shape)
            print('level %s vs %s case %d: max |diff| %.3e at %s (|ref| max %.3f), mean %.3e' %
                  (level, levels[0], k, d.max(), at, np.abs(a).max(), d.mean()))
            worst = max(worst, float(d.max()))
            k += 1
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    print('WORST', worst)
    raise SystemExit(0 if worst <= 2e-5 else 1)


if __name__ == '__main__':
    main()
