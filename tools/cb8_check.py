"""Bit-exactness of the channel-blocked activation layout of the fused Matching chain (PDS_MATCHING_CB8, api.hip) against
the planar one: the same products in the same order, so the signatures must be identical bit for bit.

    python tools/cb8_check.py            (spawns one process per level: the switch is read once per process)
    python tools/cb8_check.py 0 1 2 1,PDS_X3_QUADS=0     (a level may carry further debug switches)
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
import practicaldeepstereo_nips2018_amd as pds
dev = torch.device('cuda:0')
for (maxd, b, h, w) in ((191, 1, 144, 240), (63, 2, 32, 64), (255, 1, 96, 320)):
    torch.manual_seed(0)
    m = pds.Matching((maxd + 1) // 4 - 1, pds.MatchingOperation()).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    l = torch.randn(b, 64, h, w, generator=g).to(dev)
    r = torch.randn(b, 64, h, w, generator=g).to(dev)
    with torch.no_grad():
        out = m(l, r)
    torch.cuda.synchronize()
    print('HASH', maxd, b, h, w, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest(), float(out.double().abs().mean()))
''' % ROOT


def main():
    results = {}
    for spec in sys.argv[1:] or ['0', '1']:
        level, *extra = spec.split(',')
        env = dict(os.environ, PDS_DEBUG_SWITCHES='1', PDS_MATCHING_CB8=level)
        env.update(kv.split('=', 1) for kv in extra)
        level = spec
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        text = out.stdout.decode(errors='replace')
        lines = [t for t in text.splitlines() if t.startswith('HASH')]
        if out.returncode != 0 or not lines:
            print(text[-3000:])
            raise SystemExit('level %s failed' % level)
        results[level] = lines
        for t in lines:
            print('level', level, t)
    levels = sorted(results)
    same = all(results[levels[0]] == results[k] for k in levels[1:])
    print('IDENTICAL' if same else 'DIFFERENT')
    raise SystemExit(0 if same else 1)


if __name__ == '__main__':
    main()
