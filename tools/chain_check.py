"""The persistent chain kernel of the inner hourglass levels (csrc/conv3d_ks.hip: conv3d_ks_chain_kernel; opt-in:
PDS_DEBUG_SWITCHES=1 PDS_CONV3D_KS_CHAIN=1) against the default per-launch path (one launch + in_finalize per layer),
bit for bit, and a stress run.

    python tools/chain_check.py                 compare both paths at config-1 / config-2 / config-4 shapes (child processes)
    python tools/chain_check.py stress [N]      N (default 1000) Regularization passes over three HIP streams with random
                                                stream skew and a competing kernel: every result must equal the first
    python tools/chain_check.py stamps          per-layer completion times of one chain launch at config 2

The kernel-selection switches are read once per process, so every path runs in a fresh interpreter."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHAIN_ON = {'PDS_DEBUG_SWITCHES': '1', 'PDS_CONV3D_KS_CHAIN': '1'}
if len(sys.argv) > 1 and sys.argv[1] in ('stress', 'stamps'):
    os.environ.update(CHAIN_ON)    # (read once per process by the library: set before it is loaded)

SHAPES = [('config1', 1, 16, 32, 64), ('config2', 1, 48, 144, 240), ('config4_b2', 2, 64, 96, 320),
          ('small_odd', 1, 16, 48, 80)]


def inputs(batch, d, h, w, device):
    import torch
    g = torch.Generator().manual_seed(7)
    ms = torch.randn(batch, 8, d, h, w, generator=g).to(device)
    sc = torch.randn(batch, 8, h, w, generator=g).to(device)
    return ms, sc


def network(device):
    import torch
    import practicaldeepstereo_nips2018_amd as pds
    torch.manual_seed(0)
    return pds.PdsNetwork.default(63).eval().to(device)


def worker():
    import torch
    dev = torch.device('cuda:0')
    net = network(dev)
    out = {}
    with torch.no_grad():
        for name, batch, d, h, w in SHAPES:
            ms, sc = inputs(batch, d, h, w, dev)
            cost = net._regularization(ms, sc)
            disp = net._regularization.forward_with_estimator(ms, sc, net._estimator)
            torch.cuda.synchronize()
            out[name] = [hashlib.sha256(cost.cpu().numpy().tobytes()).hexdigest(),
                         hashlib.sha256(disp.cpu().numpy().tobytes()).hexdigest(), float(cost.double().abs().mean())]
    print('RESULT ' + json.dumps(out))


def stress(iterations):
    import random
    import torch
    dev = torch.device('cuda:0')
    net = network(dev)
    name, batch, d, h, w = SHAPES[1]
    ms, sc = inputs(batch, d, h, w, dev)
    rng = random.Random(5)
    with torch.no_grad():
        expected = net._regularization.forward_with_estimator(ms, sc, net._estimator).clone()
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(dev) for _ in range(3)]
        noise_stream = torch.cuda.Stream(dev)
        a = torch.randn(4096, 4096, device=dev)
        bad = 0
        pending = []
        for i in range(iterations):
            s = streams[i % 3]
            with torch.cuda.stream(s):
                if rng.random() < 0.5:
                    torch.cuda._sleep(rng.randrange(1000, 400000))     # random skew between the streams
                pending.append((i, net._regularization.forward_with_estimator(ms, sc, net._estimator)))
            if rng.random() < 0.3:
                with torch.cuda.stream(noise_stream):                  # a competing kernel that takes CUs away (uneven load)
                    a = (a @ a).clamp_(-1, 1)
            if len(pending) >= 6:
                for st in streams:
                    st.synchronize()
                for j, got in pending:
                    if not torch.equal(got, expected):
                        bad += 1
                        print('MISMATCH at iteration', j, float((got - expected).abs().max()))
                pending = []
        torch.cuda.synchronize()
        for j, got in pending:
            if not torch.equal(got, expected):
                bad += 1
    from practicaldeepstereo_nips2018_amd import _lib
    lib = _lib.load()
    ticks = (ctypes.c_uint * 16)()
    phases = lib.pds_debug_chain_stamps(ticks, 16)   # > 0: the chain kernel really ran
    timeouts = lib.pds_nonfinite_statistics(0)
    print('STRESS iterations %d mismatches %d nonfinite/timeouts %d chain phases %d' % (iterations, bad, timeouts, phases))
    return bad + (1 if timeouts or phases <= 0 else 0)


def stamps():
    import torch
    from practicaldeepstereo_nips2018_amd import _lib
    dev = torch.device('cuda:0')
    net = network(dev)
    name, batch, d, h, w = SHAPES[1]
    ms, sc = inputs(batch, d, h, w, dev)
    lib = _lib.load()
    with torch.no_grad():
        for _ in range(3):
            net._regularization.forward_with_estimator(ms, sc, net._estimator)
        torch.cuda.synchronize()
        for rep in range(3):
            net._regularization.forward_with_estimator(ms, sc, net._estimator)
            ticks = (ctypes.c_uint * 16)()
            n = lib.pds_debug_chain_stamps(ticks, 16)
            t = [ticks[i] / 100.0 for i in range(n)]
            print('STAMPS us since the first ticket:', ' '.join('%.1f' % v for v in t))
            print('       per layer:', ' '.join('%.1f' % (v - (t[i - 1] if i else 0.0)) for i, v in enumerate(t)))


def run_child(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), 'worker'], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode(errors='replace')
    lines = [l for l in text.splitlines() if l.startswith('RESULT ')]
    if out.returncode != 0 or not lines:
        print(text[-3000:])
        raise SystemExit('worker failed')
    return json.loads(lines[-1][7:])


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'compare'
    if mode == 'worker':
        worker()
    elif mode == 'stress':
        raise SystemExit(1 if stress(int(sys.argv[2]) if len(sys.argv) > 2 else 1000) else 0)
    elif mode == 'stamps':
        stamps()
    else:
        chained = run_child(CHAIN_ON)
        plain = run_child({})
        same = True
        for name in chained:
            ok = chained[name][:2] == plain[name][:2]
            same = same and ok
            print('%-12s %s  mean |cost| %.6f / %.6f' % (name, 'identical' if ok else 'DIFFERENT', chained[name][2], plain[name][2]))
        print('IDENTICAL' if same else 'MISMATCH')
        raise SystemExit(0 if same else 1)
