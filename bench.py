"""Benchmark of the Practical Deep Stereo cost-volume hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Pairs arrive as a stream: at N = 1 whole pairs are dealt round-robin to three HIP streams, so the HBM- and
latency-bound phases of one pair (factorised first layers, streaming kernels, the small 3-D layers) run beside the
MFMA-bound kernels of another; "value" is that throughput, "ms_per_frame" / "sequential" the un-overlapped latency of
one pair (--no-pipeline times that mode as value), and every pipelined result is checked bit for bit against the
sequential one.

A step = one stereo pair through Matching -> Regularization -> SubpixelMap (eval mode) at
BASELINE.json configs[1]: 960x540 (padded 576x960), D=192 (maximum_disparity 191), fp32, random-init
weights (seed 0), descriptors of seeded uniform images (SURVEY.md 8c recipe) resident in HBM before
the timed region.  N > 1 follows configs[2]: the disparity axis of Matching is sharded over the
ranks and one all-gather (RCCL) per pair reassembles the signatures on every rank; Regularization + the
estimator cannot be sharded, so the tail of pair i runs on rank i % N (side stream) while all ranks match
pair i + 1 -- not N redundant copies.  The total work is one pair per step, so scaling is "strong";
"latency_mode" (one pair at a time, tail replicated) and "replica_mode" (independent pairs, no collective)
are reported beside it.  Rank 0 prints ONE JSON line.

The line also carries
  roofline     - the dominant kernel (conv2d 3x3 64->64 over all disparity planes): its launch is timed in
                 isolation with HIP events through pds_conv_block_chained_fwd (input behind a deferred InstanceNorm,
                 as in the hot path); achieved = EXECUTED matrix flops per launch (3 fp16 flops per algorithmic one:
                 two-way split operands, three partial products) / mean duration against the dense 16-bit MFMA peak;
                 the algorithmic fp32 figures ride along.
  path_roofline- the whole hot path at ms_per_frame against SURVEY.md 8d's counts.
  gpu_baseline - the same restatement on the MI355X through PyTorch-ROCm / MIOpen (no hand-written kernels).
  cpu_baseline - the oracle (oracle/pds_oracle.py, PyTorch-CPU restatement of the reference) on the
                 same inputs on this host's cores (rank 0, N=1 only), and the parity of the GPU result
                 against it.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import practicaldeepstereo_nips2018_amd as pds  # noqa: E402
from practicaldeepstereo_nips2018_amd import _lib  # noqa: E402
from practicaldeepstereo_nips2018_amd.distributed import (PairStreams, ShardedHotPath, ShardedMatching,  # noqa: E402
                                                          gather_description)

HEIGHT, WIDTH, MAX_DISPARITY = 540, 960, 191
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak of the same table
# dense 2*MAC count of one 64->64 3x3 convolution over [1, 64, 48, 144, 240] (122.31 GF)
CONV64_GFLOP = 2.0 * 48 * 144 * 240 * 64 * 64 * 9 / 1e9
# HBM bytes per launch of that kernel come from the PMC record committed with the round's profiles (rocprofv3 must
# wrap the process to collect counters, so this script cannot re-collect them): tools/collect_profiles.sh writes the
# counters, profiles/r03_conv64_pmc.json holds FETCH_SIZE / WRITE_SIZE of this kernel and the guide's gfx950 correction
ROOT = os.path.dirname(os.path.abspath(__file__))
CONV64_SOURCE = os.path.join(ROOT, 'practicaldeepstereo_nips2018_amd', 'csrc', 'conv2d_x3.hip')


def conv64_pmc_record():
    """The newest committed counter record of the dominant kernel (profiles/rNN_conv64_pmc.json)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_conv64_pmc.json')))
    return found[-1] if found else os.path.join(ROOT, 'profiles', 'r04_conv64_pmc.json')


def source_sha256(path):
    import hashlib
    try:
        with open(path, 'rb') as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def conv64_hbm_traffic():
    """(bytes per launch, description, stale) from the committed PMC record, or (None, reason, None) when it is absent.
    `stale`: the record carries the sha256 of the kernel source it was measured on (tools/install_profiles.py); when the
    source has changed since, the figure describes an older kernel and the line says so."""
    path = conv64_pmc_record()
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError) as e:
        return None, 'no PMC record: %s' % e, None
    nbytes = (rec['FETCH_SIZE_KB'] * rec['fetch_correction'] + rec['WRITE_SIZE_KB']) * 1e3
    recorded = rec.get('kernel_source_sha256')
    stale = (recorded != source_sha256(CONV64_SOURCE)) if recorded else True
    name = os.path.basename(path)
    return nbytes, ('rocprofv3 FETCH_SIZE %.1f MB x %g (gfx950 wide-read correction) + WRITE_SIZE %.1f MB, separate '
                    '--pmc passes, profiles/%s <- profiles/%s (algorithmic %.1f MB)'
                    % (rec['FETCH_SIZE_KB'] / 1e3, rec['fetch_correction'], rec['WRITE_SIZE_KB'] / 1e3, name,
                       name.replace('conv64_pmc.json', 'pmc_all_kernels.txt'), rec['algorithmic_bytes'] / 1e6)), stale


# Round 3: the layer runs on the 16-bit matrix pipe (csrc/conv2d_x3.hip): every fp32 operand is split into two fp16
# parts (pre-scaled by exact powers of two) and three of the four partial products are accumulated in fp32 -- more
# accurate than the fp32 fmaf chain (measured, tools/ubench/fp16x2_probe.hip).  EXECUTED work = 3 fp16 MFMA flops per
# algorithmic flop, priced against the dense fp16 MFMA peak (== the bf16 one).  PDS_X3_FP16=0 selects the range-safe
# bf16 form (three parts, six products), PDS_X3=0 the exact-fp32 Winograd kernel of round 2 (2/3 of the flops on the
# fp32 MFMA pipe).
# (the library honours its kernel-selection variables only with PDS_DEBUG_SWITCHES=1, csrc/common.hpp: debug_switch;
# without the gate the default kernels run whatever PDS_X3 says, and this script prices them as such)
SWITCHES_ARMED = os.environ.get('PDS_DEBUG_SWITCHES', '')[:1] == '1'


def debug_switch(name, default='1'):
    return os.environ.get(name, default) if SWITCHES_ARMED else default


X3 = debug_switch('PDS_X3')[:1] != '0'
X3_PRODUCTS = 3.0 if debug_switch('PDS_X3_FP16')[:1] != '0' else 6.0
CONV64_EXECUTED_GFLOP = CONV64_GFLOP * X3_PRODUCTS if X3 else CONV64_GFLOP * (2.0 / 3.0)
CONV64_EXECUTED_PEAK = BF16_MFMA_PEAK_TFLOPS if X3 else FP32_MFMA_PEAK_TFLOPS
# algorithmic work of the whole hot path per pair at configs[1]
PATH_GFLOP_REFERENCE = 791.9   # as the reference executes it (dense first layer)
PATH_GFLOP_MINIMAL = 552.4     # with the exact layer-0 factorisation
PATH_ALGORITHMIC_MB = 3900.0 + 1148.7 + 214.5   # Matching (factorised) + Regularization + estimator activations


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # (round 6: 400 steps by default -- a timed region of about one second; the 20-step regions of rounds 1-5 lasted 50 ms,
    # of which the fill and drain of the three-stream schedule are ~3 %, and the dominant kernel is power-limited)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train-record', action='store_true',
                    help='N = 1: skip the short config-5 (training step) record that rides on the default line')
    ap.add_argument('--streams', type=int, default=3, help='N = 1: HIP streams the pairs are dealt to (round-robin)')
    ap.add_argument('--sharded-streams', type=int, default=2,
                    help='N > 1: HIP streams the (sharded) pairs are dealt to on every rank; 1 = main stream + tail stream')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='N = 1: one pair strictly after the other on one stream (latency mode) instead of dealing whole '
                         'pairs round-robin to --streams HIP streams')
    ap.add_argument('--kernel-reps', type=int, default=10)
    ap.add_argument('--windows', type=int, default=5,
                    help='timed windows of --steps steps each; the FIRST is "value", the others give the median / spread')
    ap.add_argument('--graph', action='store_true',
                    help='replay the hot path as one captured HIP graph (N = 1); measured equal to eager launches '
                         'because the GPU is saturated, so eager is the default')
    ap.add_argument('--train', action='store_true',
                    help='BASELINE.json configs[4] instead of the inference hot path: one full-size training step per '
                         'step (train-mode forward, SubpixelCrossEntropy, backward, RMSprop), one pair per GPU, '
                         'DistributedDataParallel over RCCL for N > 1 (weak scaling)')
    ap.add_argument('--sustained-seconds', type=float, default=3.0,
                    help='N = 1: length of the ONE contiguous timed window of the "sustained" sub-record (0: skip)')
    ap.add_argument('--no-sub-records', action='store_true',
                    help='skip the sub-records that ride on the default N = 1 line (exact_fp32 child run, sustained window, '
                         'configs[3]); the exact_fp32 child itself runs with this flag')
    ap.add_argument('--backend', default='nccl', help='process-group backend for N > 1 ("nccl" is RCCL)')
    ap.add_argument('--share-device', action='store_true',
                    help='functional test only: every rank uses cuda:0 (needs --backend gloo)')
    return ap.parse_args()


ARITHMETIC = (
    ('fp32 storage and fp32 accumulation everywhere; the 64-channel 3x3 convolutions (3 of 6 layers of MatchingOperation, '
     'the 64->8 layer, the 64-channel layers of the embedding) multiply on the 16-bit matrix pipe: every fp32 operand is '
     'split into %s scaled by a power of two derived from the data, %d exact partial products per multiply accumulated in '
     'fp32 (22 of 24 significand bits per operand; measured mean error below the fp32 fmaf chain\'s); all other layers '
     'are exact fp32 MFMA / VALU fma' % (('two fp16 parts', 3) if X3_PRODUCTS == 3.0 else ('three bf16 parts', 6)))
    if X3 else 'IEEE fp32 multiplies and fp32 accumulation everywhere (exact-fp32 MFMA, Winograd F(2,3) in fp32)')

PAIRS = 4   # distinct stereo pairs the timed steps rotate through


def make_inputs(device):
    """Seed-0 default network; PAIRS image pairs, seeds 1, 2, ... (pair 0 is the SURVEY.md 8c recipe).  The descriptor
    network (the producer of the hot path's inputs, itself on the HIP library) runs once per pair, untimed; the CPU
    baseline later receives host copies of the very same descriptors, so both sides see bit-identical inputs.
    Returns the network, [(left descriptor, right descriptor, left shortcut)] and [(left image, right image)]."""
    torch.manual_seed(0)
    net = pds.PdsNetwork.default(MAX_DISPARITY).eval().to(device).freeze_weights()   # inference deployment: weights stay packed
    descriptors, images = [], []
    for i in range(PAIRS):
        g = torch.Generator().manual_seed(1 + i)
        left = (torch.rand(1, 3, HEIGHT, WIDTH, generator=g) * 255).to(device)
        right = (torch.rand(1, 3, HEIGHT, WIDTH, generator=g) * 255).to(device)
        with torch.no_grad():
            ld, shortcut = net._embedding(net._size_adapter.pad(left))
            rd = net._embedding(net._size_adapter.pad(right))[0]
        descriptors.append((ld, rd, shortcut))
        images.append((left, right))
    return net, descriptors, images


def time_dominant_kernel(net, device, reps):
    """Mean duration (ms) of one conv2d 64->64 block launch over all 48 planes, HIP events on the
    stream the kernel is launched on (torch's current stream)."""
    lib = _lib.load()
    block = net._matching._operation._matching_operation_modules[1].convolutions[0]
    params = _lib.conv_block_params(block.conv, block.norm)
    n, c, d, h, w = 1, 64, (MAX_DISPARITY + 1) // 4, 144, 240
    # as in the hot path: the input is the raw output of a previous block (LeakyReLU(conv) of random data, with ITS folded
    # InstanceNorm), which the loader normalises -- white noise in its place toggles more bits and costs ~15 % of clock
    seed = torch.randn(n, c, d, h, w, device=device)
    x = torch.empty_like(seed)
    x_scale = torch.empty(n * c * d, device=device)
    x_shift = torch.empty(n * c * d, device=device)
    raw = torch.empty_like(x)
    scale = torch.empty(n * c * d, device=device)
    shift = torch.empty(n * c * d, device=device)
    nbytes = lib.pds_conv_block_workspace_bytes(n, c, c, d, h, w, 1, 1, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    stream = _lib.stream_handle(device)

    _lib.check(lib.pds_conv_block_fwd(ctypes.byref(params), _lib.ptr(seed), _lib.ptr(x), _lib.ptr(x_scale), _lib.ptr(x_shift),
                                      n, c, c, d, h, w, 1, 1, 1, _lib.ptr(ws), ws.numel(), stream), 'pds_conv_block_fwd')
    del seed

    # the range certificate the fp16-split form scales its operands by (ABI v5; inside the modules in_finalize writes
    # |gamma| sqrt(plane size) + |beta|, which is what this is)
    x_bound = (block.norm.weight.detach().abs() * (h * w) ** 0.5 + block.norm.bias.detach().abs()).max().reshape(1).contiguous()

    def launch():
        _lib.check(lib.pds_conv_block_chained_fwd(ctypes.byref(params), _lib.ptr(x), _lib.ptr(x_scale), _lib.ptr(x_shift),
                                                  1, _lib.ptr(x_bound), _lib.ptr(raw), _lib.ptr(scale), _lib.ptr(shift),
                                                  n, c, c, d, h, w, 1, 1, 1, _lib.ptr(ws), ws.numel(), stream),
                   'pds_conv_block_chained_fwd')
    for _ in range(2):
        launch()
    torch.cuda.synchronize(device)
    # Every launch is bracketed by its own pair of events on the launch stream (no host synchronisation in between); a
    # memory-bound pass over the output sits between two launches, outside the brackets, as materialize_l0 / conv2d_t8 do
    # in the hot path.  Timed back to back with nothing in between, consecutive launches of this power-limited kernel
    # run at lower clocks than inside the path (0.47-0.51 ms against the 0.36-0.38 ms rocprofv3 reports there).
    spacer = torch.empty_like(raw)

    def timed(spaced):
        events = []
        for _ in range(reps):
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            launch()
            stop.record()
            if spaced:
                torch.add(raw, 1.0, out=spacer)
            events.append((start, stop))
        torch.cuda.synchronize(device)
        return sum(a.elapsed_time(b) for a, b in events) / reps
    return {'spaced_ms': timed(True), 'back_to_back_ms': timed(False)}


def time_dominant_kernel_in_situ(run_pair, device, pairs):
    """Durations (ms) of the 48-plane conv2d_x3 launches INSIDE sequential pairs of the hot path: the library's launch
    probe (ABI v5, pds_probe_begin / pds_probe_end) puts a HIP event pair on the launch stream around every launch of
    the kernel while `run_pair(i)` enqueues whole pairs -- the kernel the path runs, with the cache state and clocks it
    finds there, not a micro-benchmark.  Returns None when no such launch was seen (PDS_X3=0)."""
    lib = _lib.load()
    full = 48 * 9 * 8   # planes x 16 x 32-pixel tiles of a 144 x 240 plane
    for i in range(2):
        run_pair(i)
    torch.cuda.synchronize(device)
    capacity = min(256, 4 * pairs)
    _lib.check(lib.pds_probe_begin(b'conv2d_x3<fp16>' if X3_PRODUCTS == 3.0 else b'conv2d_x3<bf16>', capacity), 'pds_probe_begin')
    for i in range(pairs):
        run_pair(i)
    ms = (ctypes.c_float * capacity)()
    wgs = (ctypes.c_int * capacity)()
    n = lib.pds_probe_end(ms, wgs, capacity)
    torch.cuda.synchronize(device)
    if n < 0:
        _lib.check(n, 'pds_probe_end')
    times = [ms[i] for i in range(n) if wgs[i] == full]
    if times:   # a host stall between the start event and the launch shows up as GPU idle time inside the bracket: drop it
        median = sorted(times)[len(times) // 2]
        times = [t for t in times if t <= 3.0 * median]
    if not times:
        return None
    times.sort()
    return {'launch_ms': sum(times) / len(times), 'min_ms': times[0], 'median_ms': times[len(times) // 2],
            'max_ms': times[-1], 'launches': len(times), 'pairs': pairs}


def cpu_baseline(net, ld, rd, shortcut, gpu_disparity, gpu_signatures=None, gpu_cost=None):
    from oracle import pds_oracle as oracle
    # Threads: the cores this process may actually run on, capped at 32 -- oneDNN's small 3-D
    # convolutions thrash with hundreds of threads (256 threads measured 282 s per pair on the
    # 2 x 64-core host of the GPU box versus a few seconds with 32).
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(usable, 32)))
    params = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    best = float('inf')
    disparity = None
    passes = 0
    budget_start = time.perf_counter()
    with torch.no_grad():
        for i in range(8):  # 1 warm-up + best of the rest: ~10-30 s of host time (at least 3 passes, stop after 12 s)
            t0 = time.perf_counter()
            signatures, cost, disparity = oracle.hot_path(params, ld, rd, shortcut, MAX_DISPARITY, return_stages=True)
            dt = time.perf_counter() - t0
            passes += 1
            if i > 0 or dt > 15.0:
                best = min(best, dt)
            spent = time.perf_counter() - budget_start
            if spent > 30.0 or (passes >= 3 and spent > 12.0):
                break
    delta = (gpu_disparity.double().cpu() - disparity.double()).abs()
    parity = {'disparity_mae': float(delta.mean()), 'disparity_max': float(delta.max()),
              'flip_fraction': float((delta > 0.5).double().mean()), 'tolerance_mae': 1e-3,
              'disparity_mae_without_flips': float(delta[delta <= 0.5].mean())}
    # stage-wise: an arg-max flip moves one pixel by tens of px, so the disparity MAE alone can hide or
    # exaggerate a regression; the signatures and the cost volume cannot
    if gpu_signatures is not None:
        d = (gpu_signatures.cpu() - signatures).abs()
        parity['signatures_max'] = float(d.max())
        parity['signatures_mean'] = float(d.mean())
        parity['signatures_tolerance_max'] = 2e-5
    if gpu_cost is not None:
        d = (gpu_cost.cpu() - cost).abs()
        parity['cost_max'] = float(d.max())
        parity['cost_mean'] = float(d.mean())
        parity['cost_tolerance_max'] = 1e-4
        parity['cost_tolerance_mean'] = 1e-5
    # SURVEY.md 8c, last row: "the GPU must not be further from the truth than the reference is".  The truth is the
    # oracle in fp64 on the same inputs; both distances ride on the line (the gates are in
    # tests/test_gpu_parity.py: test_config2_fp64_arbiter)
    try:
        parity['fp64_arbiter'] = fp64_arbiter(oracle, params, ld, rd, shortcut, (signatures, cost, disparity),
                                              (gpu_signatures, gpu_cost, gpu_disparity))
    except Exception as e:   # a diagnostics leg must never take the line down
        parity['fp64_arbiter'] = {'error': '%s: %s' % (type(e).__name__, e)}
    cpu_name = ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu_name = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')][0]
    except Exception:
        pass
    base = {'value': 1.0 / best, 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'best of %d full passes (first one is warm-up, ~12 s of host time) of the same 960x540 D=192 pair, '
                      'PyTorch-CPU oracle, %s, %d usable cores' % (passes, cpu_name, usable),
            'ms_per_pair': best * 1e3}
    # SURVEY.md 8d also asks for the single-thread figure: ONE pass (about 10-15 s), no warm-up
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    with torch.no_grad():
        t0 = time.perf_counter()
        oracle.hot_path(params, ld, rd, shortcut, MAX_DISPARITY)
        single = time.perf_counter() - t0
    torch.set_num_threads(threads)
    base['single_thread'] = {'value': 1.0 / single, 'unit': 'pairs/s', 'cores': 1, 'ms_per_pair': single * 1e3,
                             'sample': 'one full pass of the same pair, torch.set_num_threads(1)'}
    # thread sweep: the capped figure above is the 32-thread point; 64 and 128 threads in child
    # processes with a time limit (oneDNN thrashes on this host with hundreds of threads: 282 s per pair at 256).
    # "value" is the BEST point of the sweep, so the baseline is not handicapped by the cap.
    sweep = [{'cores': base['cores'], 'value': base['value'], 'ms_per_pair': base['ms_per_pair']}]
    for threads in (64, 128):
        if threads <= usable and threads > base['cores']:
            sweep.append(cpu_threads_leg(params, ld, rd, shortcut, threads))
    base['thread_sweep'] = sweep
    best_point = max((p for p in sweep if p.get('value')), key=lambda p: p['value'])
    if best_point['cores'] != base['cores']:
        base['capped_32_threads'] = {'value': base['value'], 'ms_per_pair': base['ms_per_pair'], 'cores': base['cores']}
        base['value'], base['ms_per_pair'], base['cores'] = best_point['value'], best_point['ms_per_pair'], best_point['cores']
        base['sample'] += '; best point of the thread sweep (%d threads, child process)' % best_point['cores']
    else:
        base['sample'] += '; best point of the thread sweep %s' % [p['cores'] for p in sweep]
    return base, parity


def fp64_arbiter(oracle, params, ld, rd, shortcut, cpu32, gpu):
    """Stage-wise distances of the CPU fp32 oracle and of the GPU result to the fp64 oracle on identical inputs."""
    with torch.no_grad():
        s64, c64, d64 = oracle.hot_path(oracle.cast_params(params, torch.float64), ld.double(), rd.double(),
                                        shortcut.double(), MAX_DISPARITY, return_stages=True)

    def stage(x, truth):
        if x is None:
            return None
        d = (x.detach().double().cpu() - truth).abs()
        return {'max': float(d.max()), 'mean': float(d.mean())}

    def disparity(x):
        d = (x.detach().double().cpu() - d64).abs()
        smooth = d[d <= 0.5]
        return {'mae': float(d.mean()), 'flips': int((d > 0.5).sum()), 'pixels': int(d.numel()),
                'smooth_mae': float(smooth.mean()) if smooth.numel() else 0.0}
    record = {'truth': 'oracle/pds_oracle.py in fp64 on the same descriptors and weights',
              'gpu_vs_fp64': {'signatures': stage(gpu[0], s64), 'cost': stage(gpu[1], c64), 'disparity': disparity(gpu[2])},
              'cpu_fp32_vs_fp64': {'signatures': stage(cpu32[0], s64), 'cost': stage(cpu32[1], c64),
                                   'disparity': disparity(cpu32[2])}}
    g, c = record['gpu_vs_fp64']['disparity'], record['cpu_fp32_vs_fp64']['disparity']
    record['gpu_no_further_from_fp64_than_reference'] = bool(g['flips'] <= c['flips'] + 2 and
                                                             g['smooth_mae'] <= 2.0 * c['smooth_mae'] + 1e-5)
    return record


ALL_CORES_TIMEOUT_S = 40.0


def cpu_threads_leg(params, ld, rd, shortcut, usable):
    """One point of the CPU baseline's thread sweep: the oracle on `usable` threads in a child process with a time
    limit (oneDNN thrashes on this host with hundreds of threads -- 282 s per pair was measured at 256 -- so a pass
    that does not finish within the limit is reported as slower than 1 / limit, not waited for)."""
    import subprocess
    import tempfile
    code = ("import sys, time, torch\n"
            "sys.path.insert(0, %r)\n"
            "from oracle import pds_oracle as oracle\n"
            "blob = torch.load(sys.argv[1])\n"
            "torch.set_num_threads(int(sys.argv[2]))\n"
            "best = 1e30\n"
            "with torch.no_grad():\n"
            "    for i in range(2):\n"
            "        t0 = time.perf_counter()\n"
            "        oracle.hot_path(blob['params'], blob['ld'], blob['rd'], blob['shortcut'], %d)\n"
            "        best = min(best, time.perf_counter() - t0)\n"
            "        print('PASS', best, flush=True)\n" % (os.path.dirname(os.path.abspath(__file__)), MAX_DISPARITY))
    with tempfile.TemporaryDirectory() as folder:
        blob = os.path.join(folder, 'inputs.pt')
        torch.save({'params': params, 'ld': ld, 'rd': rd, 'shortcut': shortcut}, blob)
        t0 = time.perf_counter()
        try:
            out = subprocess.run([sys.executable, '-c', code, blob, str(usable)], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, timeout=ALL_CORES_TIMEOUT_S)
            text = out.stdout.decode(errors='replace')
        except subprocess.TimeoutExpired as e:
            text = (e.stdout or b'').decode(errors='replace')
        spent = time.perf_counter() - t0
    passes = [float(t.split()[1]) for t in text.splitlines() if t.startswith('PASS')]
    sample = ('the same pair, torch.set_num_threads(%d), child process limited to %d s (PyTorch-CPU oracle)'
              % (usable, int(ALL_CORES_TIMEOUT_S)))
    if not passes:
        return {'value': None, 'unit': 'pairs/s', 'cores': usable, 'slower_than_pairs_per_s': 1.0 / spent,
                'sample': sample + ': no pass finished within the limit'}
    return {'value': 1.0 / passes[-1], 'unit': 'pairs/s', 'cores': usable, 'ms_per_pair': passes[-1] * 1e3,
            'sample': sample + ': best of %d passes' % len(passes)}


def gpu_baseline(net, ld, rd, shortcut, device, gpu_disparity):
    """The same restatement (oracle.hot_path: plain PyTorch ops) on the MI355X through PyTorch-ROCm / MIOpen: what a
    user gets on this GPU without the hand-written kernels.  Same inputs, after the
    timed region; best of 3 after 2 warm-up passes (MIOpen picks its kernels in the first)."""
    from oracle import pds_oracle as oracle
    params = {k: v.detach() for k, v in net.state_dict().items()}
    best = float('inf')
    disparity = None
    with torch.no_grad():
        for i in range(5):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            disparity = oracle.hot_path(params, ld, rd, shortcut, MAX_DISPARITY)
            torch.cuda.synchronize(device)
            if i >= 2:
                best = min(best, time.perf_counter() - t0)
    delta = (gpu_disparity.double() - disparity.double()).abs()
    return {'value': 1.0 / best, 'unit': 'pairs/s', 'ms_per_pair': best * 1e3, 'kind': 'port',
            'sample': 'best of 3 full passes after 2 warm-up passes of the same 960x540 D=192 pair: oracle/pds_oracle.py '
                      '(plain torch.nn.functional ops) on cuda:0 = PyTorch-ROCm %s + MIOpen' % torch.__version__,
            'disparity_mae_vs_hip_path': float(delta.mean())}


def train_record(device, steps=3, cpu=True):
    """Config 5 on the default line: a short driver-visible record of the full-size training step --
    train-mode PdsNetwork forward, SubpixelCrossEntropy, backward through the HIP modules, RMSprop -- with the same
    step of the CPU oracle (ONE step: forward + loss + backward, no optimizer) timed on the host beside it."""
    from practicaldeepstereo_nips2018_amd.training import DataParallelTrainer, synthetic_example
    torch.cuda.synchronize(device)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(device)
    resident_before = torch.cuda.memory_allocated(device)   # the inference part's tensors and workspaces stay allocated
    trainer = DataParallelTrainer(MAX_DISPARITY, device)
    left, right, truth = synthetic_example(HEIGHT, WIDTH, MAX_DISPARITY, 1, device)
    losses = [trainer.step(left, right, truth)]                # warm-up: workspaces, weight re-layout
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(trainer.step(left, right, truth))
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    trainer.optimizer.zero_grad(set_to_none=True)              # forward / backward split of one more (untimed) step
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    cost = trainer.network(left, right)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    trainer.criterion(cost, truth).backward()
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    values = [float(v) for v in losses]
    record = {'workload': 'configs[4] on one GPU: 960x540 pair, D=192, batch 1, train-mode PdsNetwork forward + '
                          'SubpixelCrossEntropy + backward through the HIP modules + RMSprop (lr 1e-2), random-init '
                          'weights seed 0, ground truth with an unknown band',
              'steps': steps, 'steps_per_s': steps / elapsed, 'ms_per_step': elapsed / steps * 1e3,
              'forward_ms': (t1 - t0) * 1e3, 'loss_backward_ms': (t2 - t1) * 1e3,
              'first_loss': values[0], 'last_loss': values[-1],
              'peak_memory_gb': (torch.cuda.max_memory_allocated(device) - resident_before) / 2 ** 30,
              'peak_memory_note': 'PyTorch-allocated peak of the training step above what the inference part of this run '
                                  'had left resident (%.1f GB)' % (resident_before / 2 ** 30)}
    if cpu:
        try:
            record['cpu_baseline'] = train_cpu_baseline(trainer.network, left, right, truth, values[0])
        except Exception as e:   # a baseline leg must never take the line down
            record['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
    del trainer, cost
    torch.cuda.empty_cache()
    return record


def train_cpu_baseline(network, left, right, truth, gpu_first_loss):
    """ONE full-size training step (forward + loss + backward) of the CPU oracle on min(32, usable) host threads."""
    from oracle import pds_oracle as oracle
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(usable, 32)))
    # the parameters as they were at the first step (seed 0): the comparison of the loss value below needs them
    torch.manual_seed(0)
    from practicaldeepstereo_nips2018_amd.network import PdsNetwork
    fresh = PdsNetwork.default(MAX_DISPARITY)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in fresh.state_dict().items()}
    t0 = time.perf_counter()
    cost = oracle.network_training_output(params, left.cpu(), right.cpu(), MAX_DISPARITY)
    loss = oracle.subpixel_cross_entropy(cost, truth.cpu())
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    return {'ms_per_step': (t2 - t0) * 1e3, 'forward_ms': (t1 - t0) * 1e3, 'loss_backward_ms': (t2 - t1) * 1e3,
            'cores': torch.get_num_threads(), 'kind': 'port', 'loss': float(loss),
            'loss_difference_to_gpu_first_step': abs(float(loss) - gpu_first_loss),
            'sample': 'one step (no warm-up, no optimizer update) of the PyTorch-CPU oracle with autograd on the same pair'}


def train_main(args, world, rank, device, collectives=None):
    """--train: the config-5 line (same timing contract: W warm-up steps, then exactly K steps between barriers and
    synchronisations, MAX over ranks; rank 0 prints one JSON line)."""
    from practicaldeepstereo_nips2018_amd.training import DataParallelTrainer, synthetic_example
    trainer = DataParallelTrainer(MAX_DISPARITY, device, share_device=args.share_device)
    left, right, truth = synthetic_example(HEIGHT, WIDTH, MAX_DISPARITY, 1 + rank, device)   # another pair per rank
    losses = [trainer.step(left, right, truth) for _ in range(args.warmup)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(trainer.step(left, right, truth))
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    in_sync = True
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)   # one timed region: MAX over ranks
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        in_sync = trainer.replicas_in_sync()
    # forward / backward split of one more (untimed) step on rank 0's pair
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    trainer.optimizer.zero_grad(set_to_none=True)
    cost = trainer.network(left, right)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    trainer.criterion(cost, truth).backward()
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    if rank == 0:
        values = [float(v) for v in losses]
        print(json.dumps({
            'metric': 'training steps (stereo pairs)/sec, 960x540 D=192: PdsNetwork train forward + SubpixelCrossEntropy '
                      '+ backward through the HIP modules + RMSprop',
            'value': world * args.steps / elapsed, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[4]: one 960x540 pair per GPU, D=192, batch 1 per rank, train mode, '
                                   'random-init weights seed 0, ground truth with an unknown band',
                       'parallelism': ('DistributedDataParallel x%d, gradient all-reduce over %s' %
                                       (world, 'RCCL' if args.backend == 'nccl' else args.backend))
                       if world > 1 else 'single GPU',
                       'collectives': collectives},
            'forward_ms': (t1 - t0) * 1e3, 'loss_backward_ms': (t2 - t1) * 1e3,
            'first_loss': values[0], 'last_loss': values[-1], 'replicas_in_sync': in_sync,
            'peak_memory_gb': torch.cuda.max_memory_allocated(device) / 2 ** 30}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class ClockSampler:
    """Shader-clock samples of the GPU while a timed window runs (the dominant kernel is
    power-limited, so a throughput figure needs the clocks it was measured at).  Reads the driver's sysfs table
    (pp_dpm_sclk: the line marked '*' is the current level) from a thread; falls back to `rocm-smi --showclocks --json`."""

    def __init__(self, period=0.05):
        import glob
        import threading
        self.period = period
        self.samples = []
        self.source = None
        self._files = sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read_sysfs(self):
        best = None
        for path in self._files:
            try:
                with open(path) as f:
                    for line in f:
                        if line.rstrip().endswith('*'):
                            mhz = float(''.join(ch for ch in line.split(':')[1] if ch.isdigit() or ch == '.'))
                            best = mhz if best is None else max(best, mhz)   # (the busy device of a multi-GPU node)
            except (OSError, ValueError, IndexError):
                continue
        return best

    def _read_smi(self):
        import subprocess
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=5).stdout.decode(errors='replace')
            best = None
            for card in json.loads(out).values():
                for key, value in card.items():
                    if 'sclk' in key.lower() and 'mhz' in str(value).lower():
                        mhz = float(''.join(ch for ch in str(value).split('(')[-1] if ch.isdigit() or ch == '.'))
                        best = mhz if best is None else max(best, mhz)
            return best
        except Exception:   # noqa: BLE001
            return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read_sysfs() if self._files else None
            if v is not None:
                self.source = 'sysfs pp_dpm_sclk'
            else:
                v = self._read_smi()
                if v is not None:
                    self.source = 'rocm-smi --showclocks'
            if v is not None:
                self.samples.append(v)
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(10)

    def summary(self):
        if not self.samples:
            return {'samples': 0, 'note': 'no clock source readable on this host (sysfs pp_dpm_sclk, rocm-smi)'}
        ordered = sorted(self.samples)
        return {'samples': len(ordered), 'sclk_mhz_min': ordered[0], 'sclk_mhz_median': ordered[len(ordered) // 2],
                'sclk_mhz_max': ordered[-1], 'source': self.source}


def sustained_record(step, finish, device, seconds, rate_guess):
    """ONE contiguous timed window of at least `seconds` (>= 800 steps at the rates of this path): the 20-step regions
    of the headline are 50 ms each, and the dominant kernel is power-limited."""
    steps = max(800, int(seconds * rate_guess))
    torch.cuda.synchronize(device)
    with ClockSampler() as clocks:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        finish()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
    return {'value': steps / elapsed, 'unit': 'pairs/s', 'steps': steps, 'seconds': elapsed,
            'ms_per_step': elapsed / steps * 1e3, 'clocks': clocks.summary(),
            'note': 'one contiguous window, same schedule as "value" (pairs over the HIP streams), synchronise - time - synchronise'}


def exact_fp32_record(args):
    """The same bench in a child process with PDS_DEBUG_SWITCHES=1 PDS_X3=0: the 64-channel layers on the exact-fp32
    kernels of round 2 (IEEE fp32 multiplies, Winograd F(2,3) / direct on v_mfma_f32_*_f32) -- the driver-timed
    exact-fp32 figure beside the fp16-split headline."""
    import subprocess
    env = dict(os.environ, PDS_DEBUG_SWITCHES='1', PDS_X3='0')
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(min(args.steps, 200)), '--warmup', str(min(args.warmup, 10)),
           '--windows', '3', '--kernel-reps', str(min(args.kernel_reps, 6)), '--streams', str(args.streams),
           '--no-cpu-baseline', '--no-train-record', '--no-sub-records']
    t0 = time.perf_counter()
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    lines = [l for l in out.stdout.decode(errors='replace').splitlines() if l.startswith('{')]
    if out.returncode != 0 or not lines:
        return {'error': 'child exited %d: %s' % (out.returncode, out.stderr.decode(errors='replace')[-300:])}
    child = json.loads(lines[-1])
    roof = child.get('roofline', {})
    return {'value': child['value'], 'unit': child['unit'], 'ms_per_step': child['ms_per_step'],
            'ms_per_frame': child.get('ms_per_frame'), 'arithmetic': child.get('arithmetic'),
            'switches': 'PDS_DEBUG_SWITCHES=1 PDS_X3=0 (child process, %.0f s)' % (time.perf_counter() - t0),
            'pipelined_equals_sequential': child.get('pipelined_equals_sequential'),
            'roofline': {k: roof.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'launch_ms',
                                                   'executed_gflop_per_launch', 'algorithmic_gflop_per_launch')}}


def config4_record(device, streams):
    """BASELINE configs[3]: KITTI shape 1242x375 (padded 384x1280 by size_adapter.py:29-43), D=256, batch 4, inference:
    the hot path on descriptors of that shape (random, seeded), sequential and over the HIP streams."""
    batch, h, w, maxd = 4, 96, 320, 255
    torch.manual_seed(0)
    net = pds.PdsNetwork.default(maxd).eval().to(device).freeze_weights()
    g = torch.Generator().manual_seed(1)
    ld = torch.randn(batch, 64, h, w, generator=g).to(device)
    rd = torch.randn(batch, 64, h, w, generator=g).to(device)
    sc = torch.randn(batch, 8, h, w, generator=g).to(device)

    def whole(a, b, c):
        return net._regularization.forward_with_estimator(net._matching(a, b), c, net._estimator)
    with torch.no_grad():
        for _ in range(2):
            expected = whole(ld, rd, sc)
        torch.cuda.synchronize(device)
        n = 8
        t0 = time.perf_counter()
        for _ in range(n):
            whole(ld, rd, sc)
        torch.cuda.synchronize(device)
        sequential = (time.perf_counter() - t0) / n
        lanes = PairStreams(whole, streams=streams)
        for _ in range(streams):
            lanes.submit(ld, rd, sc)
        lanes.drain()
        torch.cuda.synchronize(device)
        n = 12
        t0 = time.perf_counter()
        outs = [lanes.submit(ld, rd, sc) for _ in range(n)]
        lanes.drain()
        torch.cuda.synchronize(device)
        pipelined = (time.perf_counter() - t0) / n
        same = all(torch.equal(o, expected) for o in outs)
    record = {'workload': 'configs[3]: 1242x375 padded to 384x1280, D=256 (64 matching planes, 128 cost planes), batch 4, '
                          'eval mode, random-init weights seed 0, seeded random descriptors',
              'value': batch / pipelined, 'unit': 'pairs/s', 'ms_per_batch': pipelined * 1e3,
              'sequential': {'value': batch / sequential, 'unit': 'pairs/s', 'ms_per_batch': sequential * 1e3},
              'pipelined_equals_sequential': bool(same),
              'peak_memory_gb': torch.cuda.max_memory_allocated(device) / 2 ** 30}
    del net, lanes, outs, expected
    torch.cuda.empty_cache()
    return record


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def relaunch_under_torchrun(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script ourselves (one per GPU, RCCL),
    exactly as the documented ``python -m torch.distributed.run`` command line would."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(args)   # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    _lib.load()  # fail loudly when the HIP extension is missing
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    if world > 1 and not args.share_device and torch.cuda.device_count() < world:
        raise SystemExit('bench.py: %d ranks need %d GPUs, this node shows %d (functional check on one GPU: '
                         '--share-device --backend gloo)' % (world, world, torch.cuda.device_count()))
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)

    collectives = None
    if world > 1:
        # the first multi-GPU run is also the first RCCL run of this code: check the collectives on tiny tensors against
        # locally computed expectations BEFORE the timed region and say which form of the gather is in use
        from practicaldeepstereo_nips2018_amd.distributed import preflight_collectives
        try:
            collectives = preflight_collectives(device=device)
        except Exception as e:   # fail loudly, with a line the driver can parse, instead of hanging in the timed region
            if rank == 0:
                print(json.dumps({'metric': 'stereo pairs/sec, 960x540 D=192, Matching+Regularization+SubpixelMap hot path',
                                  'value': None, 'n_gpus': world, 'error': 'collective pre-flight failed: %s: %s'
                                                                           % (type(e).__name__, e)}))
            raise SystemExit(3)
    if args.train:
        return train_main(args, world, rank, device, collectives)
    net, descriptors, images = make_inputs(device)
    ld_g, rd_g, sc_g = descriptors[0]
    regularization, estimator = net._regularization, net._estimator

    def tail(signatures, shortcut):
        return regularization.forward_with_estimator(signatures, shortcut, estimator)

    # N > 1: Matching sharded along the disparity axis + one all-gather per pair on every rank; the tail of
    # pair i (Regularization + estimator, not shardable) runs on rank i % N on a side stream instead of being
    # replicated N times (distributed.ShardedHotPath).
    # N = 1: whole pairs are dealt round-robin to a few HIP streams (distributed.PairStreams): the HBM- and latency-bound
    # phases of one pair run beside the MFMA-bound kernels of another.  --no-pipeline (and --graph) time strictly
    # sequential pairs.
    if world > 1:
        pipeline = ShardedHotPath(net._matching, tail, streams=args.sharded_streams)
    elif args.graph or args.no_pipeline:
        pipeline = None
    else:
        pipeline = PairStreams(lambda left, right, shortcut: tail(net._matching(left, right), shortcut),
                               streams=args.streams)

    counter = [0]

    def step():
        # the timed steps rotate through PAIRS distinct stereo pairs
        ld, rd, sc = descriptors[counter[0] % PAIRS]
        counter[0] += 1
        if pipeline is not None:
            return pipeline.submit(ld, rd, sc)
        return tail(net._matching(ld, rd), sc)

    def finish():
        if pipeline is not None:
            pipeline.drain()

    def barrier():
        if world > 1:
            dist.barrier()

    # --graph: the ~70 launches of a step are captured once into a HIP graph and replayed (the C ABI never
    # allocates or synchronises, so the whole hot path is capturable).
    use_graph = world == 1 and args.graph
    if use_graph:
        with torch.no_grad():
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                step()
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    graph_out = step()
            torch.cuda.current_stream(device).wait_stream(side)
        def step():  # noqa: F811   (the captured step replays pair 0)
            counter[0] += 1
            graph.replay()
            return graph_out

    with torch.no_grad():
        if pipeline is not None:
            # set-up, not a warm-up step: every stream of the schedule creates its module workspaces once
            for _ in range(max(args.streams, args.sharded_streams)):
                step()
            finish()
        for _ in range(args.warmup):
            disparity = step()
        finish()
        counter[0] = 0

        def timed_window():
            barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            results = []
            for _ in range(args.steps):
                first = counter[0] % PAIRS
                out = step()
                if out is not None:
                    results.append((first if not use_graph else 0, out))
            finish()
            torch.cuda.synchronize(device)
            barrier()
            return time.perf_counter() - t0, results
        first_elapsed, mine = timed_window()     # a timed region: exactly --steps steps between barriers + synchronisations
        disparity = dict(mine).get(0)
        # --windows regions of exactly --steps steps each; "value" is the MEDIAN region (one 20-step region is 50 ms: a
        # single host stall moves it by percent), the first and the spread ride along in "windows"
        windows = [first_elapsed] + [timed_window()[0] for _ in range(args.windows - 1)]
        elapsed = sorted(windows)[len(windows) // 2]

    def reference_results():
        """unsharded, sequential hot path of every pair (the bit-exactness reference of the schedules)"""
        with torch.no_grad():
            return [tail(net._matching(ld, rd), sc) for ld, rd, sc in descriptors]
    replica_elapsed = latency_elapsed = sharded_ok = None
    if world == 1 and pipeline is not None:
        # latency of one pair (no overlap across pairs), and a bit-exactness check of the pipelined results
        with torch.no_grad():
            for i in range(2):
                tail(net._matching(*descriptors[i % PAIRS][:2]), descriptors[i % PAIRS][2])
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(args.steps):
                ld, rd, sc = descriptors[i % PAIRS]
                tail(net._matching(ld, rd), sc)
            torch.cuda.synchronize(device)
            latency_elapsed = time.perf_counter() - t0   # (exactly --steps sequential pairs)
        expected = reference_results()
        torch.cuda.synchronize(device)
        sharded_ok = all(torch.equal(m, expected[i]) for i, m in mine) and len(mine) == args.steps
    if world > 1:
        t = torch.tensor(windows, device=device, dtype=torch.float64)   # MAX over ranks of every region, then the median
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        windows = [float(v) for v in t.tolist()]
        elapsed = sorted(windows)[len(windows) // 2]
        # untimed check: every disparity map this rank produced in the timed region equals the unsharded hot path
        # on the same inputs bit for bit (same kernels, same planes)
        expected = reference_results()
        torch.cuda.synchronize(device)
        good = all(torch.equal(m, expected[i]) for i, m in mine)
        flag = torch.tensor([1.0 if good else 0.0, float(len(mine))], device=device, dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        sharded_ok = bool(flag[0].item() == world) and int(flag[1].item()) == args.steps
        # latency mode, informational: one pair at a time, tail replicated on every rank (no overlap across pairs)
        latency_matching = ShardedMatching(net._matching)
        with torch.no_grad():
            for _ in range(max(1, args.warmup // 2)):
                tail(latency_matching(ld_g, rd_g), sc_g)
            barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tail(latency_matching(ld_g, rd_g), sc_g)
            torch.cuda.synchronize(device)
            barrier()
            t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        latency_elapsed = float(t.item())
        # Second, informational mode: every rank runs the whole (unsharded) hot path on its own pair --
        # N independent replicas, no collective; aggregate throughput = N * steps / max time.
        unsharded = net._matching

        replicas = PairStreams(lambda a, b, c: regularization.forward_with_estimator(unsharded(a, b), c, estimator),
                               streams=args.streams)

        def replica_step():
            return replicas.submit(ld_g, rd_g, sc_g)
        with torch.no_grad():
            for _ in range(max(3, args.warmup // 2)):
                replica_step()
            replicas.drain()
            barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                replica_step()
            replicas.drain()
            torch.cuda.synchronize(device)
            barrier()
            t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        replica_elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            'metric': 'stereo pairs/sec, 960x540 D=192, Matching+Regularization+SubpixelMap hot path',
            'value': args.steps / elapsed,
            'unit': 'pairs/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'ms_per_frame': latency_elapsed / args.steps * 1e3 if latency_elapsed is not None else ms_per_step,
            'higher_is_better': True,
            'scaling': 'strong' if world > 1 else 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            # dtype names the type of every tensor and accumulator; the MULTIPLIES of the 64-channel layers are not IEEE
            # fp32 multiplies:
            'arithmetic': ARITHMETIC,
            'data': 'synthetic',
            'config': {'workload': 'configs[1]: 960x540 pair padded to 576x960, D=192 (48 matching planes, 96 cost '
                                   'planes), batch 1, eval mode, random-init weights seed 0',
                       'parallelism': ('disparity-axis shard x%d + one all-gather (%s) per pair; Regularization + '
                                       'estimator of pair i on rank i %% %d; pairs dealt to %d streams per rank' %
                                       (world, gather_description(), world, args.sharded_streams))
                       if world > 1 else 'single GPU',
                       'collectives': collectives,
                       'launch': 'hip graph replay' if use_graph else
                                 ('eager, whole pairs round-robin over %d HIP streams (ms_per_frame is the un-overlapped '
                                  'latency of one pair)' % args.streams
                                  if pipeline is not None and world == 1 else 'eager')},
        }
        if world == 1 and latency_elapsed is not None:
            line['pipelined_equals_sequential'] = sharded_ok
            line['sequential'] = {'value': args.steps / latency_elapsed, 'unit': 'pairs/s',
                                  'ms_per_frame': latency_elapsed / args.steps * 1e3,
                                  'note': 'one pair strictly after the other on one stream (--no-pipeline times this '
                                          'mode as value)'}
        if world > 1 and sharded_ok is not None:
            line['sharded_equals_unsharded'] = sharded_ok
            line['latency_mode'] = {'ms_per_frame': latency_elapsed / args.steps * 1e3,
                                    'note': 'one pair at a time: sharded Matching + all-gather + tail replicated on '
                                            'every rank, no overlap across pairs'}
        if replica_elapsed is not None:
            line['replica_mode'] = {'value': world * args.steps / replica_elapsed, 'unit': 'pairs/s',
                                    'ms_per_step': replica_elapsed / args.steps * 1e3, 'scaling': 'weak',
                                    'note': 'independent pairs on every rank (dealt to --streams HIP streams), no collective '
                                            '(throughput mode); '
                                            '"value" above is the disparity-sharded mode of north_star'}
        if world == 1:
            # informational: the whole PdsNetwork.forward (network.py:45-52: pad, descriptor network on both images,
            # hot path, crop), everything on the library; the headline value stays the hot path of the metric
            with torch.no_grad():
                for i in range(3):
                    net(*images[i % PAIRS])
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                full_steps = min(args.steps, 100)
                for i in range(full_steps):
                    net(*images[i % PAIRS])
                torch.cuda.synchronize(device)
            line['full_forward_ms'] = (time.perf_counter() - t0) / full_steps * 1e3
            # the reference's own time-per-image protocol (trainer.py:141-148; README: 0.62 s per image on the
            # authors' GPU): host images in, one example at a time, synchronize - time - synchronize around the network
            from practicaldeepstereo_nips2018_amd.timing import time_per_image
            host_examples = [{'left': images[i % PAIRS][0].cpu(), 'right': images[i % PAIRS][1].cpu(),
                              'disparity': (torch.rand(1, HEIGHT, WIDTH) * 190.0)} for i in range(2 + 8)]
            line['time_per_image'] = time_per_image(net, host_examples, device, warmup=2)
            line['time_per_image'].pop('mean_absolute_error', None)   # random ground truth: only the protocol counts
            line['time_per_image'].pop('three_pixels_error', None)
        with torch.no_grad():
            isolated = time_dominant_kernel(net, device, args.kernel_reps)
            in_situ = None
            if world == 1 and X3:
                def run_pair(i):
                    ld, rd, sc = descriptors[i % PAIRS]
                    return tail(net._matching(ld, rd), sc)
                in_situ = time_dominant_kernel_in_situ(run_pair, device, max(4, args.kernel_reps))
        # launch_ms: the kernel INSIDE the path (launch probe); the micro-benchmark figures ride along
        kernel_ms = in_situ['launch_ms'] if in_situ else isolated['spaced_ms']
        executed = CONV64_EXECUTED_GFLOP / kernel_ms  # GFLOP / ms == TFLOP/s
        traffic, traffic_source, traffic_stale = conv64_hbm_traffic()
        # "achieved" / "frac" are the EXECUTED matrix flops against the peak of the pipe that executes them (SURVEY.md 8d:
        # F_executed / (t * peak)); the algorithmic figures of the direct fp32 convolution are carried beside them
        line['roofline'] = {
            'kernel': 'conv2d 3x3 64->64 (+bias, LeakyReLU, InstanceNorm partials) over 48 planes of 144x240, one launch '
                      '(%s)' % ('conv2d_x3_kernel' if X3 else 'conv2d_wino16_kernel'),
            'bound': 'mfma', 'achieved': executed, 'peak': CONV64_EXECUTED_PEAK, 'unit': 'TFLOP/s',
            'frac': executed / CONV64_EXECUTED_PEAK,
            'frac_of_sustained_mfma_stream': (executed / 1800.0) if X3 else None,   # bare MFMA stream at the power limit
            'traffic': traffic, 'traffic_unit': 'bytes per launch', 'traffic_source': traffic_source,
            'traffic_stale': traffic_stale,
            'launch_ms': kernel_ms,
            'launch_ms_source': ('in situ: HIP event pairs on the launch stream around the 48-plane launches of %d sequential '
                                 'pairs of the hot path (launch probe, include/pds_hip.h ABI v5)' % in_situ['pairs'])
                                if in_situ else 'isolated launches (no in-situ record: multi-GPU run or PDS_X3=0)',
            'in_situ': in_situ,
            'isolated': dict(isolated, note='pds_conv_block_chained_fwd timed alone (weight packing and in_finalize inside '
                                            'the bracket): with a memory-bound pass between launches / back to back'),
            'executed_gflop_per_launch': CONV64_EXECUTED_GFLOP,
            'algorithm': ('fp32 operands split into %s, %d partial products per multiply on v_mfma_f32_32x32x16_%s with fp32 '
                          'accumulation (mean error below the fp32 fmaf chain\'s, tools/ubench/fp16x2_probe.hip): executed '
                          'flops = %d x algorithmic, priced against the dense 16-bit MFMA peak; the bare MFMA stream '
                          'sustains 1800-1900 TFLOP/s on this chip (power-limited clock)' %
                          (('two fp16 parts', 3, 'f16', 3) if X3_PRODUCTS == 3.0 else ('three bf16 parts', 6, 'bf16', 6)))
                         if X3 else 'Winograd F(2,3) along x on the fp32 MFMA units: executed flops = 2/3 algorithmic',
            'algorithmic_gflop_per_launch': CONV64_GFLOP,
            'algorithmic_tflops': CONV64_GFLOP / kernel_ms,
            'algorithmic_frac_of_fp32_mfma_peak': CONV64_GFLOP / kernel_ms / FP32_MFMA_PEAK_TFLOPS}
        frame_ms = line['ms_per_frame']
        line['path_roofline'] = {
            'what': 'whole hot path, one pair strictly after the other (ms_per_frame)',
            'ms_per_frame': frame_ms,
            'algorithmic_gflop_reference': PATH_GFLOP_REFERENCE,
            'algorithmic_gflop_minimal': PATH_GFLOP_MINIMAL,
            'tflops_reference_count': PATH_GFLOP_REFERENCE / frame_ms,
            'tflops_minimal_count': PATH_GFLOP_MINIMAL / frame_ms,
            'frac_of_fp32_mfma_peak_minimal_count': PATH_GFLOP_MINIMAL / frame_ms / FP32_MFMA_PEAK_TFLOPS,
            'fp32_mfma_floor_ms_minimal_count': PATH_GFLOP_MINIMAL / FP32_MFMA_PEAK_TFLOPS,
            'algorithmic_mb': PATH_ALGORITHMIC_MB,
            'hbm_gbps_algorithmic': PATH_ALGORITHMIC_MB / frame_ms,
            'hbm_frac_of_8tbps': PATH_ALGORITHMIC_MB / frame_ms / 8000.0,
            'note': ('SURVEY.md 8d counts; the 64-channel layers run on the 16-bit matrix pipe (%d executed flops per '
                     'algorithmic one), so the fp32-MFMA floor is a yardstick here, not a bound' % int(X3_PRODUCTS))
                    if X3 else 'SURVEY.md 8d counts'}
        ordered = sorted(args.steps / w for w in windows)
        line['windows'] = {'count': len(windows), 'median': ordered[len(ordered) // 2], 'min': ordered[0],
                           'max': ordered[-1], 'first': args.steps / windows[0], 'unit': 'pairs/s',
                           'note': 'timed regions of exactly --steps steps each over %d distinct pairs; "value" and '
                                   '"ms_per_step" are the median region' % PAIRS}
        if world == 1 and not args.no_cpu_baseline:
            with torch.no_grad():
                signatures_g = net._matching(ld_g, rd_g)
                cost_g = regularization(signatures_g, sc_g)
                if disparity is None:
                    disparity = tail(signatures_g, sc_g)
            base, parity = cpu_baseline(net, ld_g.cpu(), rd_g.cpu(), sc_g.cpu(), disparity, signatures_g, cost_g)
            del cost_g
            line['cpu_baseline'] = base
            line['parity'] = parity
            try:
                line['gpu_baseline'] = gpu_baseline(net, ld_g, rd_g, sc_g, device, disparity)
            except Exception as e:   # a baseline leg must never take the line down
                line['gpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and not args.no_sub_records:
            # sub-records of the metric family: none of them touches "value"
            if pipeline is not None and args.sustained_seconds > 0:
                try:
                    with torch.no_grad():
                        line['sustained'] = sustained_record(step, finish, device, args.sustained_seconds, line['value'])
                except Exception as e:
                    line['sustained'] = {'error': '%s: %s' % (type(e).__name__, e)}
            if X3:
                try:
                    line['exact_fp32'] = exact_fp32_record(args)
                except Exception as e:
                    line['exact_fp32'] = {'error': '%s: %s' % (type(e).__name__, e)}
            try:
                line['config4'] = config4_record(device, args.streams)
            except Exception as e:
                line['config4'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and not args.no_train_record:
            try:
                line['train'] = train_record(device, cpu=not args.no_cpu_baseline)
            except Exception as e:
                line['train'] = {'error': '%s: %s' % (type(e).__name__, e)}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
