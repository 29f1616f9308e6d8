"""GPU (-m gpu): the data-parallel training step of BASELINE.json configs[4], as far as one GPU can
exercise it: two processes (gloo; both bound to cuda:0 -- RCCL needs one device per rank) wrap the network in
DistributedDataParallel, each trains on its own stereo pair for two RMSprop steps (reference pds_trainer.py:35-46,
train_on_flyingthings3d.py:55-68).  Afterwards both ranks must hold bit-identical parameters, and they must equal a
single-process run that averages the two pairs' gradients by hand before each optimizer step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HEIGHT, WIDTH, MAX_DISPARITY, STEPS = 64, 128, 63, 2


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    from practicaldeepstereo_nips2018_amd.training import DataParallelTrainer, synthetic_example
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
        trainer = DataParallelTrainer(MAX_DISPARITY, dev, share_device=True)
        left, right, truth = synthetic_example(HEIGHT, WIDTH, MAX_DISPARITY, 1 + rank, dev)
        losses = [float(trainer.step(left, right, truth)) for _ in range(STEPS)]
        in_sync = trainer.replicas_in_sync()
        torch.cuda.synchronize()
        results[rank] = {'in_sync': in_sync, 'losses': losses,
                         'params': [p.detach().cpu() for p in trainer.network.parameters()]}
    finally:
        dist.destroy_process_group()


def single_process_reference(dev):
    """The same two steps in one process: gradient of each pair in turn, averaged, one RMSprop step."""
    from practicaldeepstereo_nips2018_amd.training import DataParallelTrainer, synthetic_example
    trainer = DataParallelTrainer(MAX_DISPARITY, dev)
    examples = [synthetic_example(HEIGHT, WIDTH, MAX_DISPARITY, 1 + r, dev) for r in range(2)]
    params = list(trainer.network.parameters())
    for _ in range(STEPS):
        total = [torch.zeros_like(p) for p in params]
        for left, right, truth in examples:
            trainer.optimizer.zero_grad(set_to_none=True)
            trainer.criterion(trainer.network(left, right), truth).backward()
            for t, p in zip(total, params):
                t += p.grad
        for t, p in zip(total, params):
            p.grad = t / len(examples)
        trainer.optimizer.step()
    return [p.detach().cpu() for p in params]


def test_two_rank_data_parallel_training_steps(hip_library):
    assert torch.cuda.is_available()
    ctx = mp.get_context('spawn')
    results = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, results)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        if p.is_alive():   # a hung rendezvous must not outlive the test
            p.kill()
            p.join(10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    first, second = results[0], results[1]
    assert first['in_sync'] and second['in_sync']
    assert all(torch.equal(a, b) for a, b in zip(first['params'], second['params']))
    assert all(l == l and abs(l) < 1e3 for l in first['losses'] + second['losses'])
    # the two ranks saw different pairs: their losses differ, the averaged update is shared
    assert first['losses'][0] != second['losses'][0]
    expected = single_process_reference(torch.device('cuda:0'))
    initial = single_process_initial()
    moved = 0.0
    for got, want, start in zip(first['params'], expected, initial):
        # same arithmetic up to the order of the two-term gradient sum (exact) and RMSprop's elementwise math
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), float((got - want).abs().max())
        moved = max(moved, float((got - start).abs().max()))
    assert moved > 1e-3, 'the optimizer steps must have changed the parameters'


def single_process_initial():
    import practicaldeepstereo_nips2018_amd as pds
    torch.manual_seed(0)
    return [p.detach().clone() for p in pds.PdsNetwork.default(MAX_DISPARITY).parameters()]


def _run_bench_two_ranks(extra):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), both ranks on
    cuda:0 through gloo (the box has one GPU); returns the parsed JSON line rank 0 printed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--share-device', '--backend',
           'gloo'] + extra
    out = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [l for l in out.stdout.decode(errors='replace').splitlines() if l.startswith('{')]
    assert out.returncode == 0 and lines, out.stderr.decode(errors='replace')[-2000:]
    return json.loads(lines[-1])


def test_bench_train_line_two_ranks(hip_library):
    """`bench.py --train` with world > 1 died on an undefined name right after its timed steps and never
    printed its line.  The config-5 line of two ranks: one JSON line, weak scaling, replicas in sync."""
    line = _run_bench_two_ranks(['--train', '--steps', '2', '--warmup', '1'])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['unit'] == 'pairs/s', line
    assert line['value'] > 0 and line['replicas_in_sync'] is True, line
    assert line['config']['collectives']['gather_mode'] in ('separate', 'single', 'coalesced'), line


def test_bench_sharded_line_two_ranks(hip_library):
    """The disparity-sharded line of two ranks (configs[2] with N = 2): bit-identical to the unsharded hot path, latency and
    replica modes reported beside it, the collectives pre-flight recorded."""
    line = _run_bench_two_ranks(['--steps', '8', '--warmup', '2', '--windows', '2'])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['sharded_equals_unsharded'] is True, line
    assert line['latency_mode']['ms_per_frame'] > 0 and line['replica_mode']['value'] > 0, line
    shards = line['config']['collectives']['gather_forms_tried']
    assert '[1, 8, 24, 144, 240]' in shards and '[4, 8, 32, 96, 320]' in shards, shards
