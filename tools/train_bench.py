"""Training-step benchmark: BASELINE.json configs[4] (full-res 960x540, D=192 training step, data-parallel).

    python tools/train_bench.py [--steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/train_bench.py --gpus N

One step = what reference pds_trainer.py:36-46 + trainer.py's loop do per batch: train-mode PdsNetwork forward (cost
volume), SubpixelCrossEntropy against a ground truth with an unknown band, backward through every HIP module, RMSprop
(lr 1e-2, the recipe of train_on_flyingthings3d.py:68).  One pair per GPU (the reference trains with batch 1);
N > 1 wraps the network in DistributedDataParallel: gradients are all-reduced over RCCL (backend "nccl") while the
backward still runs, so scaling is weak.  Rank 0 prints one JSON line.  This is NOT the driver's bench.py contract
(that one measures the inference hot path of the metric); it makes the configs[4] number reproducible.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import practicaldeepstereo_nips2018_amd as pds  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--share-device', action='store_true', help='functional test: all ranks on cuda:0 (gloo)')
    ap.add_argument('--size', default='540x960x191', help='HxWxmaximum_disparity')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if args.share_device else int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)
    height, width, max_disparity = (int(v) for v in args.size.split('x'))
    from practicaldeepstereo_nips2018_amd.training import DataParallelTrainer, synthetic_example
    trainer = DataParallelTrainer(max_disparity, device, share_device=args.share_device)
    net = trainer.network
    left, right, truth = synthetic_example(height, width, max_disparity, 1 + rank, device)   # another pair per rank

    def step():
        return trainer.step(left, right, truth)

    losses = []
    for _ in range(args.warmup):
        losses.append(step())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step())
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        in_sync = trainer.replicas_in_sync()
    if rank == 0:
        values = [float(v) for v in losses]
        line = {'metric': 'training steps (stereo pairs)/sec, %dx%d D=%d, PdsNetwork train + SubpixelCrossEntropy + '
                          'backward + RMSprop' % (width, height, max_disparity + 1),
                'value': world * args.steps / elapsed, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
                'scaling': 'weak', 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': 'configs[4]: one %dx%d pair per GPU, D=%d, batch 1 per rank' %
                                       (width, height, max_disparity + 1),
                           'parallelism': 'DistributedDataParallel x%d' % world if world > 1 else 'single GPU'},
                'first_loss': values[0], 'last_loss': values[-1],
                'peak_memory_gb': torch.cuda.max_memory_allocated(device) / 2 ** 30}
        if world > 1:
            line['replicas_in_sync'] = in_sync
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
