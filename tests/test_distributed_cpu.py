"""CPU, world_size 2, gloo: the disparity-axis sharding and its single all-gather.

The HIP kernels cannot run here, so the sharded module is exercised with a CPU stand-in for the
per-rank Matching (the oracle restricted to the rank's planes); what is under test is the partition
arithmetic, the gather layout and that ShardedMatching returns the unsharded result on every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pds_oracle as oracle
from practicaldeepstereo_nips2018_amd import distributed as pdist


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def mock_operation(x):
    return torch.max(x, dim=1, keepdim=True)[0] + x[:, :3].sum(1, keepdim=True)


class CpuShardMatching(torch.nn.Module):
    """Stand-in with the interface ShardedMatching drives (set_disparity_shard, _maximum_disparity)."""

    def __init__(self, maximum_disparity):
        super().__init__()
        self._maximum_disparity = maximum_disparity
        self._shard = None

    def set_maximum_disparity(self, n):
        self._maximum_disparity = n

    def set_disparity_shard(self, shard):
        self._shard = shard

    def forward(self, left, right):
        begin, count = self._shard if self._shard is not None else (0, self._maximum_disparity + 1)
        planes = [mock_operation(torch.cat([left, oracle.shift_right(right, d)], 1)) for d in range(begin, begin + count)]
        return torch.stack(planes, dim=2)


def worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        left = torch.randn(2, 5, 6, 9, generator=g)
        right = torch.randn(2, 5, 6, 9, generator=g)
        expected = oracle.matching(left, right, 15, mock_operation)
        sharded = pdist.ShardedMatching(CpuShardMatching(15))
        out = sharded(left, right)
        ok = torch.equal(out, expected)
        # gather layout on its own: rank r owns planes [r*D/N, (r+1)*D/N)
        begin, count = pdist.shard_range(16, rank, world)
        ok = ok and torch.equal(pdist.gather_planes(expected[:, :, begin:begin + count].contiguous()), expected)
        # the wrapped module is left unsharded afterwards
        ok = ok and sharded._matching._shard is None
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def mock_tail(signatures, shortcut):
    return (signatures * shortcut.unsqueeze(2)).sum(dim=(1, 2))


def pipeline_worker(rank, world, port, results):
    """ShardedHotPath: pair i's tail runs on rank i % world only and equals the unsharded result."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        hot_path = pdist.ShardedHotPath(CpuShardMatching(15), mock_tail)
        ok = True
        for i in range(5):
            g = torch.Generator().manual_seed(10 + i)
            left = torch.randn(1, 5, 6, 9, generator=g)
            right = torch.randn(1, 5, 6, 9, generator=g)
            shortcut = torch.randn(1, 1, 6, 9, generator=g)
            out = hot_path.submit(left, right, shortcut)
            if i % world == rank:
                expected = mock_tail(oracle.matching(left, right, 15, mock_operation), shortcut)
                ok = ok and out is not None and torch.equal(out, expected)
            else:
                ok = ok and out is None
            ok = ok and hot_path.owner_of(i) == i % world
        hot_path.drain()
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_hot_path_round_robin_tail(world):
    if 16 % world != 0:
        pytest.skip('16 planes do not divide over %d ranks' % world)
    ctx = mp.get_context('spawn')
    results = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=pipeline_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(results.get(r) for r in range(world)), dict(results)


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_matching_equals_unsharded(world):
    ctx = mp.get_context('spawn')
    results = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(results.get(r) for r in range(world)), dict(results)


def test_shard_range_partition():
    for planes in (16, 48, 64):
        for world in (1, 2, 4, 8):
            covered = []
            for rank in range(world):
                begin, count = pdist.shard_range(planes, rank, world)
                covered.extend(range(begin, begin + count))
            assert covered == list(range(planes))
    with pytest.raises(ValueError):
        pdist.shard_range(48, 0, 5)
    with pytest.raises(ValueError):
        pdist.shard_range(48, 3, 2)


def test_single_process_passthrough():
    sharded = pdist.ShardedMatching(CpuShardMatching(3))
    left, right = torch.randn(1, 2, 3, 5), torch.randn(1, 2, 3, 5)
    assert torch.equal(sharded(left, right), oracle.matching(left, right, 3, mock_operation))


def test_pair_streams_runs_inline_on_cpu():
    calls = []

    def hot_path(a, b):
        calls.append(1)
        return a + b

    streams = pdist.PairStreams(hot_path, streams=3)
    outs = [streams.submit(torch.full((2,), float(i)), torch.ones(2)) for i in range(4)]
    streams.drain()
    assert len(calls) == 4 and all(torch.equal(o, torch.full((2,), float(i) + 1)) for i, o in enumerate(outs))
