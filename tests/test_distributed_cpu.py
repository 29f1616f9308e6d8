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


class LearnedShardMatching(torch.nn.Module):
    """Differentiable stand-in with a parameter: operation(x) = conv1x1(x) per plane (SURVEY.md 8e interface)."""

    def __init__(self, maximum_disparity, channels):
        super().__init__()
        self._maximum_disparity = maximum_disparity
        self._shard = None
        torch.manual_seed(7)
        self._operation = torch.nn.Conv2d(2 * channels, 3, 1)

    def set_disparity_shard(self, shard):
        self._shard = shard

    def forward(self, left, right):
        begin, count = self._shard if self._shard is not None else (0, self._maximum_disparity + 1)
        planes = [self._operation(torch.cat([left, oracle.shift_right(right, d)], 1)) for d in range(begin, begin + count)]
        return torch.stack(planes, dim=2)


def gradient_worker(rank, world, port, results):
    """Training through ShardedMatching: after backward every rank holds the gradients of the unsharded network --
    for the wrapped module's parameters AND for what produced its inputs (the gather used to cut the
    autograd graph)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        left0 = torch.randn(2, 4, 5, 7, generator=g)
        right0 = torch.randn(2, 4, 5, 7, generator=g)
        weight = torch.randn(2, 3, 8, 5, 7, generator=g)       # the replicated "tail": loss = sum(weight * signatures)
        torch.manual_seed(11)                                   # identical "descriptor network" on every rank
        upstream = torch.nn.Conv2d(4, 4, 1)
        torch.nn.init.normal_(upstream.weight)

        def run(matching):
            upstream.zero_grad()
            matching.zero_grad()
            left = upstream(left0)
            right = upstream(right0)
            loss = (matching(left, right) * weight).sum()
            loss.backward()
            return (loss.detach().clone(), upstream.weight.grad.clone(),
                    [p.grad.clone() for p in matching.parameters()])
        loss_ref, up_ref, params_ref = run(LearnedShardMatching(7, 4))
        loss, up, params = run(pdist.ShardedMatching(LearnedShardMatching(7, 4)))
        ok = torch.allclose(loss, loss_ref, rtol=1e-5)
        ok = ok and torch.allclose(up, up_ref, rtol=1e-4, atol=1e-5)
        ok = ok and all(torch.allclose(a, b, rtol=1e-4, atol=1e-5) for a, b in zip(params, params_ref))
        # wrapping the SAME inner module again (the hooks used to stack, multiplying the parameter
        # gradients by the world size) leaves one all-reduce per parameter
        shared = LearnedShardMatching(7, 4)
        run(pdist.ShardedMatching(shared))
        loss2, up2, params2 = run(pdist.ShardedMatching(shared))
        ok = ok and all(torch.allclose(a, b, rtol=1e-4, atol=1e-5) for a, b in zip(params2, params_ref))
        ok = ok and len(shared._pds_grad_hooks[1]) == len(list(shared.parameters()))
        # the wrapper adds no level to the state-dict key path (reference checkpoints stay loadable)
        inner = LearnedShardMatching(7, 4)
        wrapped = pdist.ShardedMatching(inner)
        ok = ok and sorted(wrapped.state_dict().keys()) == sorted(inner.state_dict().keys())
        wrapped.load_state_dict({k: v + 1.0 for k, v in inner.state_dict().items()})
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_matching_is_differentiable():
    world = 2
    ctx = mp.get_context('spawn')
    results = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=gradient_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(results.get(r) for r in range(world)), dict(results)


def preflight_worker(rank, world, port, results):
    """preflight_collectives: the self-check picks a gather form that reproduces the expectation on every rank, a
    form that fails is abandoned by ALL ranks together, and every form that runs here gives the layout."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        info = pdist.preflight_collectives(device='cpu')
        ok = info['gather_mode'] in ('separate', 'single') and info['backend'] == 'gloo'
        ok = ok and info['all_reduce'] == 'ok' and info['gather_mode'] in pdist.gather_description()
        g = torch.Generator().manual_seed(21)
        full = torch.randn(2, 8, 4 * world, 3, 5, generator=g)
        mine = full[:, :, 4 * rank:4 * rank + 4].contiguous()
        forms = ['separate', 'single', 'coalesced']   # (gloo offers the coalesced method for CPU tensors too: the layout is checked)
        for mode in forms:
            out = torch.full_like(full, float('nan'))
            pdist._GATHER_FORMS[mode](out, mine, None)
            ok = ok and torch.equal(out, full)
        # a failing 'coalesced' form must leave the process group usable (it is tried first on RCCL; here the order is
        # forced): the chain moves on and later collectives work
        real_coalesced = pdist._GATHER_FORMS['coalesced']
        real_backend = pdist._device_backend
        def broken_coalesced(out, local, group):
            raise RuntimeError('injected failure in the coalesced form')
        pdist._GATHER_FORMS['coalesced'] = broken_coalesced
        pdist._device_backend = lambda group, device: 'nccl'
        try:
            pdist._GATHER_MODES.clear()
            mode = pdist._choose_gather_mode(torch.device('cpu'), None)
        finally:
            pdist._GATHER_FORMS['coalesced'] = real_coalesced
            pdist._device_backend = real_backend
        ok = ok and mode == 'separate' and 'coalesced failed' in pdist.gather_description()
        ok = ok and torch.equal(pdist.gather_planes(mine), full)
        # (i) a form that raises on EVERY rank (an API error -- RCCL reports misuse as RuntimeError -- is the same
        # everywhere) and (ii) a form that completes but leaves a wrong result on ONE rank: all ranks must move on to
        # the next form together
        real = pdist._GATHER_FORMS['separate']
        for kind in ('raises', 'wrong on rank 1'):
            def broken(out, local, group, kind=kind):
                if kind == 'raises':
                    raise RuntimeError('injected failure')
                real(out, local, group)
                if rank == 1:
                    out.view(-1)[3] += 1.0
            pdist._GATHER_FORMS['separate'] = broken
            pdist._GATHER_FORMS['coalesced'] = broken
            try:
                pdist._GATHER_MODES.clear()
                mode = pdist._choose_gather_mode(torch.device('cpu'), None)
            finally:
                pdist._GATHER_FORMS['separate'] = real
                pdist._GATHER_FORMS['coalesced'] = real_coalesced
            ok = ok and mode == 'single' and 'separate failed' in pdist.gather_description()
        ok = ok and mode == 'single' and 'separate failed' in pdist.gather_description()
        ok = ok and torch.equal(pdist.gather_planes(mine), full)
        # (iii) ADVICE r5: a form that raises on ONE rank on the FIRST case only, and (iv) a form that is right on the tiny
        # case and wrong on the real-size shards (gloo's coalesced method with CUDA tensors did this): the agreement runs
        # after every case, so no rank is left inside a collective the others skipped, and all move on together
        for kind in ('rank 0 raises on the first case', 'wrong at real size'):
            calls = [0]
            def broken(out, local, group, kind=kind, calls=calls):
                calls[0] += 1
                real(out, local, group)
                # (after the collective: a rank that raises BEFORE entering a collective its peers are already in cannot be
                # rescued by any protocol; an error surfacing after it -- an asynchronous RCCL error, a failed check -- can)
                if kind.startswith('rank 0') and rank == 0 and calls[0] == 1:
                    raise RuntimeError('injected one-sided failure')
                if kind.startswith('wrong') and local.numel() > 10000 and rank == 1:
                    out.view(-1)[7] += 1.0
            pdist._GATHER_FORMS['separate'] = broken
            try:
                pdist._GATHER_MODES.clear()
                mode = pdist._choose_gather_mode(torch.device('cpu'), None)
            finally:
                pdist._GATHER_FORMS['separate'] = real
            ok = ok and mode == 'single' and 'separate failed' in pdist.gather_description()
            ok = ok and torch.equal(pdist.gather_planes(mine), full)
        # the exact shards of configs[2] / configs[3] are among the checked shapes
        shapes = pdist._preflight_shapes(world, False)
        ok = ok and (1, 8, 48 // world, 144, 240) in shapes and (1, 8, 64 // world, 96, 320) in shapes
        ok = ok and pdist._preflight_shapes(8, True)[1:] == [(1, 8, 6, 144, 240), (4, 8, 8, 96, 320)]
        # PDS_FORCE_GATHER: no search, recorded in the description and in the pre-flight's knobs
        os.environ['PDS_FORCE_GATHER'] = 'single'
        try:
            pdist._GATHER_MODES.clear()
            info = pdist.preflight_collectives(device=torch.device('cpu'))
        finally:
            del os.environ['PDS_FORCE_GATHER']
        ok = ok and info['gather_mode'] == 'single' and 'PDS_FORCE_GATHER' in pdist.gather_description()
        ok = ok and info['knobs'].get('PDS_FORCE_GATHER') == 'single'
        ok = ok and torch.equal(pdist.gather_planes(mine), full)
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_collective_preflight_and_gather_forms():
    world = 2
    ctx = mp.get_context('spawn')
    results = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=preflight_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(results.get(r) for r in range(world)), dict(results)
