"""Disparity-axis sharding of Matching across the GPUs of one node.

The reference has no multi-device code; this is new design.  Every disparity plane of Matching is
an independent 2-D network evaluation with its own InstanceNorm statistics (reference
practical_deep_stereo/matching.py:56-62), so rank r of N computes planes
[r*D'/N, (r+1)*D'/N) with ``pds_matching_fwd(d_begin, d_count)`` and ONE all-gather (RCCL over xGMI
when the process group's backend is "nccl") reassembles the compact signatures
[batch, 8, D', h, w] on every rank.  Regularization and the estimator couple all planes (3-D
convolutions, volume-wide InstanceNorm, arg-max) and run replicated after the gather.

One process per GPU; ``torch.distributed`` must be initialised by the caller (torchrun env).
"""
import torch
import torch.distributed as dist
from torch import nn


def shard_range(number_of_planes, rank, world_size):
    """(first_plane, number_of_planes_on_this_rank); planes must divide evenly (D' is a multiple of
    16 by network.py:28, so 2, 4 and 8 ranks always do)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('bad rank %d of %d' % (rank, world_size))
    if number_of_planes % world_size != 0:
        raise ValueError('%d disparity planes do not divide over %d ranks' %
                         (number_of_planes, world_size))
    per_rank = number_of_planes // world_size
    return rank * per_rank, per_rank


def _gather_planes_raw(local_planes, group):
    """[batch, C, D_local, h, w] shards -> [batch, C, world * D_local, h, w], written in its FINAL layout: for every
    (batch entry, channel) the D_local planes of a rank are one contiguous block, and the world blocks of that
    (batch entry, channel) are contiguous in rank order -- exactly what ``all_gather_into_tensor`` produces.  So the
    gather is batch * C collectives on contiguous views of the output (8 at batch 1), issued as ONE coalesced group
    on RCCL (ncclGroupStart / End: one launch, every rank sends its block straight to its 7 xGMI peers), and no
    re-layout copy of the 53 MB result runs afterwards (round 2 gathered rank-major and permuted)."""
    local_planes = local_planes.contiguous()
    world_size = dist.get_world_size(group)
    batch, channels, d_local, h, w = local_planes.shape
    out = local_planes.new_empty((batch, channels, world_size * d_local, h, w))
    pairs = [(out[b, c].view(-1), local_planes[b, c].view(-1)) for b in range(batch) for c in range(channels)]
    global _COALESCE_OK
    if local_planes.is_cuda and dist.get_backend(group) == 'nccl' and len(pairs) > 1 and _COALESCE_OK:
        # (RCCL has never run this code in the build container -- one GPU at most -- so the grouped form is guarded: an
        # API error of the coalescing manager is the same on every rank, and every rank then takes the plain form below,
        # which re-issues all of the collectives)
        try:
            from torch.distributed.distributed_c10d import _coalescing_manager
            with _coalescing_manager(group=group, device=local_planes.device, async_ops=False):
                for whole, mine in pairs:
                    dist.all_gather_into_tensor(whole, mine, group=group)
            return out
        except (ImportError, TypeError, AttributeError, NotImplementedError) as error:
            _COALESCE_OK = False
            import warnings
            warnings.warn('grouped all-gather unavailable (%s: %s); using %d separate collectives per pair'
                          % (type(error).__name__, error, len(pairs)))
    for whole, mine in pairs:
        dist.all_gather_into_tensor(whole, mine, group=group)
    return out


_COALESCE_OK = True


def gather_description(group=None):
    """What the bench line records about the collective (config.parallelism): its form and the RCCL knobs in force."""
    import os
    knobs = ', '.join('%s=%s' % (k, os.environ[k]) for k in ('NCCL_ALGO', 'NCCL_PROTO', 'NCCL_MIN_NCHANNELS',
                                                                'NCCL_MAX_NCHANNELS', 'RCCL_MSCCL_ENABLE')
                      if k in os.environ)
    return ('batch x 8 all_gather_into_tensor calls on contiguous per-channel views of the [B, 8, D\', h, w] result, '
            'coalesced into one group (no staging buffer, no re-layout copy); RCCL knobs: %s' % (knobs or 'library defaults'))


class _GatherPlanes(torch.autograd.Function):
    """Differentiable all-gather along the disparity axis.  What follows the gather (Regularization, the loss) is
    REPLICATED on every rank, so every rank holds the same upstream gradient and the adjoint of the gather is the
    rank's own slice of it -- no collective in backward (the cross-rank sums happen where the replicated region is
    entered: ``_ReplicatedInput`` for the descriptors, gradient hooks for the wrapped module's parameters)."""

    @staticmethod
    def forward(ctx, local_planes, group):
        ctx.group = group
        ctx.d_local = local_planes.shape[2]
        return _gather_planes_raw(local_planes, group)

    @staticmethod
    def backward(ctx, grad_gathered):
        begin = dist.get_rank(ctx.group) * ctx.d_local
        return grad_gathered[:, :, begin:begin + ctx.d_local].contiguous(), None


class _ReplicatedInput(torch.autograd.Function):
    """Identity on the way into the sharded region; in backward the per-rank partial gradients (each rank saw only
    its disparity planes) are summed over the ranks, so whatever produced the input -- the descriptor network --
    receives the full gradient on every rank."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=ctx.group)
        return grad, None


def gather_planes(local_planes, group=None):
    """All-gathers [batch, C, D_local, h, w] shards along dim 2 in rank order.

    One coalesced group of collectives straight into the [batch, C, D, h, w] layout Regularization consumes (see
    ``_gather_planes_raw``).  Differentiable: the gradient of a shard is the matching slice of the gradient of the
    gathered tensor, which MUST be replicated (identical on every rank) -- i.e. everything after the gather up to
    the loss has to run on every rank on the same data; a rank-dependent tail (``ShardedHotPath`` runs the tail of
    pair i on rank i % N only) is an inference schedule and must not be differentiated through."""
    world_size = dist.get_world_size(group)
    if world_size == 1:
        return local_planes
    if torch.is_grad_enabled() and local_planes.requires_grad:
        return _GatherPlanes.apply(local_planes, group)
    return _gather_planes_raw(local_planes, group)


class ShardedMatching(nn.Module):
    """Wraps a ``Matching`` module so that each rank evaluates its slice of the disparity range and
    the full set of matching signatures is reassembled with one all-gather.  Same call signature
    and result as the wrapped module (matching.py:34-63).

    Training: the region after the gather is replicated, so with gradients enabled (i) the gather's
    adjoint hands every rank its own slice, (ii) the gradients of the two descriptor inputs are summed
    over the ranks on the way out and (iii) the wrapped module's parameter gradients -- partial sums
    over the rank's planes -- are all-reduced by gradient hooks, all in autograd order, identical on
    every rank.  After ``backward()`` every rank holds the full gradients of the unsharded network.
    PRECONDITION: the computation between the gather and the loss is replicated -- same code, same data on every
    rank -- so that the gradient arriving at the gather is identical everywhere (it is not checked).

    State dict: the wrapper adds no level to the key path (``net._matching = ShardedMatching(net._matching)``
    keeps the reference's ``_matching._operation...`` keys), so reference checkpoints load into, and
    are saved from, a wrapped network unchanged."""

    def __init__(self, matching_module, group=None):
        super(ShardedMatching, self).__init__()
        self._matching = matching_module
        self._group = group
        self._register_state_dict_hook(self._strip_wrapper_prefix)
        self._register_load_state_dict_pre_hook(self._add_wrapper_prefix)

    @staticmethod
    def _strip_wrapper_prefix(module, state_dict, prefix, local_metadata):
        inner = prefix + '_matching.'
        for key in [k for k in state_dict if k.startswith(inner)]:
            state_dict[prefix + key[len(inner):]] = state_dict.pop(key)
        return state_dict

    @staticmethod
    def _add_wrapper_prefix(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        inner = prefix + '_matching.'
        for key in [k for k in state_dict if k.startswith(prefix) and not k.startswith(inner)]:
            state_dict[inner + key[len(prefix):]] = state_dict.pop(key)

    def set_maximum_disparity(self, maximum_disparity):
        self._matching.set_maximum_disparity(maximum_disparity)

    def _hook_parameter_gradients(self):
        """All-reduce hooks on the INNER module's parameters, registered once per (inner module, group): the handles
        and the group live on the inner module, so wrapping the same Matching again (another wrapper object, or
        another group) re-uses or replaces them instead of stacking a second all-reduce (which would multiply the
        gradients by the world size)."""
        inner = self._matching
        state = getattr(inner, '_pds_grad_hooks', None)
        if state is not None and state[0] is self._group:
            return
        if state is not None:
            for handle in state[1]:
                handle.remove()
        group = self._group

        def reduce_over_ranks(grad):
            if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
                return grad
            grad = grad.contiguous().clone()
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
            return grad
        handles = [p.register_hook(reduce_over_ranks) for p in inner.parameters()]
        inner._pds_grad_hooks = (group, handles)

    def forward(self, left_embedding, right_embedding):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self._group) == 1:
            return self._matching(left_embedding, right_embedding)
        planes = self._matching._maximum_disparity + 1
        shard = shard_range(planes, dist.get_rank(self._group), dist.get_world_size(self._group))
        if torch.is_grad_enabled():
            if left_embedding.requires_grad:
                left_embedding = _ReplicatedInput.apply(left_embedding, self._group)
            if right_embedding.requires_grad:
                right_embedding = _ReplicatedInput.apply(right_embedding, self._group)
            if any(p.requires_grad for p in self._matching.parameters()):
                self._hook_parameter_gradients()
        self._matching.set_disparity_shard(shard)
        try:
            local = self._matching(left_embedding, right_embedding)
        finally:
            self._matching.set_disparity_shard(None)
        return gather_planes(local, self._group)


class ShardedHotPath(object):
    """The whole hot path over N GPUs for a STREAM of stereo pairs.

    Per pair: every rank evaluates its slice of the disparity planes and ONE all-gather reassembles the
    matching signatures (exactly ``ShardedMatching``).  What cannot be sharded -- Regularization and the
    estimator couple all planes -- is not replicated N times: the tail of pair ``i`` runs on rank
    ``i % N`` only, on a side HIP stream, while every rank already matches pair ``i + 1`` on its main
    stream.  Per pair and rank that is Matching / N + tail / N of GPU time instead of Matching / N + tail,
    and the latency of one pair stays Matching / N + all-gather + tail.

    ``tail(signatures, shortcut)`` is any callable, normally
    ``lambda s, c: regularization.forward_with_estimator(s, c, estimator)``.  ``submit`` returns the
    pair's result on its owner rank (a tensor that is valid once ``drain()`` -- or a wait on the side
    stream -- has returned) and ``None`` on the other ranks.
    """

    def __init__(self, matching_module, tail, group=None, max_pending=4, streams=1):
        self._sharded = ShardedMatching(matching_module, group)
        self._tail = tail
        self._group = group
        self._max_pending = max(1, int(max_pending))
        self._submitted = 0
        self._side = None
        self._pending = []   # completion events of this rank's unfinished tails (GPU only)
        # streams > 1: whole pairs (this rank's planes, the all-gather and, on the owner, the tail) are dealt
        # round-robin to that many HIP streams (PairStreams below): the small per-rank shards of one pair leave the
        # GPU under-used, the next pair fills it.  Collectives stay in program order on every rank (they are issued
        # by one host thread and funnel through the process group's communication stream).
        self._lanes = PairStreams(self._whole_pair, streams=streams) if streams > 1 else None

    def _whole_pair(self, index, left_embedding, right_embedding, shortcut_from_left):
        rank, world = self._world()
        signatures = self._sharded(left_embedding, right_embedding)
        return self._tail(signatures, shortcut_from_left) if index % world == rank else None

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self._group), dist.get_world_size(self._group)
        return 0, 1

    def owner_of(self, pair_index):
        return pair_index % self._world()[1]

    def submit(self, left_embedding, right_embedding, shortcut_from_left):
        rank, world = self._world()
        index = self._submitted
        self._submitted += 1
        if self._lanes is not None and left_embedding.is_cuda:
            return self._lanes.submit(index, left_embedding, right_embedding, shortcut_from_left)
        signatures = self._sharded(left_embedding, right_embedding)
        if index % world != rank:
            return None
        if not signatures.is_cuda:
            return self._tail(signatures, shortcut_from_left)
        device = signatures.device
        main = torch.cuda.current_stream(device)
        if self._side is None:
            self._side = torch.cuda.Stream(device)
        # bounded backlog: the main stream waits for this rank's oldest unfinished tail before running ahead
        while len(self._pending) >= self._max_pending:
            main.wait_event(self._pending.pop(0))
        ready = torch.cuda.Event()
        ready.record(main)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            result = self._tail(signatures, shortcut_from_left)
            done = torch.cuda.Event()
            done.record(self._side)
        # both were allocated on the main stream and are read on the side stream
        signatures.record_stream(self._side)
        shortcut_from_left.record_stream(self._side)
        self._pending.append(done)
        return result

    def drain(self):
        """Blocks the main stream (and the host) until every tail submitted on this rank has finished."""
        if self._lanes is not None:
            self._lanes.drain()
        if self._side is not None:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            self._side.synchronize()
        self._pending = []


class PairStreams(object):
    """Single-GPU schedule for a STREAM of stereo pairs: whole pairs round-robin over a few HIP streams.

    One pair does not fill an MI355X all the time: the 64-channel convolutions of Matching are MFMA-bound, but the
    factorised first layers, the streaming kernels and the small 3-D layers of Regularization are HBM- or
    latency-bound and leave most CUs idle.  With two or three pairs in flight on separate streams those phases run
    beside another pair's MFMA-bound kernels (measured at 960x540, D=192: 5.26 ms per pair on one stream, 4.64 on two,
    4.52 on three; results bit-identical).  Every module keeps one workspace per stream (``_lib.Workspace``), so
    concurrent pairs never share scratch memory.

    ``hot_path(*inputs)`` is any callable that enqueues the whole path on the current stream and returns its result;
    ``submit`` returns that result, valid once ``drain()`` (or a wait on its stream) has returned.
    """

    def __init__(self, hot_path, streams=3, max_ahead=2):
        self._hot_path = hot_path
        self._count = max(1, int(streams))
        self._streams = None
        self._events = None
        self._max_ahead = max(1, int(max_ahead))
        self._submitted = 0

    def submit(self, *inputs):
        index = self._submitted
        self._submitted += 1
        tensors = [t for t in inputs if isinstance(t, torch.Tensor)]
        if not tensors or not tensors[0].is_cuda:
            return self._hot_path(*inputs)
        device = tensors[0].device
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device) for _ in range(self._count)]
            self._events = [[] for _ in range(self._count)]
        lane = index % self._count
        stream, events = self._streams[lane], self._events[lane]
        while len(events) >= self._max_ahead:      # bounded run-ahead of the host per stream
            events.pop(0).synchronize()
        stream.wait_stream(torch.cuda.current_stream(device))   # the inputs were produced on the caller's stream
        with torch.cuda.stream(stream):
            result = self._hot_path(*inputs)
            done = torch.cuda.Event()
            done.record(stream)
        for t in tensors:
            t.record_stream(stream)
        events.append(done)
        return result

    def drain(self):
        """Makes the caller's stream (and the host) wait for every pair submitted so far."""
        if self._streams is not None:
            current = torch.cuda.current_stream(self._streams[0].device)
            for stream in self._streams:
                current.wait_stream(stream)
                stream.synchronize()
            self._events = [[] for _ in range(self._count)]
