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


# How the signatures are all-gathered (chosen once per process, by `preflight_collectives` or at the first gather):
#   'coalesced'  batch * C all_gather_into_tensor calls on contiguous per-channel views of the FINAL [B, C, D, h, w]
#                tensor, issued as one group (ncclGroupStart / End): one launch, no staging buffer, no re-layout copy
#   'separate'   the same collectives one by one (every backend: gloo in the CPU tests)
#   'single'     ONE all_gather_into_tensor into a rank-major staging buffer + a permuting copy (round 2's form: the
#                most conservative use of the library)
_GATHER_MODE = None     # the form chosen last (what the bench line reports)
_GATHER_NOTE = 'not decided yet (no gather has run)'
_GATHER_MODES = {}      # (process group, device type) -> form: a CPU / gloo pre-flight must not decide for an RCCL group


def _device_backend(group, device):
    """Name of the backend that serves `device` in this group ("nccl" is RCCL).  `dist.get_backend` names the group's
    configuration, which for a multi-backend group ("cpu:gloo,cuda:nccl") is not the backend a CUDA tensor uses."""
    try:
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        return pg._get_backend(torch.device(device)).name().lower()
    except Exception:   # older / different process-group objects: fall back to the configured name
        name = str(dist.get_backend(group)).lower()
        if ':' in name:   # "cpu:gloo,cuda:nccl"
            kind = 'cuda' if torch.device(device).type == 'cuda' else 'cpu'
            parts = dict(item.split(':') for item in name.split(','))
            return parts.get(kind, name)
        return name


def _views(out, local_planes):
    batch, channels = local_planes.shape[:2]
    return [(out[b, c].view(-1), local_planes[b, c].view(-1)) for b in range(batch) for c in range(channels)]


def _gather_coalesced(out, local_planes, group):
    # ProcessGroup.allgather_into_tensor_coalesced is what torch's (private) _coalescing_manager itself ends in for
    # all_gather_into_tensor; calling it directly leaves no half-open coalescing state behind when it raises
    # (the context manager has no try / finally, a failure inside it poisoned every later collective)
    pg = group if group is not None else dist.group.WORLD
    views = _views(out, local_planes)
    work = pg.allgather_into_tensor_coalesced([whole for whole, _ in views], [mine for _, mine in views])
    if work is not None:
        work.wait()


def _gather_separate(out, local_planes, group):
    for whole, mine in _views(out, local_planes):
        dist.all_gather_into_tensor(whole, mine, group=group)


def _gather_single(out, local_planes, group):
    world_size = dist.get_world_size(group)
    batch, channels, d_local, h, w = local_planes.shape
    staging = local_planes.new_empty((world_size, batch, channels, d_local, h, w))
    dist.all_gather_into_tensor(staging.view(-1), local_planes.view(-1), group=group)
    out.view(batch, channels, world_size, d_local, h, w).copy_(staging.permute(1, 2, 0, 3, 4, 5))


_GATHER_FORMS = {'coalesced': _gather_coalesced, 'separate': _gather_separate, 'single': _gather_single}


def _all_ranks_agree(ok, device, group):
    """True when `ok` holds on every rank (one tiny MIN all-reduce -- the most basic collective there is)."""
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)


def _preflight_shapes(world, on_gpu):
    """Per-rank shard shapes the pre-flight gathers: a tiny one (layout errors show at once), then the EXACT shards of
    BASELINE configs[2] ([1, 8, 48 / N, 144, 240]: per-channel blocks of 3.3 / 1.7 / 0.8 MB at N = 2 / 4 / 8) and of
    configs[3] ([4, 8, 64 / N, 96, 320]; batch 1 on a CPU group, where only the block size matters and gloo is slow) --
    a form can pass on a few hundred bytes and fail on real transfers (gloo's coalesced method with CUDA tensors did)."""
    shapes = [(2, 3, 2, 3, 5)]
    if 48 % world == 0:
        shapes.append((1, 8, 48 // world, 144, 240))
    if 64 % world == 0:
        shapes.append((4 if on_gpu else 1, 8, 64 // world, 96, 320))
    return shapes


def _choose_gather_mode(device, group):
    """Tries the forms in order of preference and takes the first one that runs AND reproduces the locally computed
    expectation on every rank, on every shape of `_preflight_shapes`.  An exception (of any type: RCCL reports API
    misuse as RuntimeError) or a wrong result on ANY rank moves all ranks on together: the agreement all-reduce runs
    after EVERY case, so a rank that fails a case locally and the ranks that did not still issue the same sequence of
    collectives (with one agreement per form a one-sided failure on the first case left the others inside
    the second case's gather).  PDS_FORCE_GATHER=coalesced|separate|single skips the search (recorded in the note)."""
    global _GATHER_MODE, _GATHER_NOTE
    import os
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = _device_backend(group, device)
    forced = os.environ.get('PDS_FORCE_GATHER', '').strip().lower()
    if forced:
        if forced not in _GATHER_FORMS:
            raise RuntimeError('PDS_FORCE_GATHER=%s: not one of %s' % (forced, ', '.join(sorted(_GATHER_FORMS))))
        order = [forced]
    else:
        # (the coalesced form is RCCL's ncclGroup; gloo offers the same method, but with CUDA tensors its result was wrong
        # on real-size tensors although the tiny self-check passed -- tests/test_gpu_sharded.py -- so gloo keeps the
        # separate calls)
        order = ['coalesced', 'separate', 'single'] if backend == 'nccl' else ['separate', 'single']

    def shard(r, shape):
        n = 1
        for v in shape:
            n *= v
        base = (torch.arange(n, dtype=torch.float32) % 8191.0).view(shape)
        return base + 10000.0 * (r + 1)
    shapes = _preflight_shapes(world, torch.device(device).type == 'cuda')
    tried = []
    for mode in order:
        agreed, why = True, ''
        for shape in shapes:
            ok = True
            try:
                mine = shard(rank, shape).to(device).contiguous()
                expect = torch.cat([shard(r, shape) for r in range(world)], dim=2).to(device)
                out = mine.new_full((shape[0], shape[1], world * shape[2], shape[3], shape[4]), float('nan'))
                _GATHER_FORMS[mode](out, mine, group)
                if out.is_cuda:
                    torch.cuda.synchronize(out.device)
                ok = bool(torch.equal(out, expect))
                why = '' if ok else 'wrong result at shard %s' % (list(shape),)
                del mine, expect, out
            except Exception as error:   # noqa: BLE001  (any failure of this form means: use the next one)
                ok, why = False, '%s at shard %s: %s' % (type(error).__name__, list(shape), str(error).split('\n')[0][:120])
            agreed = _all_ranks_agree(ok, device, group)   # after EVERY case: the same collective sequence on every rank
            if not agreed:
                break
        tried.append('%s %s' % (mode, 'ok' if agreed else ('failed (%s)' % (why or 'on another rank'))))
        if agreed:
            _GATHER_MODE = mode
            _GATHER_NOTE = 'backend %s; %s%s; shards checked %s' % (
                backend, 'PDS_FORCE_GATHER; ' if forced else '', ', '.join(tried), [list(v) for v in shapes])
            _GATHER_MODES[(group, torch.device(device).type)] = mode
            return mode
    raise RuntimeError('no form of the all-gather works on this process group: %s' % ', '.join(tried))


def _communicator_facts(group, device):
    """What the communication library ITSELF says about the communicator behind this group (RCCL only): ncclCommCount /
    ncclCommUserRank / ncclCommCuDevice through the handle torch exposes -- so the bench line can confirm that RCCL, not
    just torch.distributed, saw N ranks (the first multi-GPU run of this code is the driver's)."""
    import ctypes
    import glob
    import os
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    backend = pg._get_backend(torch.device(device))
    handle = int(backend._comm_ptr())
    if not handle:
        return {'nranks_seen': None, 'note': 'communicator not created yet'}
    found = glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so*'))
    lib = ctypes.CDLL(found[0] if found else 'librccl.so')
    facts = {}
    for key, name in (('nranks_seen', 'ncclCommCount'), ('user_rank', 'ncclCommUserRank'), ('device', 'ncclCommCuDevice')):
        value = ctypes.c_int(-1)
        fn = getattr(lib, name)
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        fn.restype = ctypes.c_int
        rc = fn(ctypes.c_void_p(handle), ctypes.byref(value))
        facts[key] = int(value.value) if rc == 0 else 'error %d' % rc
    return facts


def preflight_collectives(group=None, device=None):
    """Self-check of the collectives this package uses, on tiny tensors against locally computed expectations, BEFORE
    anything is timed: all_reduce (SUM, MAX), then the all-gather forms (the form chosen is the one every later
    gather uses).  Returns a description for the bench line; raises on failure (same outcome on every rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError('torch.distributed is not initialised')
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    device = torch.device(device)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    t = torch.tensor([float(rank + 1), 2.0], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if t.tolist() != [world * (world + 1) / 2.0, 2.0 * world]:
        raise RuntimeError('all_reduce(SUM) returned %s on rank %d of %d' % (t.tolist(), rank, world))
    t = torch.tensor([float(rank)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if t.item() != world - 1:
        raise RuntimeError('all_reduce(MAX) returned %s on rank %d of %d' % (t.item(), rank, world))
    mode = _choose_gather_mode(device, group)
    info = {'all_reduce': 'ok', 'gather_mode': mode, 'gather_forms_tried': _GATHER_NOTE,
            'backend': _device_backend(group, device), 'world_size': world}
    if device.type == 'cuda':
        try:
            info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as error:   # noqa: BLE001
            info['rccl_version'] = 'unknown (%s)' % type(error).__name__
        try:   # RCCL's own view of the communicator (ncclCommCount): every rank's count, MIN / MAX over the ranks
            facts = _communicator_facts(group, device)
            seen = facts.get('nranks_seen')
            t = torch.tensor([float(seen) if isinstance(seen, int) else -1.0], device=device, dtype=torch.float64)
            lo, hi = t.clone(), t.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            facts['nranks_seen_min_over_ranks'] = int(lo.item())
            facts['nranks_seen_max_over_ranks'] = int(hi.item())
            info.update(facts)
        except Exception as error:   # noqa: BLE001  (diagnostics only: never take the run down)
            info['nranks_seen'] = 'unknown (%s: %s)' % (type(error).__name__, str(error)[:80])
    import os
    info['knobs'] = {k: os.environ[k] for k in sorted(os.environ) if k.startswith(('NCCL_', 'RCCL_', 'PDS_FORCE_'))}
    return info


def _gather_planes_raw(local_planes, group):
    """[batch, C, D_local, h, w] shards -> [batch, C, world * D_local, h, w], written in its FINAL layout: for every
    (batch entry, channel) the D_local planes of a rank are one contiguous block, and the world blocks of that
    (batch entry, channel) are contiguous in rank order -- exactly what ``all_gather_into_tensor`` produces.  So the
    gather is batch * C collectives on contiguous views of the output (8 at batch 1), issued as ONE coalesced group
    on RCCL (ncclGroupStart / End: one launch, every rank sends its block straight to its 7 xGMI peers), and no
    re-layout copy of the 53 MB result runs afterwards.  The form is decided ONCE per process by a self-check on a tiny
    tensor (`_choose_gather_mode`; no multi-GPU node was available to the build, so nothing about RCCL is assumed)."""
    local_planes = local_planes.contiguous()
    world_size = dist.get_world_size(group)
    batch, channels, d_local, h, w = local_planes.shape
    out = local_planes.new_empty((batch, channels, world_size * d_local, h, w))
    mode = _GATHER_MODES.get((group, local_planes.device.type)) or _choose_gather_mode(local_planes.device, group)
    _GATHER_FORMS[mode](out, local_planes, group)
    return out


def gather_description(group=None):
    """What the bench line records about the collective (config.parallelism): the form ACTUALLY in use."""
    forms = {'coalesced': "batch x 8 all_gather_into_tensor calls on contiguous per-channel views of the [B, 8, D', h, w] "
                          'result, coalesced into one group (no staging buffer, no re-layout copy)',
             'separate': "batch x 8 separate all_gather_into_tensor calls on contiguous per-channel views of the "
                         "[B, 8, D', h, w] result (no staging buffer, no re-layout copy)",
             'single': 'one all_gather_into_tensor into a rank-major staging buffer + one permuting copy',
             None: 'form not decided yet'}
    return '%s [%s]' % (forms[_GATHER_MODE], _GATHER_NOTE)


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
        return gather_planes(self.local_planes(left_embedding, right_embedding), self._group)

    def local_planes(self, left_embedding, right_embedding):
        """This rank's slice [batch, C, D' / N, h, w] of the signatures -- the input of the all-gather."""
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
            return self._matching(left_embedding, right_embedding)
        finally:
            self._matching.set_disparity_shard(None)


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
        # GPU under-used, the next pair fills it.  The collectives do NOT ride on those streams: every all-gather of
        # this object is enqueued on ONE dedicated stream, by the one host thread, in submission order -- the same
        # order on every rank by construction, whatever the process group does internally -- and events tie it to
        # the lane that produced its input and consumes its result (`_ordered_gather`).
        self._lanes = PairStreams(self._whole_pair, streams=streams) if streams > 1 else None
        self._gather_stream = None
        self.gathers_issued = 0      # (diagnostics / tests: collectives issued so far, identical on every rank)

    def _ordered_gather(self, local_planes):
        """all-gather of this rank's planes on the dedicated collective stream:
        lane --(event: planes produced)--> gather stream: all-gather --(event: gathered)--> lane."""
        self.gathers_issued += 1
        # (the raw gather is not differentiable: this schedule is the inference path -- ADVICE r5)
        assert not (torch.is_grad_enabled() and local_planes.requires_grad), \
            'ShardedHotPath with streams > 1 is inference-only: call it under torch.no_grad()'
        if not local_planes.is_cuda:
            return _gather_planes_raw(local_planes, self._group)
        device = local_planes.device
        lane = torch.cuda.current_stream(device)
        if self._gather_stream is None:
            self._gather_stream = torch.cuda.Stream(device)
        produced = torch.cuda.Event()
        produced.record(lane)
        self._gather_stream.wait_event(produced)
        with torch.cuda.stream(self._gather_stream):
            gathered = _gather_planes_raw(local_planes, self._group)
            done = torch.cuda.Event()
            done.record(self._gather_stream)
        local_planes.record_stream(self._gather_stream)   # allocated on the lane, read on the gather stream
        gathered.record_stream(lane)                      # allocated on the gather stream, read on the lane
        lane.wait_event(done)
        return gathered

    def _whole_pair(self, index, left_embedding, right_embedding, shortcut_from_left):
        rank, world = self._world()
        if world == 1:
            signatures = self._sharded(left_embedding, right_embedding)
        else:
            signatures = self._ordered_gather(self._sharded.local_planes(left_embedding, right_embedding))
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
