"""The training step of BASELINE.json configs[4]: what reference pds_trainer.py:35-46 and the batch loop
of trainer.py do per example -- train-mode ``PdsNetwork`` forward (matching cost), ``SubpixelCrossEntropy`` against the
ground-truth disparity, backward through the HIP modules, one optimizer step (RMSprop, lr 1e-2:
train_on_flyingthings3d.py:66-68) -- plus the data-parallel wrapping the reference does not have: one process per GPU,
one stereo pair per rank, gradients averaged by ``DistributedDataParallel`` (RCCL when the backend is "nccl") while
backward still runs.  Used by ``bench.py --train``, ``tools/train_bench.py`` and tests/test_gpu_training_ddp.py."""
import torch
import torch.distributed as dist

from practicaldeepstereo_nips2018_amd.loss import SubpixelCrossEntropy
from practicaldeepstereo_nips2018_amd.network import PdsNetwork


def synthetic_example(height, width, maximum_disparity, seed, device):
    """(left, right, ground truth) of one synthetic pair: uniform images (SURVEY.md 8c recipe), uniform ground truth
    with a band of unknown (inf) disparities -- the masked case of loss.py:52-60."""
    g = torch.Generator().manual_seed(seed)
    left = (torch.rand(1, 3, height, width, generator=g) * 255).to(device)
    right = (torch.rand(1, 3, height, width, generator=g) * 255).to(device)
    truth = (torch.rand(1, height, width, generator=g) * (maximum_disparity - 1)).to(device)
    truth[:, :max(1, height // 32)] = float('inf')
    return left, right, truth


class DataParallelTrainer(object):
    """Seed-0 ``PdsNetwork.default(maximum_disparity)`` in train mode on ``device``, wrapped in DistributedDataParallel
    when a process group with more than one rank is initialised (``share_device``: every rank on the same GPU -- the
    functional test on a one-GPU box, gloo)."""

    def __init__(self, maximum_disparity, device, learning_rate=1e-2, share_device=False):
        torch.manual_seed(0)                      # identical initial weights on every rank
        self.network = PdsNetwork.default(maximum_disparity).to(device).train()
        self.model = self.network
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self.world > 1:
            from torch.nn.parallel import DistributedDataParallel
            self.model = DistributedDataParallel(self.network, device_ids=None if share_device else [device.index])
        self.optimizer = torch.optim.RMSprop(self.network.parameters(), lr=learning_rate)
        self.criterion = SubpixelCrossEntropy()

    def step(self, left, right, truth):
        """pds_trainer.py:35-46 + the optimizer step of trainer.py's loop; returns the detached loss."""
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.criterion(self.model(left, right), truth)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def replicas_in_sync(self):
        """True when every rank holds bit-identical parameters (after the same number of synchronised steps)."""
        if self.world == 1:
            return True
        flat = torch.cat([p.detach().flatten() for p in self.network.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))
