"""A one-rank RCCL process group on cuda:0: the collectives pre-flight of distributed.py against the real library (all-reduce,
all-gather forms at the shard sizes of configs[2] / configs[3], ncclCommCount) -- as far as a 1-GPU box can take RCCL."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from practicaldeepstereo_nips2018_amd.distributed import preflight_collectives
info = preflight_collectives(device=torch.device('cuda', 0))
print('PREFLIGHT', {k: info[k] for k in info if k != 'knobs'})
dist.destroy_process_group()
