"""Run under `python -m torch.distributed.run --nproc-per-node 1 ...` on a GPU box: RCCL init + the frame all_gather
at world size 1 (the only multi-process GPU check a 1-GPU box allows)."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relightableavatar_amd import shard
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
x = torch.arange(1, 1 + 1001 * 4, device='cuda', dtype=torch.float32).view(1, 1001, 4)
y = shard.gather_maps(x, 1001, 0, 1, force_collective=True)
assert torch.equal(x, y), 'all_gather round trip differs'
z = shard.gather_maps(x[..., 0], 1001, 0, 1, force_collective=True)
assert torch.equal(x[..., 0], z)
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print('NCCL_WORLD1_OK')
