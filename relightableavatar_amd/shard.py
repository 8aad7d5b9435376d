"""Ray sharding across the GPUs of one node + the one exchange step of the path.

Rays are independent units (no cross-ray reduction anywhere in the path), but their cost varies
~1000x between a miss and a hit pixel and hits are spatially clustered, so rays are dealt to ranks
round-robin (ray i -> rank i % world).  Every rank renders its shard with the single-GPU pipeline;
one all_gather of fixed-size shards (RCCL over xGMI; `nccl` backend on ROCm) assembles the frame on
every rank.  The reference has no multi-GPU inference (run.py is single-process); this is new.
"""
import torch
import torch.distributed as dist

from .base_utils import dotdict

RAY_KEYS = ('ray_o', 'ray_d', 'near', 'far')


def shard_indices(P: int, rank: int, world: int) -> torch.Tensor:
    return torch.arange(rank, P, world)


def shard_batch(batch, rank: int, world: int):
    """view of `batch` holding only this rank's rays (frame state is replicated)."""
    if world == 1:
        return batch
    P = batch.ray_o.shape[1]
    idx = shard_indices(P, rank, world).to(batch.ray_o.device)
    out = dotdict(batch)
    for k in RAY_KEYS:
        out[k] = batch[k][:, idx].contiguous()
    out.wbounds = batch.wbounds.clone()      # the renderer grows it in place per chunk (quirk 1)
    return out


def gather_maps(local: torch.Tensor, P: int, rank: int, world: int, group=None, force_collective=False) -> torch.Tensor:
    """local: (1, P_local, C) or (1, P_local) maps of this rank's rays -> (1, P, C) on every rank.
    `force_collective` runs the all_gather even at world size 1 (exercises the RCCL path on a 1-GPU box)."""
    if world == 1 and not force_collective:
        return local
    squeeze = local.ndim == 2
    x = local[0] if not squeeze else local[0, :, None]
    n_max = (P + world - 1) // world
    buf = x.new_zeros(n_max, x.shape[-1])
    buf[:x.shape[0]] = x
    out = x.new_empty(world * n_max, x.shape[-1])
    dist.all_gather_into_tensor(out, buf, group=group)
    # rank r, slot j  ->  ray j * world + r
    full = out.view(world, n_max, -1).permute(1, 0, 2).reshape(world * n_max, -1)[:P]
    full = full[None]
    return full[..., 0] if squeeze else full


def render_sharded(renderer, batch, keys=('rgb_map', 'acc_map'), rank=None, world=None, group=None):
    """render this rank's rays and all_gather the requested maps (packed into one collective)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    P = batch.ray_o.shape[1]
    out = renderer.render(shard_batch(batch, rank, world))
    if world == 1:
        return dotdict({k: out[k] for k in keys})
    parts = [out[k] if out[k].ndim == 3 else out[k][..., None] for k in keys]
    widths = [p.shape[-1] for p in parts]
    packed = gather_maps(torch.cat(parts, dim=-1), P, rank, world, group)
    res, c = dotdict(), 0
    for k, w, p in zip(keys, widths, parts):
        v = packed[..., c:c + w]
        res[k] = v if out[k].ndim == 3 else v[..., 0]
        c += w
    return res
