"""Ray sharding across the GPUs of one node + the one exchange step of the path.

Rays are independent units (no cross-ray reduction anywhere in the path), but their cost varies
~1000x between a miss and a hit pixel and hits are spatially clustered, so pixels are dealt to ranks
in small tiles, round-robin: 8x8 pixel tiles when the batch carries `mask_at_box` + `meta.H/W`
(the in-box rays are the mask's pixels in row-major order), runs of 64 consecutive rays otherwise.
Tiles — not single pixels — because the coarse level of the distance query sweeps its vertex boxes
per wave of 64 neighbouring rays: a rank that owned every N-th pixel would spread each wave over an
N times larger image area and lose most of the pruning (measured: 6.4 ms instead of 1.3 ms of
coarse-level time per frame for a 1/8 shard).  Every rank renders its shard with the single-GPU
pipeline; one all_gather of fixed-size shards (RCCL over xGMI; `nccl` backend on ROCm) assembles the
frame on every rank.  The reference has no multi-GPU inference (run.py is single-process); this is new.
"""
import torch
import torch.distributed as dist

from .base_utils import dotdict

RAY_KEYS = ('ray_o', 'ray_d', 'near', 'far')
TILE = 8          # pixels per tile side
RUN = 64          # rays per run when pixel coordinates are unknown


_PLANS = {}


def _owner(P: int, world: int, batch, device) -> torch.Tensor:
    mask = None if batch is None else batch.get('mask_at_box', None)
    meta = None if batch is None else batch.get('meta', None)
    if mask is not None and meta is not None and 'H' in meta and 'W' in meta:
        H, W = int(torch.as_tensor(meta['H']).reshape(-1)[0]), int(torch.as_tensor(meta['W']).reshape(-1)[0])
        m = mask.reshape(-1).to(device)
        if m.numel() == H * W:
            pix = m.nonzero()[:, 0]
            if pix.numel() == P:
                ty, tx = (pix // W) // TILE, (pix % W) // TILE
                return (ty + tx) % world      # diagonal stripes: horizontally AND vertically adjacent tiles differ
    return (torch.arange(P, device=device) // RUN) % world


def plan(P: int, world: int, batch=None, device=None) -> dotdict:
    """who owns which ray, and the index vectors of the exchange.  A pure function of (P, world, H, W, mask content), hence
    identical on all ranks.  Cached per mask tensor OBJECT: the cache holds a strong reference to the mask, so its address
    cannot be recycled for another frame's mask, and an in-place change bumps its version."""
    device = torch.device('cpu') if device is None else torch.device(device)
    mask = None if batch is None else batch.get('mask_at_box', None)
    key = (P, world, str(device))
    hit = _PLANS.get(key)
    if hit is not None and hit[0] is mask and hit[1] == (None if mask is None else mask._version):
        return hit[2]
    owner = _owner(P, world, batch, device)
    counts = torch.bincount(owner, minlength=world)
    n_max = int(counts.max()) if P else 0
    order = torch.argsort(owner, stable=True)                 # rays grouped by owner, original order inside
    starts = torch.cumsum(counts, 0) - counts
    slot = torch.arange(P, device=device) - starts[owner[order]]
    pl = dotdict(owner=owner, counts=counts, n_max=n_max, order=order, src=owner[order] * n_max + slot,
                 idx=[order[int(starts[r]):int(starts[r]) + int(counts[r])] for r in range(world)])
    if len(_PLANS) > 16:
        _PLANS.clear()
    _PLANS[key] = (mask, None if mask is None else mask._version, pl)
    return pl


def ray_owner(P: int, world: int, batch=None) -> torch.Tensor:
    """(P,) int64: owning rank of every in-box ray."""
    return plan(P, world, batch).owner


def shard_indices(P: int, rank: int, world: int, batch=None, device=None) -> torch.Tensor:
    return plan(P, world, batch, device).idx[rank]


def shard_batch(batch, rank: int, world: int, render_chunk_size=None):
    """view of `batch` holding only this rank's rays (frame state is replicated).
    The single-GPU renderer grows batch.wbounds in place once per chunk of cfg.render_chunk_size rays (quirk 1), so a ray's
    shadow-ray box depends on the chunk it falls in.  With `render_chunk_size` the shard carries `render_chunks`: the ranges of
    ITS rays that belong to each chunk of the whole frame (rays keep their order inside a shard, so they are contiguous; empty
    ranges still grow the box), which makes the merged shards identical to the single-GPU frame for multi-chunk frames too."""
    if world == 1:
        return batch
    P = batch.ray_o.shape[1]
    idx = shard_indices(P, rank, world, batch, batch.ray_o.device)
    out = dotdict(batch)
    for k in RAY_KEYS:
        out[k] = batch[k][:, idx].contiguous()
    out.wbounds = batch.wbounds.clone()      # the renderer grows it in place per chunk (quirk 1)
    if render_chunk_size is not None and P > 0:
        import math
        actual = math.ceil(P / math.ceil(P / render_chunk_size))          # chunkify's size rule (net_utils.py:323)
        edges = torch.arange(0, P + actual, actual, device=idx.device).clamp(max=P)
        pos = torch.searchsorted(idx, edges).tolist()                     # idx is ascending
        out.render_chunks = [(pos[i], pos[i + 1]) for i in range(len(pos) - 1)]
    return out


def gather_maps(local: torch.Tensor, P: int, rank: int, world: int, group=None, force_collective=False, batch=None) -> torch.Tensor:
    """local: (1, P_local, C) or (1, P_local) maps of this rank's rays -> (1, P, C) on every rank.
    `force_collective` runs the all_gather even at world size 1 (exercises the RCCL path on a 1-GPU box)."""
    if world == 1 and not force_collective:
        return local
    squeeze = local.ndim == 2
    x = local[0] if not squeeze else local[0, :, None]
    pl = plan(P, world, batch, x.device)
    buf = x.new_zeros(pl.n_max, x.shape[-1])
    buf[:x.shape[0]] = x
    out = x.new_empty(world * pl.n_max, x.shape[-1])
    dist.all_gather_into_tensor(out, buf, group=group)
    full = x.new_empty(P, x.shape[-1])
    full[pl.order] = out[pl.src]             # rank r, slot j  ->  the j-th ray owned by r
    full = full[None]
    return full[..., 0] if squeeze else full


def render_sharded(renderer, batch, keys=('rgb_map', 'acc_map'), rank=None, world=None, group=None):
    """render this rank's rays and all_gather the requested maps (packed into one collective)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    P = batch.ray_o.shape[1]
    cfg = getattr(renderer, 'cfg', None)
    if world > 1 and cfg is not None and cfg.get('vis_ground_shading', False):
        raise ValueError('render_sharded: the ground-plane pass (cfg.vis_ground_shading) needs the whole frame\'s human layer '
                         'and is not sharded; render it on one rank')
    out = renderer.render(shard_batch(batch, rank, world, None if cfg is None else cfg.render_chunk_size))
    if world == 1:
        return dotdict({k: out[k] for k in keys})
    parts = [out[k] if out[k].ndim == 3 else out[k][..., None] for k in keys]
    widths = [p.shape[-1] for p in parts]
    packed = gather_maps(torch.cat(parts, dim=-1), P, rank, world, group, batch=batch)
    res, c = dotdict(), 0
    for k, w, p in zip(keys, widths, parts):
        v = packed[..., c:c + w]
        res[k] = v if out[k].ndim == 3 else v[..., 0]
        c += w
    return res
