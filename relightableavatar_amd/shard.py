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

The PLAN (who owns which ray, the index vectors of the exchange) is a pure function of (mask content, H, W, world), hence
identical on all ranks.  It is HOST work (numpy on the loader's CPU mask, as the reference's dataset produces it,
lib/utils/data_utils.py:925-938): no device round trip, no host sync in the render loop; its index vectors travel to the GPU
with the batch.  `make_plan` is explicit — callers compute it once per frame BEFORE rendering and hand it to `shard_batch` /
`gather_maps` / `render_sharded`, so a renderer that mutates `mask_at_box` (the ground pass does, quirk :1103) cannot change
the ownership between the shard and the gather.

The ground-plane pass (cfg.vis_ground_shading, the README's relight command) is full-frame work: its pixels are dealt with the
SAME tile rule over the whole H x W frame, so a rank's in-box (human) pixels are a subset of its ground pixels, the alpha blend
of the two layers is rank-local, and the frame still needs ONE all_gather (of the blended full-frame maps).
"""
import math
import zlib

import numpy as np
import torch
import torch.distributed as dist

from .base_utils import dotdict

RAY_KEYS = ('ray_o', 'ray_d', 'near', 'far')
TILE = 8          # pixels per tile side
RUN = 64          # rays per run when pixel coordinates are unknown


_PLANS = {}


def _deal(owner: np.ndarray, world: int):
    """owner (n,) -> per-rank ascending index lists, counts, and the exchange's inverse (order, src)."""
    idx = [np.flatnonzero(owner == r) for r in range(world)]          # world passes over n: cheaper than a stable argsort
    counts = [int(i.size) for i in idx]
    n_max = max(counts) if counts else 0
    order = np.concatenate(idx) if idx else np.zeros(0, np.int64)      # rays grouped by owner, original order inside
    src = np.concatenate([r * n_max + np.arange(c, dtype=np.int64) for r, c in enumerate(counts)]) if idx else np.zeros(0, np.int64)
    return idx, counts, n_max, order, src


def _chunk_ranges(idx: np.ndarray, total: int, chunk_size: int):
    """ranges of the ascending index list `idx` that fall into each chunk of the WHOLE list of `total` items (chunkify's size
    rule, net_utils.py:323): the renderer grows batch.wbounds in place once per chunk (quirk 1), so a shard must walk the
    frame's chunks — empty ranges still grow the box."""
    if total <= 0:
        return []
    actual = math.ceil(total / math.ceil(total / chunk_size))
    edges = np.minimum(np.arange(0, total + actual, actual), total)
    pos = np.searchsorted(idx, edges).tolist()
    return [(pos[i], pos[i + 1]) for i in range(len(pos) - 1)]


def _frame_of(batch):
    mask = None if batch is None else batch.get('mask_at_box', None)
    meta = None if batch is None else batch.get('meta', None)
    if mask is None or meta is None or 'H' not in meta or 'W' not in meta:
        return None, 0, 0
    H, W = int(torch.as_tensor(meta['H']).reshape(-1)[0]), int(torch.as_tensor(meta['W']).reshape(-1)[0])
    return mask, H, W


def make_plan(P: int, world: int, batch=None, device=None, mask=None, ground: bool = False, render_chunk_size=None, use_cache: bool = True) -> dotdict:
    """The frame's ownership + exchange index vectors.  `mask`: the frame's mask_at_box on the HOST (torch / numpy) — what a
    loader hands over; when omitted it is taken from `batch` (a device-resident mask costs one D2H copy = one sync; pass the
    host copy to avoid it).  `ground`: also plan the full-frame ground pass.  Cached on the mask CONTENT (crc32), so neither a
    recycled tensor address nor an in-place edit can alias two frames; `use_cache=False` = a new frame every call (bench.py)."""
    device = torch.device('cpu') if device is None else torch.device(device)
    bmask, H, W = _frame_of(batch)
    if mask is None:
        mask = bmask
    m = None
    if mask is not None and H * W > 0:
        m = mask.detach().reshape(-1).cpu().numpy() if torch.is_tensor(mask) else np.asarray(mask).reshape(-1)
        m = m.astype(bool, copy=False)
        if m.size != H * W:
            m = None
    key = None
    if use_cache:
        key = (P, world, H, W, str(device), bool(ground), render_chunk_size, None if m is None else zlib.crc32(np.packbits(m).tobytes()))
        hit = _PLANS.get(key)
        if hit is not None:
            return hit
    pix = None
    if m is not None:
        pix = np.flatnonzero(m)
        if pix.size != P:
            pix = None
    if pix is not None:
        owner = ((pix // W) // TILE + (pix % W) // TILE) % world      # diagonal stripes: horizontally AND vertically adjacent tiles differ
    else:
        owner = (np.arange(P) // RUN) % world
    idx, counts, n_max, order, src = _deal(owner, world)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(device, non_blocking=True)
    pl = dotdict(P=P, world=world, H=H, W=W, owner=torch.from_numpy(owner.astype(np.int64)), counts=counts, n_max=n_max,
                 idx=[up(i) for i in idx], order=up(order), src=up(src), idx_host=idx)
    if render_chunk_size is not None:
        pl.render_chunks = [_chunk_ranges(i, P, render_chunk_size) for i in idx]
    if ground:
        if pix is None:
            raise ValueError('shard.make_plan: the ground-plane pass needs mask_at_box and meta.H / meta.W (full-frame pixels)')
        F = H * W
        fp = np.arange(F)
        g_owner = ((fp // W) // TILE + (fp % W) // TILE) % world
        g_idx, g_counts, g_n_max, g_order, g_src = _deal(g_owner, world)
        g = dotdict(F=F, counts=g_counts, n_max=g_n_max, idx=[up(i) for i in g_idx], order=up(g_order), src=up(g_src), idx_host=g_idx)
        # where this rank's human rays sit in its ground pixel list (both ascending, human pixels are a subset)
        g.inds = [up(np.searchsorted(g_idx[r], pix[idx[r]])) for r in range(world)]
        if render_chunk_size is not None:
            g.chunks = [_chunk_ranges(i, F, render_chunk_size) for i in g_idx]
        pl.ground = g
    if key is not None:
        if len(_PLANS) > 16:
            _PLANS.clear()
        _PLANS[key] = pl
    return pl


def plan(P: int, world: int, batch=None, device=None) -> dotdict:
    return make_plan(P, world, batch, device)


def ray_owner(P: int, world: int, batch=None) -> torch.Tensor:
    """(P,) int64: owning rank of every in-box ray."""
    return make_plan(P, world, batch).owner


def shard_indices(P: int, rank: int, world: int, batch=None, device=None) -> torch.Tensor:
    return make_plan(P, world, batch, device).idx[rank]


def shard_batch(batch, rank: int, world: int, render_chunk_size=None, plan=None, ground: bool = False):
    """view of `batch` holding only this rank's rays (frame state is replicated).
    The single-GPU renderer grows batch.wbounds in place once per chunk of cfg.render_chunk_size rays (quirk 1), so a ray's
    shadow-ray box depends on the chunk it falls in.  With `render_chunk_size` the shard carries `render_chunks`: the ranges of
    ITS rays that belong to each chunk of the whole frame (rays keep their order inside a shard, so they are contiguous; empty
    ranges still grow the box), which makes the merged shards identical to the single-GPU frame for multi-chunk frames too.
    `ground`: the shard also carries this rank's full-frame pixels of the ground pass (`ground_pix`, `ground_chunks`) and the
    positions of its human rays among them (`ground_inds`)."""
    if world == 1:
        return batch
    P = batch.ray_o.shape[1]
    dev = batch.ray_o.device
    if plan is None:
        plan = make_plan(P, world, batch, dev, ground=ground, render_chunk_size=render_chunk_size)
    idx = plan.idx[rank]
    out = dotdict(batch)
    for k in RAY_KEYS:
        out[k] = batch[k][:, idx].contiguous()
    out.wbounds = batch.wbounds.clone()      # the renderer grows it in place per chunk (quirk 1)
    if batch.get('wbounds_host', None) is not None and batch.get('wbounds_host_version', None) == batch.wbounds._version:
        out.wbounds_host, out.wbounds_host_version = batch.wbounds_host.clone(), out.wbounds._version      # its host mirror follows (no read-back)
    else:
        out.pop('wbounds_host', None)
    if batch.get('mask_at_box', None) is not None:
        out.mask_at_box = batch.mask_at_box.clone()     # the ground pass overwrites it in place (:1103): not on the caller's frame
    if render_chunk_size is not None and P > 0:
        out.render_chunks = plan.render_chunks[rank] if 'render_chunks' in plan else _chunk_ranges(plan.idx_host[rank], P, render_chunk_size)
    if ground:
        g = plan.ground
        out.ground_pix, out.ground_inds = g.idx[rank], g.inds[rank]
        out.ground_chunks = g.chunks[rank] if 'chunks' in g else _chunk_ranges(g.idx_host[rank], g.F, render_chunk_size or g.F)
    return out


def _exchange(x: torch.Tensor, n_max: int, order, src, total: int, world: int, group=None) -> torch.Tensor:
    buf = x.new_zeros(n_max, x.shape[-1])
    buf[:x.shape[0]] = x
    out = x.new_empty(world * n_max, x.shape[-1])
    dist.all_gather_into_tensor(out, buf, group=group)
    full = x.new_empty(total, x.shape[-1])
    full[order] = out[src]                   # rank r, slot j  ->  the j-th item owned by r
    return full


def gather_maps(local: torch.Tensor, P: int, rank: int, world: int, group=None, force_collective=False, batch=None, plan=None,
                ground: bool = False) -> torch.Tensor:
    """local: (1, P_local, C) or (1, P_local) maps of this rank's rays -> (1, P, C) on every rank (`ground`: full-frame maps of
    this rank's ground pixels -> (1, H*W, C)).  `force_collective` runs the all_gather even at world size 1 (exercises the RCCL
    path on a 1-GPU box)."""
    if world == 1 and not force_collective:
        return local
    squeeze = local.ndim == 2
    x = local[0] if not squeeze else local[0, :, None]
    if plan is None:
        plan = make_plan(P, world, batch, x.device, ground=ground)
    pl = plan.ground if ground else plan
    full = _exchange(x, pl.n_max, pl.order, pl.src, pl.F if ground else P, world, group)[None]
    return full[..., 0] if squeeze else full


def render_sharded(renderer, batch, keys=('rgb_map', 'acc_map'), rank=None, world=None, group=None, plan=None):
    """render this rank's rays and all_gather the requested maps (packed into one collective).  With cfg.vis_ground_shading the
    rank also renders its full-frame tiles of the ground pass and the gathered maps are the blended full-frame ones."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    P = batch.ray_o.shape[1]
    cfg = getattr(renderer, 'cfg', None)
    ground = bool(cfg is not None and cfg.get('vis_ground_shading', False))
    chunk = None if cfg is None else cfg.render_chunk_size
    if world > 1 and plan is None:
        plan = make_plan(P, world, batch, batch.ray_o.device, ground=ground, render_chunk_size=chunk)
    out = renderer.render(shard_batch(batch, rank, world, chunk, plan, ground))
    if world == 1:
        return dotdict({k: out[k] for k in keys})
    parts = [out[k] if out[k].ndim == 3 else out[k][..., None] for k in keys]
    widths = [p.shape[-1] for p in parts]
    packed = gather_maps(torch.cat(parts, dim=-1), P, rank, world, group, plan=plan, ground=ground)
    res, c = dotdict(), 0
    for k, w, p in zip(keys, widths, parts):
        v = packed[..., c:c + w]
        res[k] = v if out[k].ndim == 3 else v[..., 0]
        c += w
    return res
