"""Ray sharding across the GPUs of one node + the one exchange step of the path.

Rays are independent units (no cross-ray reduction anywhere in the path), but their cost varies
~1000x between a miss and a hit pixel and hits are spatially clustered, so pixels are dealt to ranks
in small tiles, round-robin: 8x8 pixel tiles when the batch carries `mask_at_box` + `meta.H/W`
(the in-box rays are the mask's pixels in row-major order; the tiles that hold in-box pixels are dealt in raster order), runs of 64
consecutive rays otherwise.
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
    """owner (n,) -> per-rank ascending index lists, counts, and the exchange's inverse (order, src).
    The lists are views of `order` (rays grouped by owner, original order inside): one stable sort of the one-byte owners (numpy
    sorts bytes by counting) instead of `world` passes over the rays."""
    order = np.argsort(owner.astype(np.uint8 if world <= 256 else np.int64, copy=False), kind='stable').astype(np.int64, copy=False)
    counts = np.bincount(owner, minlength=world)[:world].tolist()
    offs = np.concatenate([[0], np.cumsum(counts)]).tolist()
    idx = [order[offs[r]:offs[r + 1]] for r in range(world)]
    n_max = max(counts) if counts else 0
    src = np.arange(order.size, dtype=np.int64) + np.repeat(np.arange(world, dtype=np.int64) * n_max - np.asarray(offs[:-1], np.int64), counts)
    return idx, counts, n_max, order, src, offs


def _upload(arrays, device):
    """int64 index vectors -> device tensors with ONE transfer.  Through a pinned staging block and non-blocking: a pageable
    `tensor.to(device)` waits for everything queued on the stream before it (measured: 1.5 ms each with a frame in flight,
    10 per plan — the host could not run ahead of the GPU any more)."""
    sizes = [int(a.size) for a in arrays]
    if device.type != 'cuda':
        return [torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)) for a in arrays]
    stage = torch.empty(max(sum(sizes), 1), dtype=torch.int64, pin_memory=True)        # the caching host allocator recycles the block
    view, o = stage.numpy(), 0
    for a, n in zip(arrays, sizes):
        view[o:o + n] = a
        o += n
    dev = stage.to(device, non_blocking=True)
    return list(torch.split(dev[:o], sizes)) if o else [dev[:0] for _ in sizes]


def _mark_ready(pl, device):
    """the plan's index vectors were uploaded (pinned, non-blocking) on the CURRENT stream; a cached plan may be consumed later from any
    other stream (frames in flight: every replica has its own): remember the event that ends the upload and the block it wrote"""
    if torch.device(device).type == 'cuda':
        pl.ready = torch.cuda.Event()
        pl.ready.record(torch.cuda.current_stream(device))
    return pl


def _use(pl):
    """called by every consumer of a plan's device tensors: the current stream waits for the plan's upload, and the caching allocator
    learns that this stream reads the block (so `_PLANS.clear()` cannot recycle it under a running gather)"""
    ev = pl.get('ready', None)
    if ev is not None and pl.order.is_cuda:
        cur = torch.cuda.current_stream(pl.order.device)
        cur.wait_event(ev)
        pl.order.record_stream(cur)          # order / src / idx / inds are views of ONE uploaded block


_FRAMES = {}


def _frame_deal(H: int, W: int, world: int, device):
    """What the tile rule fixes for an H x W frame whatever its mask: the owner of every pixel (one byte each) and the whole deal of the
    full-frame ground pass (per-rank pixel lists, exchange index vectors — on the device —, the position of every pixel in its owner's
    list).  A function of the frame size only: computed once and kept (a 1024 x 1024 frame: 25 ms of numpy per call otherwise)."""
    key = (H, W, world, str(device))
    f = _FRAMES.get(key)
    if f is None:
        if len(_FRAMES) > 8:
            _FRAMES.clear()
        y, x = np.divmod(np.arange(H * W), W)
        owner = ((y // TILE + x // TILE) % world).astype(np.uint8 if world <= 256 else np.int64)
        tiles_x = (W + TILE - 1) // TILE
        tile = ((y // TILE) * tiles_x + x // TILE).astype(np.int32)           # the 8 x 8 tile of every pixel, raster order of tiles
        f = _FRAMES[key] = dotdict(owner=owner, tile=tile, n_tiles=tiles_x * ((H + TILE - 1) // TILE), ground=None)
    return f


def _frame_ground(f, H: int, W: int, world: int, device):
    if f.ground is None:
        idx, counts, n_max, order, src, offs = _deal(f.owner, world)
        pos = np.empty(H * W, np.int64)
        pos[order] = np.arange(H * W, dtype=np.int64) - np.repeat(np.asarray(offs[:-1], np.int64), counts)
        order_d, src_d = _upload([order, src], device)
        if torch.device(device).type == 'cuda':
            torch.cuda.current_stream(device).synchronize()     # kept and used from any stream later (frames in flight): make the one upload visible to all, once
        f.ground = dotdict(idx=idx, counts=counts, n_max=n_max, offs=offs, pos=pos, order=order_d, src=src_d, chunks={})
    return f.ground


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


_NUMPY_PLAN = False      # tests: True forces the numpy restatement below (the two must agree element for element)


def _chunk_edges(total: int, chunk_size: int):
    """chunk boundaries of a list of `total` items (chunkify's size rule, net_utils.py:323)"""
    actual = math.ceil(total / math.ceil(total / chunk_size))
    return np.minimum(np.arange(0, total + actual, actual), total).astype(np.int64)


def _make_plan_native(m: np.ndarray, P: int, world: int, H: int, W: int, device, ground: bool, render_chunk_size):
    """the per-frame part of the plan in ONE pass over the mask (C ABI: ra_shard_plan, csrc/ra_shard.cpp), its index vectors written
    straight into the pinned staging block that is uploaded with one non-blocking copy.  Returns None when the mask does not hold P
    pixels (the caller then deals runs of rays)."""
    import ctypes as C
    from . import _lib
    mb = np.ascontiguousarray(m.view(np.uint8) if m.dtype == np.bool_ else (m != 0).astype(np.uint8))
    if int(np.count_nonzero(mb)) != P:
        return None
    cuda = torch.device(device).type == 'cuda'
    fg = _frame_ground(_frame_deal(H, W, world, device), H, W, world, device) if ground else None
    n_vec = 3 if ground else 2
    stage = torch.empty(max(n_vec * P, 1), dtype=torch.int64, pin_memory=cuda)
    sv = stage.numpy()
    owner = np.empty(max(P, 1), np.uint8)
    counts = np.zeros(world, np.int64)
    edges = _chunk_edges(P, render_chunk_size) if (render_chunk_size is not None and P > 0) else np.zeros(0, np.int64)
    cpos = np.zeros(max(world * edges.size, 1), np.int64)
    n_max = C.c_longlong(0)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    base = stage.data_ptr()
    _lib.check(_lib.lib().ra_shard_plan(ptr(mb), H, W, world, int(bool(ground)), P, ptr(fg.pos) if ground else None,
                                        ptr(edges) if edges.size else None, int(edges.size), ptr(owner), C.c_void_p(base), C.c_void_p(base + 8 * P),
                                        C.c_void_p(base + 16 * P) if ground else None, ptr(counts), ptr(cpos) if edges.size else None, C.byref(n_max)),
               'ra_shard_plan')
    counts_l = counts.tolist()
    offs = np.concatenate([[0], np.cumsum(counts)]).tolist()
    dev = stage.to(device, non_blocking=True) if cuda else stage
    per_rank = lambda t, o: [t[o[r]:o[r + 1]] for r in range(world)]
    order_d, src_d = dev[:P], dev[P:2 * P]
    pl = dotdict(P=P, world=world, H=H, W=W, counts=counts_l, n_max=int(n_max.value), idx=per_rank(order_d, offs), order=order_d, src=src_d,
                 idx_host=per_rank(sv[:P], offs), owner_host=owner[:P], _stage=stage)
    pl.owner = torch.from_numpy(owner[:P].astype(np.int64))
    if edges.size:
        cp = cpos[:world * edges.size].reshape(world, edges.size).tolist()
        pl.render_chunks = [[(cp[r][i], cp[r][i + 1]) for i in range(edges.size - 1)] for r in range(world)]
    elif render_chunk_size is not None:
        pl.render_chunks = [[] for _ in range(world)]
    if ground:
        F = H * W
        inds_d = dev[2 * P:3 * P]
        g = dotdict(F=F, counts=fg.counts, n_max=fg.n_max, idx=per_rank(fg.order, fg.offs), order=fg.order, src=fg.src, idx_host=fg.idx,
                    inds=per_rank(inds_d, offs))
        if render_chunk_size is not None:
            if render_chunk_size not in fg.chunks:
                fg.chunks[render_chunk_size] = [_chunk_ranges(i, F, render_chunk_size) for i in fg.idx]
            g.chunks = fg.chunks[render_chunk_size]
        pl.ground = g
    return _mark_ready(pl, device)


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
    if m is not None and world <= 256 and not _NUMPY_PLAN:
        pl = _make_plan_native(m, P, world, H, W, device, ground, render_chunk_size)
        if pl is not None:
            if key is not None:
                if len(_PLANS) > 16:
                    _PLANS.clear()
                _PLANS[key] = pl
            return pl
    pix = None
    if m is not None:
        pix = np.flatnonzero(m)
        if pix.size != P:
            pix = None
    fd = _frame_deal(H, W, world, device) if pix is not None else None
    if pix is not None and not ground:
        # the tiles that hold in-box pixels, dealt round robin in raster order: every rank gets the same number of tiles (+-1) whatever the
        # shape of the mask, and its cost — hit pixels — spreads evenly (512 x 512 frame, 8 ranks: hits per rank max / mean 1.03 against
        # 1.08 for fixed diagonal stripes, min / mean 0.97 against 0.87; the slowest rank sets the frame time)
        tid = fd.tile[pix]
        present = np.bincount(tid, minlength=fd.n_tiles) > 0
        tile_rank = ((np.cumsum(present) - 1) % world).astype(fd.owner.dtype)
        owner = tile_rank[tid]
    elif pix is not None:
        # with the ground pass: fixed diagonal stripes over the WHOLE frame, so that a rank's in-box pixels are a subset of its ground pixels
        owner = fd.owner[pix]
    else:
        owner = ((np.arange(P) // RUN) % world).astype(np.uint8 if world <= 256 else np.int64)
    idx, counts, n_max, order, src, offs = _deal(owner, world)
    host = [order, src]
    if ground:
        if pix is None:
            raise ValueError('shard.make_plan: the ground-plane pass needs mask_at_box and meta.H / meta.W (full-frame pixels)')
        fg = _frame_ground(fd, H, W, world, device)
        # where a rank's human rays sit in its ground pixel list (human pixels are a subset of the rank's ground pixels)
        host += [fg.pos[pix[i]] for i in idx]
    dev = _upload(host, device)                      # every per-frame index vector of the plan in one transfer
    per_rank = lambda t, o: [t[o[r]:o[r + 1]] for r in range(world)]
    pl = dotdict(P=P, world=world, H=H, W=W, owner=torch.from_numpy(owner.astype(np.int64)), counts=counts, n_max=n_max,
                 idx=per_rank(dev[0], offs), order=dev[0], src=dev[1], idx_host=idx)
    if render_chunk_size is not None:
        pl.render_chunks = [_chunk_ranges(i, P, render_chunk_size) for i in idx]
    if ground:
        F = H * W
        g = dotdict(F=F, counts=fg.counts, n_max=fg.n_max, idx=per_rank(fg.order, fg.offs), order=fg.order, src=fg.src, idx_host=fg.idx,
                    inds=dev[2:2 + world])
        if render_chunk_size is not None:
            if render_chunk_size not in fg.chunks:
                fg.chunks[render_chunk_size] = [_chunk_ranges(i, F, render_chunk_size) for i in fg.idx]
            g.chunks = fg.chunks[render_chunk_size]
        pl.ground = g
    _mark_ready(pl, device)
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
    _use(plan)
    idx = plan.idx[rank]
    out = dotdict(batch)
    if batch.ray_o.is_cuda and all(batch[k].dtype == torch.float32 and batch[k].is_contiguous() for k in RAY_KEYS) and idx.dtype == torch.int64:
        # on the device: the library's own gather, one launch for the four arrays (ra_gather_rays)
        from . import _lib
        import ctypes as C
        n = int(idx.shape[0])
        for k in RAY_KEYS:
            out[k] = batch[k].new_empty((1, n) + tuple(batch[k].shape[2:]))
        pt = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().ra_gather_rays(dev.index or 0, pt(idx), n, pt(batch.ray_o), pt(batch.ray_d), pt(batch.near), pt(batch.far),
                                             pt(out.ray_o), pt(out.ray_d), pt(out.near), pt(out.far),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'ra_gather_rays')
    else:
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


def _unshuffle(out: torch.Tensor, order: torch.Tensor, src: torch.Tensor, total: int) -> torch.Tensor:
    """full[order] = out[src] (rank r, slot j -> the j-th item owned by r): on the device the library's own one-launch scatter"""
    full = out.new_empty(total, out.shape[-1])
    if out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and order.dtype == torch.int64 and src.dtype == torch.int64:
        from . import _lib
        import ctypes as C
        pt = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().ra_scatter_rows(out.device.index or 0, pt(out), pt(src), pt(order), int(order.shape[0]), int(out.shape[-1]), pt(full),
                                              C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)), 'ra_scatter_rows')
    else:
        full[order] = out[src]
    return full


def _host_staged(x: torch.Tensor, group=None) -> bool:
    """device tensors over a process group that cannot move them (gloo: N rank processes sharing ONE GPU — RCCL refuses two ranks on a
    device —, the multi-process test of the rank program): the collective runs on pinned host copies"""
    return x.is_cuda and dist.get_backend(group) == 'gloo'


def _exchange(x: torch.Tensor, n_max: int, order, src, total: int, world: int, group=None) -> torch.Tensor:
    buf = x.new_zeros(n_max, x.shape[-1])
    buf[:x.shape[0]] = x
    if _host_staged(x, group):
        st = torch.cuda.current_stream(x.device)
        hb = torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
        hb.copy_(buf, non_blocking=True)
        st.synchronize()                         # the frame's last kernel: a host-staged gather needs the data on the host
        ho = torch.empty(world * n_max, x.shape[-1], dtype=x.dtype, pin_memory=True)
        dist.all_gather_into_tensor(ho, hb, group=group)
        out = ho.to(x.device, non_blocking=True)
        out.record_stream(st)
    else:
        out = x.new_empty(world * n_max, x.shape[-1])
        dist.all_gather_into_tensor(out, buf, group=group)
    return _unshuffle(out, order, src, total)          # rank r, slot j  ->  the j-th item owned by r


def gather_maps_async(local: torch.Tensor, P: int, rank: int, world: int, plan, group=None, ground: bool = False):
    """the frame all_gather issued WITHOUT waiting for it: returns finish() -> (1, P, C) (or (1, H*W, C) with `ground`).  Several frames'
    collectives may be in flight in one process group — what frames in flight do on the GPU, where every frame's gather is queued on its
    replica's stream; the process group runs them in submission order, which must be the same on every rank.  (On the GPU the plain
    `gather_maps` is already asynchronous for the host: the collective is stream-ordered.)"""
    if world == 1:
        return lambda: local
    squeeze = local.ndim == 2
    x = local[0] if not squeeze else local[0, :, None]
    _use(plan)
    pl = plan.ground if ground else plan
    total = pl.F if ground else P
    issued_on = torch.cuda.current_stream(x.device) if x.is_cuda else None
    buf = x.new_zeros(pl.n_max, x.shape[-1])
    buf[:x.shape[0]] = x
    out = x.new_empty(world * pl.n_max, x.shape[-1])
    work = dist.all_gather_into_tensor(out, buf, group=group, async_op=True)

    def finish():
        work.wait()          # on a device, this orders only the CURRENT stream behind the collective
        if issued_on is not None:
            cur = torch.cuda.current_stream(x.device)
            if cur != issued_on:      # buffers of the issuing stream, read on this one: the caching allocator must not recycle them under it
                out.record_stream(cur)
                buf.record_stream(cur)
        full = _unshuffle(out, pl.order, pl.src, total)[None]
        return full[..., 0] if squeeze else full
    return finish


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
    _use(plan)
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
