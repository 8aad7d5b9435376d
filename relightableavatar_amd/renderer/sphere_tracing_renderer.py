"""Sphere-tracing renderer, host-side mirror of lib/networks/renderer/sphere_tracing_renderer.py
(Renderer.render :1066-1115, get_pixel_value :981-1039, render_human :551-784, and with
cfg.vis_ground_shading the ground pass get_ground_value :1041-1064 / render_ground :463-548 / blend_output_
:434-451; eval path).  One ra_render_sphere_chunk (ra_render_ground_chunk) call per chunk of cfg.render_chunk_size rays."""
import torch
from torch import nn

from .. import config
from ..base_utils import dotdict, lazydict
from .chunking import chunks


class Renderer(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.cfg = config.active_cfg()
        # reference quirk: the ground pass scatters the human layer through batch_aware_indexing = topk(sorted=False)
        # (net_utils.py:381-389), whose index order is implementation-defined; the build uses the ascending (row-major)
        # order the mask means.  Tests set this to reproduce the order of the platform a golden frame was made on.
        self.ground_inds = None

    def _grow_bounds(self, batch):
        """batch.wbounds -= / += cfg.env_lvis.bbox_margin in place (once per render chunk, :1020-1022 / :1054-1056) and the grown box
        as six host floats for the chunk's launch.  A batch whose loader kept the host copy it made the box from
        (`wbounds_host`, e.g. synthetic.to_device) is grown on both sides and costs no device round trip; otherwise the box is
        read back (one stream sync per chunk, what the reference's own .item() calls cost)."""
        m = self.cfg.env_lvis.bbox_margin
        wb = batch.wbounds
        host = batch.get('wbounds_host', None)
        if host is not None and wb.is_cuda and batch.get('wbounds_host_version', None) != wb._version:
            host = None                      # somebody else wrote to the device tensor since the mirror was made: read it back
        if wb.is_cuda and wb.dtype == torch.float32 and wb.is_contiguous() and wb.numel() == 6:
            self.net.engine().grow_bounds(wb, m)          # one launch (the slice-and-add pair of torch: four)
        else:
            wb[:, 0] -= m
            wb[:, 1] += m
        if not wb.is_cuda:
            return wb[0].reshape(-1).tolist()
        if host is None:
            host = wb.detach().cpu()         # one stream sync; the mirror is valid from here on
        else:
            host[:, 0] -= m
            host[:, 1] += m
        batch['wbounds_host'], batch['wbounds_host_version'] = host, wb._version
        return host[0].reshape(-1).tolist()

    def _envmap(self, batch):
        cfg = self.cfg
        if not self.net.training and cfg.replace_light:
            return batch.novel_lights[cfg.replace_light]           # :1068-1069
        if hasattr(self.net, 'global_env_map'):
            return dotdict(probe=self.net.global_env_map[None])   # :1070-1071
        return None

    @torch.no_grad()
    def render(self, batch):
        cfg = self.cfg
        ground = bool(cfg.vis_ground_shading) and not self.net.training
        if ground and not (cfg.relighting and hasattr(self.net, 'global_env_map')):
            raise ValueError('vis_ground_shading needs the relighting renderer (light set + probe)')
        # render_human's early return (:702-705): with none of the rendering / shading / specular maps wanted there is no shading at all —
        # no light visibility, no rgb_map — only the geometry and material maps (the reference's albedo / normal visualisations)
        early = not (cfg.get('vis_rendering_map', True) or cfg.get('vis_shading_map', False) or cfg.get('vis_specular_map', False)) \
            and not self.net.training
        if early and (ground or cfg.vis_novel_light):
            raise NotImplementedError('vis_rendering_map / vis_shading_map / vis_specular_map all off leaves no rgb_map for the ground pass / '
                                      'the novel-light re-shade (the reference fails there too)')
        only_vis = bool(cfg.get('only_visibility', False)) and bool(cfg.relighting)
        eng = self.net.set_frame(batch)
        eng.begin_render()
        dev = eng.device
        f = lambda t: t[0].to(dev, torch.float32).contiguous()
        ray_o, ray_d, near, far = f(batch.ray_o), f(batch.ray_d), f(batch.near), f(batch.far)
        P = ray_o.shape[0]
        envmap = self._envmap(batch)
        probe = None
        if envmap is not None:
            probe = envmap.probe
            probe = (probe[0] if probe.ndim == 4 else probe).to(dev, torch.float32).contiguous()
        relit = bool(cfg.relighting) and not early
        params = eng.sphere_params()
        params.relighting = int(relit)
        if ground:
            params.premultiply = 0          # blend_output_ replaces alpha_output_ (:1108-1113)
        names = ([] if early else ['rgb']) + ['acc', 'depth', 'surf', 'norm', 'cpts', 'bpts', 'resd', 'ray_o']
        if eng.relight:
            names += ['albedo', 'roughness']
        if relit:
            names += ['shade']
            if cfg.vis_specular_map:
                names += ['spec']
            if cfg.vis_novel_light:
                names += ['lvis', 'ldot']
        raw_c = 17 if eng.relight else 16
        want_raw = bool(cfg.get('ret_raw', True))          # render_human's per-hit leftovers raw / volume_albedo / volume_roughness (:616-650)
        if want_raw:
            names += ['raw'] + (['volume_albedo', 'volume_roughness'] if eng.relight else [])
        width = dict(acc=0, depth=0, roughness=0, volume_roughness=0, lvis=cfg.env_h * cfg.env_w, ldot=cfg.env_h * cfg.env_w,
                     raw=cfg.n_samples * raw_c)
        full = dotdict()
        for k in names:
            w = width.get(k, 3)
            full[k] = torch.empty((P, w) if w else (P,), device=dev)      # every ray of every map is written by the chunks (zeros for misses)
        # The reference's chunks bound ITS memory; rays are independent and only the shadow rays' clip depends on the box.  Consecutive chunks
        # are rendered by ONE launch sequence of up to cfg.sphere_chunk_rays rays (0: chunk exactly as the reference), the shadow rays of
        # every ray clipped against the box its own chunk had reached (the in-place growth, once per chunk incl. empty ones, is data
        # independent): identical pixels, and a frame of several chunks runs one 16-iteration surface loop instead of one per chunk.
        limit = int(cfg.get('sphere_chunk_rays', 0))
        group = []                                  # (a, b, box) of the chunks waiting for a launch

        def flush():
            if group:
                a0, b1 = group[0][0], group[-1][1]
                eng.render_sphere_chunk(ray_o[a0:b1], ray_d[a0:b1], near[a0:b1], far[a0:b1], group[0][2], probe, params,
                                        {k: v[a0:b1] for k, v in full.items()},
                                        boxes=[g[2] for g in group], box_start=[g[0] - a0 for g in group] + [b1 - a0])
                group.clear()
        for a, b in (batch.get('render_chunks', None) or chunks(P, cfg.render_chunk_size)):      # render_chunks: a shard's view of the frame's chunks (shard.py)
            # quirk (sphere_tracing_renderer.py:1020-1022): the box grows IN PLACE on the batch every chunk
            bbox6 = self._grow_bounds(batch)
            if b <= a:
                continue
            if group and (a != group[-1][1] or len(group) == 32 or b - group[0][0] > limit):
                flush()
            group.append((a, b, bbox6))
        flush()
        ret = lazydict()
        if want_raw:
            # per-hit arrays in ascending ray order (the reference's order is topk(sorted=False)'s, implementation-defined).  Their
            # shape needs the hit count on the host: evaluated only when read (the trainers' losses, relight_trainer.py:78-79)
            hits = lambda: (full.acc > 0).nonzero()[:, 0]
            ret.lazy('raw', lambda: full.raw[hits()].reshape(1, -1, raw_c), deps=(full.raw, full.acc))                     # B, P_hit * S, C
            if eng.relight:
                ret.lazy('volume_albedo', lambda: full.volume_albedo[hits()][None], deps=(full.volume_albedo, full.acc))              # B, P_hit, 3
                ret.lazy('volume_roughness', lambda: full.volume_roughness[hits()][None, :, None], deps=(full.volume_roughness, full.acc))   # B, P_hit, 1
        ret.acc_map = full.acc[None]
        ret.ray_o = full.ray_o[None]
        ret.surf_map, ret.depth_map = full.surf[None], full.depth[None]
        ret.cpts_map, ret.bpts_map, ret.resd_map, ret.norm_map = full.cpts[None], full.bpts[None], full.resd[None], full.norm[None]
        if eng.relight:
            ret.albedo_map, ret.roughness_map = full.albedo[None], full.roughness[None]
        if not early:
            ret.rgb_map = full.rgb[None]
        if relit:
            # only_visibility: a one-channel light (:723, :749-751); vis_lvis_map / vis_ldot_map expand to three (:756-757)
            one_ch = only_vis and not (cfg.get('vis_lvis_map', False) or cfg.get('vis_ldot_map', False))
            # (with the ground pass the cut comes after the blend: both layers' shade maps have one channel in the reference, :516-519, and the
            # kernels write it three times)
            ret.shade_map = full.shade[None, :, :1] if (one_ch and not ground) else full.shade[None]
            if 'spec' in full:
                ret.spec_map = full.spec[None]
            if 'lvis' in full:
                ret.lvis_map, ret.ldot_map = full.lvis[None], full.ldot[None]
                if only_vis:
                    # the debugging option replaces the cosines the novel-light re-shade caches too (:720-722, :758-759): 1 on the hit rays —
                    # times acc where the maps are premultiplied (alpha_output_, no ground pass), 0 elsewhere (multi_scatter_zeros)
                    one = ret.acc_map if not ground else (ret.acc_map > 0).to(ret.ldot_map.dtype)
                    ret.ldot_map = one[..., None].expand_as(ret.ldot_map).contiguous()
        ret.envmap = envmap
        if ground:
            grd = self._ground(batch, ret, eng, probe)
            if cfg.vis_novel_light:
                ret.ground = grd             # the novel-light renderer re-shades both layers per probe, then blends (:1106-1107)
            else:
                ret = self.blend_output_(grd.acc_map, grd.inds, grd, ret, eng)
                if relit and only_vis and not (cfg.get('vis_lvis_map', False) or cfg.get('vis_ldot_map', False)):
                    ret.shade_map = ret.shade_map[..., :1]
        return ret

    BLEND_KEYS = ('rgb_map', 'rfl_map', 'surf_map', 'albedo_map', 'roughness_map', 'norm_map', 'cpts_map', 'bpts_map', 'spec_map',
                  'depth_map', 'lvis_map', 'ldot_map', 'brdf_map', 'shade_map')

    @classmethod
    def blend_output_(cls, acc, inds, grd, ret, eng):
        """blend_output_ (sphere_tracing_renderer.py:434-451): modifies and returns ret.  acc (1,F), inds (1,P) int64; maps of grd
        are (1,F,C) / (1,F), maps of ret (1,P,C) / (1,P)."""
        a, i = acc[0].contiguous(), inds[0].to(torch.int64).contiguous()
        for k in cls.BLEND_KEYS:
            if k in ret and k in grd:
                ret[k] = eng.blend_ground(grd[k][0].contiguous(), ret[k][0], i, a)[None]
            elif k in grd:
                ret[k] = eng.blend_ground(grd[k][0].contiguous(), None, i, a)[None]
        ret.acc_map = eng.blend_ground(None, ret.acc_map[0], i, a)[None]      # alpha_blend(acc, inds, zeros, acc_map) (:449)
        return ret

    def _ground(self, batch, ret, eng, probe):
        """Renderer.render :1084-1104: full-frame rays, acc = 1 - human acc on the in-box pixels, ground chunks (each growing
        batch.wbounds again, :1054-1056).  Returns the ground layer (render_ground's dotdict + ray_o, ray_d, acc_map, inds)."""
        cfg = self.cfg
        dev = eng.device
        H, W = int(batch.meta.H.item()), int(batch.meta.W.item())
        big = torch.tensor([[-1e9] * 3, [1e9] * 3])
        # the camera as host numbers: the loader's host copies when it kept them (synthetic.to_device), else one read-back each
        cam = [(batch[k + '_host'] if k + '_host' in batch else batch[k])[0].cpu().numpy() for k in ('cam_K', 'cam_R', 'cam_T')]
        rays = eng.gen_rays(H, W, cam[0], cam[1], cam[2], big, count=H * W)        # unbounded box: every pixel is a ray, no count to wait for
        g_o, g_d = rays.ray_o.contiguous(), rays.ray_d.contiguous()
        assert g_o.shape[0] == H * W
        pix = batch.get('ground_pix', None)
        if pix is not None:
            # one rank of a sharded frame (shard.py): its full-frame tiles only; its human rays are a subset of them, so the blend
            # stays local.  ground_chunks = its pixels per chunk of the WHOLE frame (every chunk grows the box, also an empty one)
            g_o, g_d = g_o[pix].contiguous(), g_d[pix].contiguous()
            inds, ranges = batch.ground_inds.to(dev), batch.ground_chunks
        else:
            # the in-box pixels' frame indices; their number is the ray count, so nonzero need not report it (no host sync)
            inds = (torch.nonzero_static(batch.mask_at_box.reshape(-1).to(dev), size=ret.acc_map.shape[1])[:, 0] if self.ground_inds is None
                    else self.ground_inds.to(dev))
            ranges = chunks(H * W, cfg.render_chunk_size)
        F = g_o.shape[0]
        acc_h = ret.acc_map[0]
        acc_g = torch.ones(F, device=dev)
        acc_g[inds] = 1 - acc_h
        gp = eng.ground_params()
        out = dotdict({k: torch.zeros(F, 3, device=dev) for k in ('rgb', 'surf', 'albedo', 'shade', 'spec')})
        out.depth = torch.zeros(F, device=dev)
        if cfg.vis_novel_light:              # cached for the per-probe re-shade (render_ground :541-543; 4 KB per frame pixel)
            L = cfg.env_h * cfg.env_w
            out.lvis, out.ldot = torch.zeros(F, L, device=dev), torch.zeros(F, L, device=dev)
        # The reference's chunks bound ITS memory; pixels are independent.  Consecutive chunks are rendered by ONE launch sequence of up to
        # cfg.ground_chunk_rays pixels (0: chunk exactly as the reference), every pixel clipped against the box its own chunk had reached
        # (the in-place growth, once per chunk incl. empty ones, is data independent): identical pixels, 17 x fewer launches at 512 x 512.
        limit = int(cfg.get('ground_chunk_rays', 0))
        group = []                                  # (a, b, box) of the chunks waiting for a launch

        def flush():
            if group:
                a0, b1 = group[0][0], group[-1][1]
                eng.render_ground_chunk(g_o[a0:b1], g_d[a0:b1], acc_g[a0:b1], group[0][2], probe, gp, {k: v[a0:b1] for k, v in out.items()},
                                        boxes=[g[2] for g in group], box_start=[g[0] - a0 for g in group] + [b1 - a0])
                group.clear()
        for a, b in ranges:
            bbox6 = self._grow_bounds(batch)
            if b <= a:
                continue
            if group and (a != group[-1][1] or len(group) == 32 or b - group[0][0] > limit):
                flush()
            group.append((a, b, bbox6))
        flush()
        if getattr(self, '_ground_n', None) is None or self._ground_n.device != dev:
            self._ground_n = torch.nn.functional.normalize(torch.tensor(cfg.ground_normal, dtype=torch.float32), dim=0).to(dev)     # constant of cfg
            torch.cuda.current_stream(dev).synchronize()         # made once, then read from whatever stream renders (frames in flight)
        n = self._ground_n
        grd = dotdict(rgb_map=out.rgb[None], surf_map=out.surf[None], albedo_map=out.albedo[None], roughness_map=torch.ones(1, F, device=dev),
                      spec_map=out.spec[None], norm_map=n[None, None].expand(1, F, 3), shade_map=out.shade[None],
                      cpts_map=torch.zeros(1, F, 3, device=dev), bpts_map=torch.zeros(1, F, 3, device=dev), depth_map=out.depth[None])
        if 'lvis' in out:
            grd.lvis_map, grd.ldot_map = out.lvis[None], out.ldot[None]
        grd.ray_o, grd.ray_d, grd.acc_map, grd.inds = g_o[None], g_d[None], acc_g[None], inds.to(torch.int64)[None]
        batch.mask_at_box[:] = True                          # :1103
        return grd
