"""Sphere-tracing renderer, host-side mirror of lib/networks/renderer/sphere_tracing_renderer.py
(Renderer.render :1066-1115, get_pixel_value :981-1039, render_human :551-784; eval path, no
ground pass).  One ra_render_sphere_chunk call per chunk of cfg.render_chunk_size rays."""
import torch
from torch import nn

from .. import config
from ..base_utils import dotdict
from .chunking import chunks


class Renderer(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.cfg = config.active_cfg()

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
        if cfg.vis_ground_shading:
            raise NotImplementedError('ground pass (render_ground) is a SURVEY.md section 8f "next" row')
        eng = self.net.set_frame(batch)
        dev = eng.device
        f = lambda t: t[0].to(dev, torch.float32).contiguous()
        ray_o, ray_d, near, far = f(batch.ray_o), f(batch.ray_d), f(batch.near), f(batch.far)
        P = ray_o.shape[0]
        envmap = self._envmap(batch)
        probe = None
        if envmap is not None:
            probe = envmap.probe
            probe = (probe[0] if probe.ndim == 4 else probe).to(dev, torch.float32).contiguous()
        relit = bool(cfg.relighting)
        params = eng.sphere_params()
        names = ['rgb', 'acc', 'depth', 'surf', 'norm', 'cpts', 'bpts', 'resd', 'ray_o']
        if eng.relight:
            names += ['albedo', 'roughness']
        if relit:
            names += ['shade']
            if cfg.vis_specular_map:
                names += ['spec']
            if cfg.vis_novel_light:
                names += ['lvis', 'ldot']
        width = dict(acc=0, depth=0, roughness=0, lvis=cfg.env_h * cfg.env_w, ldot=cfg.env_h * cfg.env_w)
        full = dotdict()
        for k in names:
            w = width.get(k, 3)
            full[k] = torch.zeros((P, w) if w else (P,), device=dev)
        for a, b in chunks(P, cfg.render_chunk_size):
            # quirk (sphere_tracing_renderer.py:1020-1022): the box grows IN PLACE on the batch every chunk
            wb = batch.wbounds
            wb[:, 0] -= cfg.env_lvis.bbox_margin
            wb[:, 1] += cfg.env_lvis.bbox_margin
            bbox6 = wb[0].reshape(-1).tolist()
            eng.render_sphere_chunk(ray_o[a:b], ray_d[a:b], near[a:b], far[a:b], bbox6, probe, params,
                                    {k: v[a:b] for k, v in full.items()})
        ret = dotdict()
        ret.acc_map = full.acc[None]
        ret.ray_o = full.ray_o[None]
        ret.surf_map, ret.depth_map = full.surf[None], full.depth[None]
        ret.cpts_map, ret.bpts_map, ret.resd_map, ret.norm_map = full.cpts[None], full.bpts[None], full.resd[None], full.norm[None]
        if eng.relight:
            ret.albedo_map, ret.roughness_map = full.albedo[None], full.roughness[None]
        ret.rgb_map = full.rgb[None]
        if relit:
            ret.shade_map = full.shade[None]
            if 'spec' in full:
                ret.spec_map = full.spec[None]
            if 'lvis' in full:
                ret.lvis_map, ret.ldot_map = full.lvis[None], full.ldot[None]
        ret.envmap = envmap
        return ret
