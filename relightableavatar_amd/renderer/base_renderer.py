"""Volume renderer, host-side mirror of lib/networks/renderer/base_renderer.py (eval path):
uniform samples between near/far, Network.forward per sample, alpha compositing — one
ra_render_volume_chunk call per chunk of max(cfg.render_chunk_size, cfg.volume_chunk_rays) rays."""
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

    @torch.no_grad()
    def render(self, batch):
        cfg = self.cfg
        eng = self.net.set_frame(batch)
        eng.begin_render()
        dev = eng.device
        f = lambda t: t[0].to(dev, torch.float32).contiguous()
        ray_o, ray_d = f(batch.ray_o), f(batch.ray_d)
        near, far = f(batch.near), f(batch.far)           # clipped to [cfg.clip_near, .] / [., cfg.clip_far] by the library (base_renderer.py:120-121)
        P = ray_o.shape[0]
        # every output is written for every ray by the compositor: no zero fill, no torch kernel in the frame
        full = dotdict(rgb=torch.empty(P, 3, device=dev), acc=torch.empty(P, device=dev), depth=torch.empty(P, device=dev),
                       norm=torch.empty(P, 3, device=dev), cpts=torch.empty(P, 3, device=dev), bpts=torch.empty(P, 3, device=dev),
                       resd=torch.empty(P, 3, device=dev))
        # the reference's render_chunk_size bounds its activation memory on a 24 GB card; rays are independent here (the pixels do
        # not depend on the chunking: test_full_size_properties_volume_config2), so launches are sized for the device instead:
        # 428 k full queries per 8192-ray chunk are 6.5 rounds of 256-point tiles over the 256 CUs, i.e. 7 % idle in the last one
        for a, b in chunks(P, max(cfg.render_chunk_size, cfg.get('volume_chunk_rays', 0))):
            eng.render_volume_chunk(ray_o[a:b], ray_d[a:b], near[a:b], far[a:b], cfg.n_samples, cfg.dist_th,
                                    {k: v[a:b] for k, v in full.items()})
        ret = dotdict()
        ret.depth_map = full.depth[None]
        ret.cpts_map, ret.bpts_map, ret.resd_map = full.cpts[None], full.bpts[None], full.resd[None]
        ret.norm_map = full.norm[None]
        ret.rgb_map = full.rgb[None]
        ret.acc_map = full.acc[None]
        return ret
