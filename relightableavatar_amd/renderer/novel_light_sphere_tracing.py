"""Novel-light renderer, host-side mirror of lib/networks/renderer/novel_light_sphere_tracing.py:
the main pass computes intersection + visibility once, then every probe in batch.novel_lights is
re-shaded from the cached maps — here all probes in ONE fused ra_reshade launch (the BRDF,
visibility and area weights are light-independent), instead of the reference's Python loop."""
import time

import torch

from ..base_utils import dotdict
from . import sphere_tracing_renderer


class Renderer(sphere_tracing_renderer.Renderer):
    @torch.no_grad()
    def render(self, batch):
        cfg = self.cfg
        torch.cuda.synchronize()
        tick = time.perf_counter()
        main = super().render(batch)
        torch.cuda.synchronize()
        diff = time.perf_counter() - tick
        visual = ['rgb_map', 'acc_map', 'norm_map', 'surf_map', 'bpts_map', 'cpts_map', 'spec_map', 'shade_map', 'depth_map',
                  'albedo_map', 'roughness_map', 'envmap']
        relight = dotdict()
        if 'main' in cfg.test_light:
            relight.main = dotdict({k: main[k] for k in visual if k in main})
        lights = batch.novel_lights
        if cfg.vis_rotate_light and len(lights):
            # rotating-light sequence (novel_light_sphere_tracing.py:163-171): every probe in rotate_ratio * env_w steps
            from ..relight_utils import rotate_envmap
            eng = self.net.engine()
            rotated = dotdict()
            for i in range(len(lights) * cfg.rotate_ratio * cfg.env_w):
                name, env = rotate_envmap(lights, i, cfg.rotate_ratio, cfg.env_w, cfg.env_image_w, eng)
                rotated[name] = env
            lights = rotated
        names = list(lights.keys())
        if names:
            eng = self.net.engine()
            probes = torch.stack([(lambda p: p[0] if p.ndim == 4 else p)(lights[n].probe) for n in names]).to(eng.device)
            rgb, shade, spec = eng.reshade(main.ray_o, main.surf_map, main.norm_map, main.albedo_map, main.roughness_map,
                                           main.lvis_map, main.ldot_map, probes)
            for i, n in enumerate(names):
                human = dotdict({k: main[k] for k in main if k not in ('lvis_map', 'ldot_map')})
                human.rgb_map, human.shade_map, human.spec_map = rgb[i][None], shade[i][None], spec[i][None]
                human.envmap = dotdict(probe=lights[n].probe)
                relight[n] = human
        relight.diff = diff
        return relight
