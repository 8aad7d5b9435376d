"""Novel-light renderer, host-side mirror of lib/networks/renderer/novel_light_sphere_tracing.py:
the main pass computes intersection + visibility once, then every probe in batch.novel_lights is
re-shaded from the cached maps — here all probes in ONE fused ra_reshade launch (the BRDF,
visibility and area weights are light-independent), instead of the reference's Python loop.
With cfg.vis_ground_shading (the README's relight command, readme.md:64) the ground layer is re-shaded per probe
too (render_ground :70-99 -> ra_reshade_ground) and blended with the human layer per light (:191-213)."""
import time

import torch

from ..base_utils import dotdict
from . import sphere_tracing_renderer


class Renderer(sphere_tracing_renderer.Renderer):
    @torch.no_grad()
    def render(self, batch):
        cfg = self.cfg
        # the reference brackets the main pass with two device synchronisations to report its time as `diff` (:107-112); a caller
        # that does not read it (cfg.novel_light_timing = False) keeps the host running ahead of the GPU
        timing = bool(cfg.get('novel_light_timing', True))
        eng = self.net.engine()
        lights = batch.novel_lights
        if cfg.vis_rotate_light and len(lights):
            # rotating-light sequence (novel_light_sphere_tracing.py:163-171): every probe in rotate_ratio * env_w steps
            from ..relight_utils import rotate_envmap
            rotated = dotdict()
            for i in range(len(lights) * cfg.rotate_ratio * cfg.env_w):
                name, env = rotate_envmap(lights, i, cfg.rotate_ratio, cfg.env_w, cfg.env_image_w, eng)
                rotated[name] = env
            lights = rotated
        names = list(lights.keys())
        pr = lambda p: p[0] if p.ndim == 4 else p
        probes = torch.stack([pr(lights[n].probe) for n in names]).to(eng.device) if names else None
        # the frame is traced once and shaded under every probe: its key lights (cfg.key_light_share) are those of all of them
        env0 = self._envmap(batch)
        eng.set_key_probes(([env0.probe] if env0 is not None else []) + ([probes] if probes is not None else []))
        if timing:
            torch.cuda.synchronize()
        tick = time.perf_counter()
        try:
            main = super().render(batch)
        finally:
            eng.set_key_probes([])
        if timing:
            torch.cuda.synchronize()
        diff = time.perf_counter() - tick if timing else float('nan')
        visual = ['rgb_map', 'acc_map', 'norm_map', 'surf_map', 'bpts_map', 'cpts_map', 'spec_map', 'shade_map', 'depth_map',
                  'albedo_map', 'roughness_map', 'envmap']
        relight = dotdict()
        grd = main.get('ground', None)
        if 'main' in cfg.test_light:
            relight.main = dotdict({k: main[k] for k in visual if k in main})
            if grd is not None:
                relight.main = self.blend_output_(grd.acc_map, grd.inds, grd, relight.main, eng)      # :160-161
        if names:
            rgb, shade, spec = eng.reshade(main.ray_o, main.surf_map, main.norm_map, main.albedo_map, main.roughness_map,
                                           main.lvis_map, main.ldot_map, probes)
            if grd is not None:
                has_img = ['image' in lights[n] for n in names]
                images = torch.stack([pr(lights[n].image) for n in names]).to(eng.device) if all(has_img) else None
                if any(has_img) and images is None:
                    raise ValueError('either every novel light carries an `image` or none does')
                g_rgb, g_alb, g_shade, g_spec = eng.reshade_ground(grd.ray_d, grd.albedo_map, grd.lvis_map, grd.ldot_map, probes, images,
                                                                   cfg.ground_attach_envmap)
            for i, n in enumerate(names):
                human = main.copy()                                               # references, not copies (:183); lazily evaluated entries stay lazy
                human.pop('ground', None)
                human.rgb_map, human.shade_map, human.spec_map = rgb[i][None], shade[i][None], spec[i][None]
                if grd is not None:
                    ground = dotdict({k: grd[k] for k in visual if k in grd})
                    ground.rgb_map, ground.albedo_map, ground.shade_map, ground.spec_map = g_rgb[i][None], g_alb[i][None], g_shade[i][None], g_spec[i][None]
                    human = dotdict({k: human[k] for k in visual if k in human})
                    human = self.blend_output_(grd.acc_map, grd.inds, ground, human, eng)             # per light (:205-211)
                human.envmap = dotdict(probe=lights[n].probe)
                relight[n] = human
        relight.diff = diff
        return relight
