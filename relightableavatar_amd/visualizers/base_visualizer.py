"""Device-side mirror of the reference visualiser's map -> image step (third item).

    Visualizer.generate_image(output, batch, type)      lib/visualizers/base_visualizer.py:54-231
    Output                                              lib/config/config.py:364-378

Same name, arguments and return value (an H x W x 3|4 numpy image) as the reference's static method; the per-type
normalisation, the scatter of the in-box rays into the frame and the alpha plane run in the HIP library
(ra_map_to_image), the light-probe inset in ra_add_light_probe.  Writing images to disk, ground-truth images and the
deprecated Semantic / Feature types stay out of scope (no GPU work in them).  There is no CPU fallback.
One deviation: Output.Depth stretches between the 1 % quantiles of the HIT rays' depths; with fewer hit rays than 1 % of the rays the
reference's topk raises, the build clamps the rank to the hit count (the count lives on the device; no read-back per image).
"""
import ctypes as C
from enum import Enum, auto

import torch

from .. import config
from .._lib import check, ra_image_params
from ..relight_utils import add_light_probe


class Output(Enum):
    Semantic = auto()
    Feature = auto()
    Surface = auto()
    Residual = auto()
    Depth = auto()
    Alpha = auto()
    Normal = auto()
    Specular = auto()
    Albedo = auto()
    Roughness = auto()
    Shading = auto()
    Rendering = auto()
    Envmap = auto()


# which map of `output` each type reads (base_visualizer.py:57-186)
_MAPS = {Output.Normal: 'norm_map', Output.Depth: 'depth_map', Output.Shading: 'shade_map', Output.Albedo: 'albedo_map',
         Output.Roughness: 'roughness_map', Output.Rendering: 'rgb_map', Output.Specular: 'spec_map', Output.Alpha: 'acc_map'}


class Visualizer:
    engine = None      # the Engine that owns the HIP context (set once: Visualizer.engine = net.engine())

    @staticmethod
    def generate_image(output, batch, type: Output = Output.Rendering, engine=None):
        cfg = config.active_cfg()
        eng = engine or Visualizer.engine
        if eng is None:
            raise RuntimeError('Visualizer.generate_image needs the engine of the network (Visualizer.engine = net.engine()); no CPU fallback')
        H, W = int(batch.meta.H.item()), int(batch.meta.W.item())
        if type == Output.Envmap:
            return output.envmap.probe[0].detach().cpu().numpy()                  # :173-174
        if type in (Output.Semantic, Output.Feature):
            raise NotImplementedError(f'deprecated output type {type} is not mirrored')
        dev = eng.device
        f = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
        acc = f(output.acc_map[0])
        b = None
        if type == Output.Surface:
            a = output.cpts_map[0] if 'cpts_map' in output else output.surf_map[0]
        elif type == Output.Residual:
            a, b = output.cpts_map[0], f(output.bpts_map[0])
        elif type == Output.Depth and cfg.get('vis_median_depth', False):
            a = output.median_map[0]
        else:
            a = output[_MAPS[type]][0]
        a = f(a)
        P = a.shape[0]
        p = ra_image_params(type=type.value, H=H, W=W, bg_brightness=float(cfg.bg_brightness),
                            normalize=int(cfg.get('normalize_specular', True) if type == Output.Specular else cfg.get('normalize_shading', False)),
                            tonemap=int(cfg.get('tonemapping_albedo', True)), min_clip=float(cfg.get('min_clip', 1.0)))
        if type == Output.Normal:
            p.cam_R = (C.c_float * 9)(*[float(v) for v in batch.cam_R[0].reshape(-1).tolist()])
        if type == Output.Surface:
            p.tbounds = (C.c_float * 6)(*[float(v) for v in batch.tbounds[0].reshape(-1).tolist()])
        pix = None
        if P != H * W:                                                             # rgb_map.ndim == 2 branch (:188-193): scatter through mask_at_box
            pix = batch.mask_at_box[0].reshape(-1).to(dev).nonzero()[:, 0].contiguous()
            assert pix.numel() == P, 'mask_at_box does not match the number of rays'
        image = torch.empty(H * W, 3, device=dev)
        store_alpha = bool(cfg.get('store_alpha_channel', True))
        alpha = torch.empty(H * W, device=dev) if store_alpha else None
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        check(eng.lib.ra_map_to_image(eng.ctx, C.byref(p), ptr(a), ptr(b), ptr(acc), ptr(pix), P, ptr(image), ptr(alpha), eng.stream), 'ra_map_to_image')
        img = image.view(H, W, 3)
        if cfg.probe_size_ratio > 0 and output.get('envmap', None) is not None:   # :197-199
            img = add_light_probe(img.reshape(1, H * W, 3), output.envmap.probe, batch, cfg, eng)[0].view(H, W, 3)
        if store_alpha:                                                            # :201-208
            img = torch.cat([img, alpha.view(H, W, 1)], dim=-1)
        if 'orig_H' in batch and 'orig_W' in batch:                                # :214-215: a cropped batch is pasted into the full-size frame
            img = Visualizer.fill_image(img, int(batch.orig_H.item()), int(batch.orig_W.item()), batch.crop_bbox[0], cfg)
        return img.detach().cpu().numpy()

    @staticmethod
    def fill_image(img, orig_H, orig_W, bbox, cfg=None):
        """Visualizer.fill_image (base_visualizer.py:233-239): the crop goes back to its place in an orig_H x orig_W frame of
        cfg.bg_brightness.  Like the reference's, the frame has 3 channels: a 4-channel image (cfg.store_alpha_channel) does not
        fit and raises, as the reference's slice assignment does."""
        cfg = cfg or config.active_cfg()
        full = img.new_ones(orig_H, orig_W, 3) * cfg.bg_brightness
        x0, y0, x1, y1 = int(bbox[0, 0]), int(bbox[0, 1]), int(bbox[1, 0]), int(bbox[1, 1])
        full[y0:y1, x0:x1] = img[:y1 - y0, :x1 - x0]
        return full
