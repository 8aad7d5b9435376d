"""Device-side mirror of the reference's environment-map utilities.

    rotate_envmap(novel_light, index, repeat, probe_width, image_width)   lib/utils/relight_utils.py:57-103
    add_light_probe(rgb, probe, batch, cfg)                               lib/utils/relight_utils.py:38-54 (+ gen_light_dir :9-35)

Same names, argument meaning and return values as the reference plus the engine that owns the HIP context; there is no
CPU fallback.
"""
from .base_utils import dotdict


def rotate_envmap(novel_light, index, repeat, probe_width, image_width, engine):
    keys = list(novel_light.keys())
    if repeat <= 0:
        return keys[index], novel_light[keys[index]]
    n_rotation = probe_width * repeat
    i, j = index // n_rotation, index % n_rotation
    name = f'{keys[i]}-{j:04d}'
    envmap = novel_light[keys[i]]
    eW = envmap.probe.shape[-2]
    uW = eW * repeat
    out = dotdict(probe=engine.shift_envmap(envmap.probe, eW / uW * j))
    if 'image' in envmap:
        out.image = engine.shift_envmap(envmap.image, envmap.image.shape[-2] / uW * j)
    return name, out


def add_light_probe(rgb, probe, batch, cfg, engine):
    H, W = int(batch.meta.H.item()), int(batch.meta.W.item())
    uW = int(W * cfg.probe_size_ratio)
    uH = int(uW * cfg.env_h / cfg.env_w)
    return engine.add_light_probe(rgb, probe, H, W, batch.cam_R[0], uH, uW).reshape(rgb.shape)
