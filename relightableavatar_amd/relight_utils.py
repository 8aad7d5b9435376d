"""Device-side mirror of the reference's environment-map utilities.

    rotate_envmap(novel_light, index, repeat, probe_width, image_width)   lib/utils/relight_utils.py:57-103
    add_light_probe(rgb, probe, batch, cfg)                               lib/utils/relight_utils.py:38-54 (+ gen_light_dir :9-35)
    gen_light_xyz(env_h, env_w, env_r)                                    lib/utils/relight_utils.py:423-465 (host, once per network)

Same names, argument meaning and return values as the reference plus the engine that owns the HIP context; there is no
CPU fallback.
"""
import math

import torch

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


def gen_light_xyz(env_h: int, env_w: int, env_r: float):
    """Light-probe geometry; restates lib/utils/relight_utils.py:423-465 (lat/long cell centres)."""
    lat_half = math.pi / env_h / 2
    lng_half = 2 * math.pi / env_w / 2
    lats = torch.linspace(math.pi / 2 - lat_half, -math.pi / 2 + lat_half, env_h)
    lngs = torch.linspace(math.pi - lng_half, -math.pi + lng_half, env_w)
    lngs, lats = torch.meshgrid(lngs, lats, indexing='xy')  # (eH, eW)
    z = env_r * torch.sin(lats)
    x = env_r * torch.cos(lats) * torch.cos(lngs)
    y = env_r * torch.cos(lats) * torch.sin(lngs)
    xyz = torch.stack((x, y, z), dim=-1)
    sin_colat = torch.sin(math.pi / 2 - lats)
    area = 4 * math.pi * sin_colat / torch.sum(sin_colat)
    return xyz, area
