"""RelightableAvatar network, host-side mirror of lib/networks/relight/relight_network.py:
AniSDF + albedo/roughness heads + optimisable achromatic env map + light-probe buffers."""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ... import config
from ...relight_utils import gen_light_xyz
from ...base_utils import dotdict
from ..deform import base_network
from ..deform.base_network import _Embedder, _MLP, _buffer


class Microfacet:
    """parameters of the GGX model evaluated inside the shading kernel (relight_utils.py:468-483)."""

    def __init__(self, f0=0.04, lambert_only=False, glossy_only=False, cancel_cosine=True):
        self.f0, self.lambert_only, self.glossy_only, self.cancel_cosine = f0, lambert_only, glossy_only, cancel_cosine


class Network(base_network.Network):
    def __init__(self):
        super().__init__()
        cfg = self.cfg
        for p in self.render_network.parameters():        # freeze_module (relight_network.py:37)
            p.requires_grad_(False)
        self.xyz_embedder = _Embedder(10)
        self.view_embedder = _Embedder(4)
        self.albedo_network = _MLP(cfg.feat_dim, cfg.relight_network_width, cfg.relight_network_depth, 3)
        self.roughness_network = _MLP(cfg.feat_dim, cfg.relight_network_width, cfg.relight_network_depth, 1)
        for m in (self.albedo_network, self.roughness_network):
            for l in m.linears:
                nn.init.kaiming_normal_(l.weight)
        self.microfacet = Microfacet(f0=cfg.fresnel_f0, lambert_only=cfg.lambert_only, glossy_only=cfg.glossy_only)
        ch = 1 if cfg.achro_light else 3
        self.global_env_map_ = nn.Parameter(torch.rand(cfg.env_h * cfg.envmap_upscale, cfg.env_w * cfg.envmap_upscale, ch) * cfg.envmap_init_intensity)
        xyz, area = gen_light_xyz(cfg.env_h, cfg.env_w, cfg.env_r)
        self.light_xyz_ = _buffer(xyz)
        self.light_area = _buffer(area)
        self.light_sharp = _buffer(1 / (area / math.pi).sqrt())
        self.env_h, self.env_w, self.env_r = cfg.env_h, cfg.env_w, cfg.env_r

    @property
    def light_xyz(self):
        return self.light_xyz_

    def invalidate_env_map(self):
        """drop the eval-mode cache of `global_env_map` (needed after a write to `global_env_map_` that bypasses autograd's version
        counter, e.g. through `.data`)"""
        self._env_cache_key = None

    def load_state_dict(self, *a, **kw):
        self._env_cache_key = None
        return super().load_state_dict(*a, **kw)

    @property
    def global_env_map(self):
        """softplus of the optimisable map (relight_network.py:86-89).  In eval mode the parameter does not change from frame to frame:
        the result is kept until the parameter is written again (its version counter), so a render loop launches nothing for it."""
        p = self.global_env_map_
        if self.training or torch.is_grad_enabled() and p.requires_grad:
            return F.softplus(p.expand(*p.shape[:2], 3))
        # `.data` writes do not bump the version counter: call invalidate_env_map() after one (load_state_dict does)
        key = (p._version, p.data_ptr(), p.device)
        if getattr(self, '_env_cache_key', None) != key:
            with torch.no_grad():
                self._env_cache = F.softplus(p.expand(*p.shape[:2], 3)).contiguous()
            self._env_cache_key = key
        return self._env_cache
