"""make_renderer(cfg, network): lib/networks/renderer/make_renderer.py:5-8."""
import importlib

from .. import config


def make_renderer(cfg, network):
    config.check_supported(cfg)
    config.set_active_cfg(cfg)
    return importlib.import_module(cfg.renderer_module).Renderer(network)
