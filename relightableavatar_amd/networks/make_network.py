"""make_network(cfg): same plugin mechanism as lib/networks/make_network.py:4-7 —
importlib.import_module(cfg.network_module).Network().  The module reads the active cfg from
relightableavatar_amd.config (set_active_cfg) the way the reference reads its global cfg."""
import importlib

from .. import config


def make_network(cfg):
    config.check_supported(cfg)
    config.set_active_cfg(cfg)
    return importlib.import_module(cfg.network_module).Network()
