"""relightableavatar_amd: MI355X-native per-ray render hot path of RelightableAvatar.

Only the path named by BASELINE.json:north_star lives here:
  csrc/       hand-written gfx950 HIP kernels + the C-ABI (include/relightableavatar.h)
  _lib.py     ctypes binding of the C-ABI; fails loudly when the .so is missing
  config.py   the resolved constants the reference reads from its global ``cfg``
  networks/   host-side mirror of lib/networks/{deform,relight} (same class names, state_dict keys)
  renderer/   host-side mirror of lib/networks/renderer/* (Renderer.render(batch) -> dotdict)
  shard.py    ray sharding across ranks + RCCL all_gather
  synthetic.py  build-owned deterministic weights / body / camera / envmaps
"""
from .base_utils import dotdict  # noqa: F401

__version__ = "0.1.0"
