"""Device-side mirror of the reference's per-frame ray set-up.

    get_rays_within_bounds(H, W, K, R, T, bounds)   lib/utils/data_utils.py:925-938 (+ get_rays :827-845,
                                                    get_full_near_far :860-875), called by pose_dataset.py:66

Same name, argument meaning and return order as the reference; the arrays come back as device tensors
(no H*W*32 B upload per frame) and `mask_at_box` is the (H, W) box mask.  There is no CPU fallback.
"""
from .engine import Engine


def get_rays_within_bounds(H, W, K, R, T, bounds, engine: Engine):
    o = engine.gen_rays(H, W, K, R, T, bounds)
    return o.ray_o, o.ray_d, o.near, o.far, o.mask_at_box
