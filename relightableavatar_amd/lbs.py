"""Device-side mirror of the reference's per-frame body state.

    get_lbs_params / get_blend          lib/datasets/base_dataset.py:308-397 (cfg.use_geometry path)
    get_rigid_transform                 lib/utils/net_utils.py:1164-1183 (smplx.lbs), data_utils.py:1004-1069
    pose_points_to_tpose_points, tpose_points_to_pose_points, pose_points_to_world_points   lib/utils/blend_utils.py:264-313
    Meshes.verts_normals                pytorch3d (restated from its published algorithm)
    get_bounds                          lib/utils/data_utils.py:616-622

`get_blend` returns the frame-state keys the networks consume (`Network.set_frame(batch)`), as device tensors with the
leading batch dimension of the reference's collated batch.  There is no CPU fallback.
"""
from .base_utils import dotdict
from .engine import Engine


def get_blend(engine: Engine, poses, tjoints, parents, tverts, weights, big_A, faces, Rh, Th) -> dotdict:
    o = engine.pose_frame(poses, tjoints, parents, tverts, weights, big_A, faces, Rh, Th)
    ret = dotdict(A=o.A[None], joints=o.joints[None], pverts=o.pverts[None], wverts=o.wverts[None], pnorm=o.pnorm[None], R=o.R[None],
                  Th=None, pbounds=o.pbounds[None], wbounds=o.wbounds[None], tpose_verts=o.tverts[None])
    return ret
