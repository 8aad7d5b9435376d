"""Device-side mirror of the reference's per-frame batch assembly (SURVEY.md 8f, rows N2 + N3).

    get_rays_within_bounds(H, W, K, R, T, bounds)   lib/utils/data_utils.py:925-938 (+ get_rays :827-845, get_full_near_far :860-875),
                                                    called by pose_dataset.py:66
    get_blend(...)                                  lib/datasets/base_dataset.py:308-397 (cfg.use_geometry path): get_lbs_params,
                                                    get_rigid_transform (net_utils.py:1164-1183 / data_utils.py:1004-1069),
                                                    pose_points_to_tpose_points / tpose_points_to_pose_points / pose_points_to_world_points
                                                    (blend_utils.py:264-313), Meshes.verts_normals (pytorch3d, restated), get_bounds (:616-622)
    DeviceFrameLoader                               pose_dataset.Dataset.__getitem__ (pose_dataset.py:45-113) for a camera + a pose
                                                    sequence: the §8b batch of one frame, assembled on the device

Same names, argument meaning and return order as the reference where a counterpart exists; arrays come back as device tensors (no
H * W * 32 B upload per frame).  There is no CPU fallback.

The reference's loader does this work on the CPU (numpy LBS, pytorch3d normals, numpy ray set-up: 56 + 7 ms per 512 x 512 frame) and its
frame loop waits for it.  Here N3 and N2 are a dozen small launches on the frame's own stream and NOTHING waits for them: the rays of a
frame are generated one pipeline turn before the frame is rendered, and the two numbers the host needs to size the frame's outputs — the
in-box ray count and the body's box — arrive in pinned memory behind an event that has long passed when they are read.
"""
import numpy as np
import torch

from .base_utils import dotdict
from .engine import Engine


def get_rays_within_bounds(H, W, K, R, T, bounds, engine: Engine):
    o = engine.gen_rays(H, W, K, R, T, bounds)
    return o.ray_o, o.ray_d, o.near, o.far, o.mask_at_box


def get_blend(engine: Engine, poses, tjoints, parents, tverts, weights, big_A, faces, Rh, Th) -> dotdict:
    """the frame-state keys the networks consume (`Network.set_frame(batch)`), as device tensors with the leading batch dimension of the
    reference's collated batch"""
    o = engine.pose_frame(poses, tjoints, parents, tverts, weights, big_A, faces, Rh, Th)
    return dotdict(A=o.A[None], joints=o.joints[None], pverts=o.pverts[None], wverts=o.wverts[None], pnorm=o.pnorm[None], R=o.R[None],
                   Th=o.Th.reshape(1, 1, 3), poses=o.poses[None], pbounds=o.pbounds[None], wbounds=o.wbounds[None], tpose_verts=o.tverts[None])


class DeviceFrameLoader:
    """One camera, one body, a sequence of poses -> the §8b batches of its frames, assembled on the device.

        loader = DeviceFrameLoader(H, W, K, R, T, tjoints, parents, tverts, weights, big_A, faces)
        pend = loader.issue(engine, poses_f, Rh_f, Th_f)        # queues N3 + N2 on the current stream; returns at once
        ...                                                     # (typically: render the frames issued before)
        batch = loader.batch(pend)                              # waits for an EVENT (the kernels above), not for the stream

    `mask_to_host`: the batch also carries the host copy of mask_at_box (`mask_host`) that a shard plan is computed from (shard.make_plan)."""

    def __init__(self, H, W, K, R, T, tjoints, parents, tverts, weights, big_A, faces, device=None, padding=0.05, mask_to_host=False):
        self.H, self.W, self.K, self.R, self.T = int(H), int(W), K, R, T
        dev = tverts.device if isinstance(tverts, torch.Tensor) and tverts.is_cuda else device
        f32 = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).to(dev, torch.float32).contiguous()
        self.tverts, self.weights = f32(tverts).reshape(-1, 3), f32(weights)
        self.big_A_host = np.ascontiguousarray(big_A.detach().cpu().numpy() if isinstance(big_A, torch.Tensor) else big_A, dtype=np.float32).reshape(-1, 4, 4)
        self.big_A = f32(self.big_A_host)
        # converted once: Engine.pose_frame takes arrays of the right type as they are (int64 -> int32 of 13 776 faces costs 20 us per frame)
        npa = lambda a, t: np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=t)
        self.tjoints, self.parents, self.faces = npa(tjoints, np.float32).reshape(-1, 3), npa(parents, np.int32), npa(faces, np.int32).reshape(-1, 3)
        self.padding, self.mask_to_host = float(padding), bool(mask_to_host)
        self.meta = dotdict(H=torch.tensor([self.H]), W=torch.tensor([self.W]), frame_index=torch.tensor([0]), view_index=torch.tensor([0]))

    def issue(self, engine: Engine, poses, Rh, Th) -> dotdict:
        body = engine.pose_frame(poses, self.tjoints, self.parents, self.tverts, self.weights, self.big_A_host, self.faces, Rh, Th, self.padding)
        rays = engine.gen_rays_async(self.H, self.W, self.K, self.R, self.T, body.wbounds, mask_to_host=self.mask_to_host)
        return dotdict(body=body, rays=rays)

    def batch(self, pend) -> dotdict:
        rays, body = pend.rays.result(), pend.body
        b = dotdict(ray_o=rays.ray_o[None], ray_d=rays.ray_d[None], near=rays.near[None], far=rays.far[None],
                    R=body.R[None], Th=body.Th.reshape(1, 1, 3), poses=body.poses[None], weights=self.weights[None], A=body.A[None],
                    big_A=self.big_A[None], pverts=body.pverts[None], pnorm=body.pnorm[None], tverts=body.tverts[None],
                    wbounds=body.wbounds[None], mask_at_box=rays.mask_at_box.reshape(1, -1), meta=self.meta)
        # the renderers grow the box once per chunk and hand it to the launches as host numbers: the mirror saves the read-back
        b.wbounds_host, b.wbounds_host_version = rays.wbounds_host, b.wbounds._version
        if 'mask_host' in rays:
            b.mask_host = rays.mask_host
        return b
