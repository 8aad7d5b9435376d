"""Build-owned deterministic synthetic inputs (SURVEY.md §8d "synthetic inputs").

Nothing here comes from the reference: there is no network for datasets or checkpoints, so
bench.py, smoke() and the tests render a synthetic avatar:

* weights: numpy PCG64 streams keyed by (seed, tensor name); shapes and key names are the
  reference's state_dict (SURVEY.md §8b "weights on disk"), statistics follow its initialisers
  (SDF geometric init net_utils.py:1303-1324; nn.Linear default U(+-1/sqrt(fan_in)); kaiming
  normal for the material heads relight_network.py:46-47) so the zero level set is a blob of
  radius ~0.4-0.5 in big-pose space.
* body: N=6890 Fibonacci-sphere "SMPL" vertices, J=52 random rigid bone transforms.
* camera: pinhole at (0,0,-2), focal 0.8*H; rays clipped to the body AABB the way
  lib/utils/data_utils.py:827-875,925-938 does (restated, numpy).
* envmaps: lognormal HDR-like 16x32 probes + one OLAT-style probe.
"""
import math
import zlib

import numpy as np
import torch

from .base_utils import dotdict
from .relight_utils import gen_light_xyz          # noqa: F401  (re-exported: the synthetic state_dict carries the light buffers)

N_VERTS = 6890
N_BONES = 52


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _uniform(seed, name, shape, bound):
    return torch.from_numpy(_rng(seed, name).uniform(-bound, bound, size=shape).astype(np.float32))


def _normal(seed, name, shape, mean, std):
    return torch.from_numpy((_rng(seed, name).standard_normal(size=shape) * std + mean).astype(np.float32))


def _linear(sd, seed, prefix, fan_in, fan_out, kind='uniform'):
    if kind == 'uniform':  # nn.Linear default
        b = 1.0 / math.sqrt(fan_in)
        sd[prefix + '.weight'] = _uniform(seed, prefix + '.weight', (fan_out, fan_in), b)
    elif kind == 'kaiming':  # kaiming_normal_, fan_in, gain sqrt(2)
        sd[prefix + '.weight'] = _normal(seed, prefix + '.weight', (fan_out, fan_in), 0.0, math.sqrt(2.0 / fan_in))
    sd[prefix + '.bias'] = _uniform(seed, prefix + '.bias', (fan_out,), 1.0 / math.sqrt(fan_in))


def _weight_norm(sd, seed, prefix, w: torch.Tensor, bias: torch.Tensor, jitter=0.05):
    """store W as (weight_g, weight_v) with g != |v| so the fold is exercised."""
    v = w
    g = v.norm(dim=1, keepdim=True) * (1.0 + _uniform(seed, prefix + '.g', (w.shape[0], 1), jitter))
    sd[prefix + '.weight_g'] = g
    sd[prefix + '.weight_v'] = v
    sd[prefix + '.bias'] = bias


def freq_bands(multires: int) -> torch.Tensor:
    fb = 2.0 ** torch.linspace(0.0, multires - 1, steps=multires)
    return fb[:, None, None].expand(multires, 2, 1).clone()


# 'sharp' weights: relative scale (to the coordinate columns' sqrt(2)/sqrt(out)) of the SDF net's encoding columns per frequency band
# 2^0 .. 2^7.  The geometric init leaves them at ~0 (net_utils.py:1310-1318: zero; here 0.02 so the path is exercised); a trained net
# uses them: this profile puts 1.5x .. 4x that energy on the 78 cm .. 5 cm bands and gives centimetre-scale surface detail (11.7 mm rms /
# 47 mm peak-to-peak over a 21 cm arc after removing a cubic, |grad sdf| = 1.33 +- 0.54 near the zero set instead of 0.94 +- 0.30, 1.14
# zero crossings per radial line: overhangs).  The SAME scale on every band (1.0, "the hidden columns' scale") is not a distance field
# at all: |grad| ~ 20 and sin^2 + cos^2 = 1 adds a constant 24 rho^2 under the root of the geometric init's |W x| — no zero set is left.
SHARP_BANDS = (0.02, 0.02, 0.02, 0.05, 0.08, 0.06, 0.04, 0.03)
SHARP_SDF_BIAS = -0.559         # keeps the zero set's mean radius at 0.42 m under SHARP_BANDS (the constant above moves it inwards)
# residual deformation: 2.5 cm instead of the default init's 1.7 mm, with the Jacobian of a smooth warp (|d resd / d bpts|_2 = 0.18 mean,
# 0.38 max: wrinkles of centimetres over decimetres).  The gain alone (the first version of this kind) gives |J| = 0.87 mean / 1.7 max —
# a warp that folds space, which no trained net has (the reference regularises resd) and which amplifies every rounding of the encoding
# path 512 x: the encoding columns of the net's two input layers therefore fall off by 0.7 per band above 2^4
SHARP_RESD_GAIN = 16.0
SHARP_RESD_DECAY, SHARP_RESD_BAND0 = 0.7, 4


FRONT_LIGHT_DIR = (-0.80, -0.30, -0.52)
# the hard-case body (tests/golden/make_golden.py `split_body`, bench.py --body split): the cap of the template around split_axis (cosine
# to the axis > 0.8: ~260 vertices move fully, ~800 in the sleeve behind them) is pulled 0.57 m out, tangentially: a horn from the top of
# the body over its shoulder towards the camera, with a gap between horn and body that make_state_dict(env='front')'s key light shines through
SPLIT_BODY_KW = dict(split_axis=[-0.25, -0.94, -0.26], split_offset=[-0.45, 0.0, -0.35], split_cos=0.8, skin_sharpness=6.0)


def _env_pixel(d, eh, ew):
    """(row, column) of the light map's texel that direction d samples (relight_utils.py:106-127: theta = acos(d_z), phi = atan2(d_y, d_x),
    grid (-phi / pi, 2 theta / pi - 1), align_corners=False)"""
    n = math.sqrt(sum(c * c for c in d))
    theta, phi = math.acos(d[2] / n), math.atan2(d[1], d[0])
    return ((2 * theta / math.pi) * eh / 2 - 0.5, (1 - phi / math.pi) * ew / 2 - 0.5)


def make_state_dict(seed: int = 0, relight: bool = True, cfg=None, kind: str = 'init', env: str = 'back') -> dict:
    """state_dict with the reference's key names.  kind: 'init' — the reference initialisers' statistics; 'sharp' —
    trained-like: live high-frequency encoding columns in the SDF net (SHARP_BANDS) and a centimetre-scale residual deformation.
    env: where the learned light map's lobes sit — 'back' (behind / beside the body as seen from the synthetic camera: the
    self-shadowing of a convex body is all penumbra) or 'front' (one strong lobe on the camera's side, so that a body part in front
    of another casts a shadow the camera sees)."""
    from .config import default_cfg
    cfg = cfg or default_cfg()
    assert kind in ('init', 'sharp') and env in ('back', 'front'), (kind, env)
    bands = SHARP_BANDS if kind == 'sharp' else (0.02,) * 8
    band_scale = torch.tensor([bands[min(k, len(bands) - 1)] / 0.02 for k in range(cfg.sdf_res)]).repeat_interleave(6)
    sd = {}
    xyz_dim = 3 + 6 * cfg.xyz_res        # 63
    sdf_dim = 3 + 6 * cfg.sdf_res        # 51
    view_dim = 3 + 6 * cfg.view_res      # 27
    W = 256
    # residual deformation MLP (base_network.py:14-42): 9 plain linears, skip at 4 (x first)
    p = 'residual_deformation_network'
    sd[p + '.embedder.freq_bands'] = freq_bands(cfg.xyz_res)
    in_ch = xyz_dim + cfg.cond_dim
    for i in range(9):
        I = in_ch if i == 0 else (W + in_ch if i == 4 else W)
        O = 3 if i == 8 else W
        _linear(sd, seed, f'{p}.mlp.linears.{i}', I, O)
    sd[f'{p}.mlp.linears.8.bias'] = torch.zeros(3)
    if kind == 'sharp':
        sd[f'{p}.mlp.linears.8.weight'] = sd[f'{p}.mlp.linears.8.weight'] * SHARP_RESD_GAIN
        for l, off in ((0, 0), (4, W)):
            w = sd[f'{p}.mlp.linears.{l}.weight'].clone()
            for k in range(cfg.xyz_res):
                w[:, off + 3 + 6 * k: off + 9 + 6 * k] *= SHARP_RESD_DECAY ** max(0, k - SHARP_RESD_BAND0)
            sd[f'{p}.mlp.linears.{l}.weight'] = w
    # signed distance net (net_utils.py:1276-1352), geometric init, weight-normed
    p = 'signed_distance_network'
    sd[p + '._beta'] = torch.tensor(float(cfg.sdf_beta_init_value))
    sd[p + '.embedder.freq_bands'] = freq_bands(cfg.sdf_res)
    dims = [sdf_dim] + [W] * 8 + [1 + cfg.feat_dim]
    for l in range(9):
        out_dim = dims[l + 1] - dims[0] if l + 1 == 4 else dims[l + 1]
        name = f'{p}.mlp.lin{l}'
        if l == 8:
            w = _normal(seed, name + '.w', (out_dim, dims[l]), math.sqrt(math.pi) / math.sqrt(dims[l]), 1e-4)
            # feature rows: small random so feat is not a copy of sdf
            w[1:] = _normal(seed, name + '.wf', (out_dim - 1, dims[l]), 0.0, 1.0 / math.sqrt(dims[l]))
            b = torch.full((out_dim,), SHARP_SDF_BIAS if kind == 'sharp' else -0.5)
            b[1:] = _uniform(seed, name + '.bf', (out_dim - 1,), 0.1)
        elif l == 0:
            w = torch.zeros(out_dim, dims[l])
            w[:, :3] = _normal(seed, name + '.w', (out_dim, 3), 0.0, math.sqrt(2) / math.sqrt(out_dim))
            # a little energy on the encoded inputs so the PE path is exercised
            w[:, 3:] = _normal(seed, name + '.wpe', (out_dim, dims[l] - 3), 0.0, 0.02 / math.sqrt(out_dim)) * band_scale
            b = torch.zeros(out_dim)
        elif l == 4:
            w = _normal(seed, name + '.w', (out_dim, dims[l]), 0.0, math.sqrt(2) / math.sqrt(out_dim))
            w[:, -(dims[0] - 3):] = _normal(seed, name + '.wpe', (out_dim, dims[0] - 3), 0.0, 0.02 / math.sqrt(out_dim)) * band_scale
            b = torch.zeros(out_dim)
        else:
            w = _normal(seed, name + '.w', (out_dim, dims[l]), 0.0, math.sqrt(2) / math.sqrt(out_dim))
            b = torch.zeros(out_dim)
        _weight_norm(sd, seed, name, w, b, jitter=0.02)
    # colour net (base_network.py:132-171), weight-normed
    p = 'render_network'
    sd[p + '.embedder.freq_bands'] = freq_bands(cfg.view_res)
    shapes = [(view_dim + 3 + cfg.feat_dim, W), (W, W), (W, W), (W + cfg.n_bones * 3, W), (W, 3)]
    for i, (I, O) in enumerate(shapes):
        name = f'{p}.l{i}'
        b = 1.0 / math.sqrt(I)
        _weight_norm(sd, seed, name, _uniform(seed, name + '.w', (O, I), b), _uniform(seed, name + '.b', (O,), b))
    if relight:
        for net, out in (('albedo_network', 3), ('roughness_network', 1)):
            Wm = cfg.relight_network_width
            for i in range(cfg.relight_network_depth + 1):
                I = cfg.feat_dim if i == 0 else Wm
                O = out if i == cfg.relight_network_depth else Wm
                _linear(sd, seed, f'{net}.linears.{i}', I, O, kind='kaiming')
        ch = 1 if cfg.achro_light else 3
        eh, ew = cfg.env_h * cfg.envmap_upscale, cfg.env_w * cfg.envmap_upscale
        # learned probe is softplus(param); give it lobes so shading is not flat
        r = _rng(seed, 'global_env_map_')
        base = r.uniform(0, 1, size=(eh, ew, ch)) * cfg.envmap_init_intensity
        yy, xx = np.mgrid[0:eh, 0:ew]
        lobes = ((eh * 0.3, ew * 0.25, 4.0), (eh * 0.45, ew * 0.7, 2.0))
        if env == 'front':      # a key light towards (-0.80, -0.30, -0.52) — left of and above the synthetic camera's axis, on the camera's
            # side of the body — over a dim ambient (softplus(-2.4) = 0.09): ~85 % of the light's power sits in ~20 of the 512 lights
            base = base - 2.5
            lobes = (_env_pixel(FRONT_LIGHT_DIR, eh, ew) + (14.0,), (eh * 0.45, ew * 0.7, 2.0))
        for (cy, cx, amp) in lobes:
            base += amp * np.exp(-(((yy - cy) / 3.0) ** 2 + ((xx - cx) / 4.0) ** 2))[..., None]
        sd['global_env_map_'] = torch.from_numpy(base.astype(np.float32))
        xyz, area = gen_light_xyz(cfg.env_h, cfg.env_w, cfg.env_r)
        sd['light_xyz_'] = xyz
        sd['light_area'] = area
        sd['light_sharp'] = 1.0 / (area / math.pi).sqrt()
        sd['xyz_embedder.freq_bands'] = freq_bands(10)
        sd['view_embedder.freq_bands'] = freq_bands(4)
    return sd


def _rodrigues(rvec: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def make_body(seed: int = 0, posed: bool = True, radius: float = 0.4, skin_noise: float = 2.0, n_bones: int = None, n_verts: int = None,
              split_axis=None, split_offset=None, split_cos: float = 0.0, skin_sharpness: float = 6.0) -> dotdict:
    """SMPL-shaped frame state with the §8b batch keys (leading batch dim 1).
    skin_noise: std of the per-vertex white noise in the skinning logits.  The default (2.0) makes
    neighbouring vertices follow different bones, so the world -> big-pose warp jumps by ~1 cm wherever the nearest
    vertices change and the reference's own sphere trace ends in a limit cycle on ~9 % of the hit rays; 0.0 gives a spatially smooth
    skinning field like a real SMPL body's (the trace converges) — the well-conditioned case of the parity tests.
    split_axis / split_offset: the bones that own the side of the template facing `split_axis` (their skinning plane's normal has a
    positive component along it) are additionally translated by `split_offset` (metres, pose space): the posed body comes apart into
    two pieces decimetres apart — an arm held in front of the trunk —, so that light-visibility rays leaving one piece meet the other
    at distance (sphere_tracing_renderer.py:265-344).  LBS only: the canonical distance field stays one zero set."""
    N_VERTS, N_BONES = (n_verts or globals()['N_VERTS']), (n_bones or globals()['N_BONES'])      # another body model (SMPL: 24 bones)
    i = np.arange(N_VERTS, dtype=np.float64) + 0.5
    phi = np.arccos(1 - 2 * i / N_VERTS)
    theta = math.pi * (1 + 5 ** 0.5) * i
    nrm = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], -1)
    # squash into a capsule-ish blob so the bbox is not a cube
    scale = np.array([0.8, 0.7, 1.1])
    tverts = nrm * radius * scale
    tnorm = nrm / scale
    tnorm /= np.linalg.norm(tnorm, axis=-1, keepdims=True)
    r = _rng(seed, 'body')
    weights = r.standard_normal((N_VERTS, N_BONES)) * 4.0
    # smooth the skinning field a little: weight depends on position through random planes
    planes = r.standard_normal((N_BONES, 3))
    weights = skin_sharpness * (nrm @ planes.T) + (skin_noise / 4.0) * weights
    weights = np.exp(weights - weights.max(-1, keepdims=True))
    weights /= weights.sum(-1, keepdims=True)
    A = np.tile(np.eye(4), (N_BONES, 1, 1))
    big_A = np.tile(np.eye(4), (N_BONES, 1, 1))
    if posed:
        for j in range(N_BONES):
            A[j, :3, :3] = _rodrigues(r.uniform(-1, 1, 3) * 0.3 / math.sqrt(3))
            A[j, :3, 3] = r.uniform(-0.05, 0.05, 3)
            big_A[j, :3, :3] = _rodrigues(r.uniform(-1, 1, 3) * 0.2 / math.sqrt(3))
            big_A[j, :3, 3] = r.uniform(-0.03, 0.03, 3)
    if split_axis is not None:
        ax = np.asarray(split_axis, dtype=np.float64)
        moved = planes @ ax > split_cos * np.linalg.norm(planes, axis=-1) * np.linalg.norm(ax)
        A[moved, :3, 3] += np.asarray(split_offset, dtype=np.float64)
    # posed verts / normals by forward LBS of the T-pose mesh (what the dataset does on CPU)
    Av = np.einsum('nj,jab->nab', weights, A)
    pverts = np.einsum('nab,nb->na', Av[:, :3, :3], tverts) + Av[:, :3, 3]
    pnorm = np.einsum('nab,nb->na', Av[:, :3, :3], tnorm)
    pnorm /= np.linalg.norm(pnorm, axis=-1, keepdims=True)
    if split_axis is not None:
        # a deformation this large is not near-rigid: rotating the template's normals no longer gives the posed surface's.  Vertex
        # normals of the posed mesh instead, as the dataset computes them (base_dataset.py:222-241: face normals weighted by area,
        # summed per vertex), on the convex hull's triangulation of the template
        faces = _template_faces(nrm)
        fn = np.cross(pverts[faces[:, 1]] - pverts[faces[:, 0]], pverts[faces[:, 2]] - pverts[faces[:, 0]])
        pnorm = np.zeros_like(pverts)
        for k in range(3):
            np.add.at(pnorm, faces[:, k], fn)
        pnorm /= np.maximum(np.linalg.norm(pnorm, axis=-1, keepdims=True), 1e-12)
    if posed:
        R = _rodrigues(np.array([0.1, -0.2, 0.15]))
        Th = np.array([[0.03, -0.02, 0.05]])
    else:
        R = np.eye(3)
        Th = np.zeros((1, 3))
    wverts = pverts @ R.T + Th
    margin = 0.05
    wbounds = np.stack([wverts.min(0) - margin, wverts.max(0) + margin])
    poses = r.standard_normal((N_BONES, 3)) * 0.1
    train_poses = r.standard_normal((4, N_BONES * 3)) * 0.1
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None]
    b = dotdict()
    b.R, b.Th = f(R), f(Th)
    b.poses = f(poses)
    b.weights = f(weights)
    b.A, b.big_A = f(A), f(big_A)
    b.pverts, b.pnorm, b.tverts, b.tnorm = f(pverts), f(pnorm), f(tverts), f(tnorm)
    b.wbounds = f(wbounds)
    b.tbounds = f(np.stack([tverts.min(0) - margin, tverts.max(0) + margin]))      # big-pose box (visualiser's Surface type)
    b.train_motion = dotdict(poses=f(train_poses))
    return b


def _template_faces(nrm: np.ndarray) -> np.ndarray:
    """outward-oriented triangulation of the template: the convex hull of its points on the unit sphere"""
    from scipy.spatial import ConvexHull
    faces = ConvexHull(nrm).simplices.astype(np.int64)
    c = np.cross(nrm[faces[:, 1]] - nrm[faces[:, 0]], nrm[faces[:, 2]] - nrm[faces[:, 0]])
    flip = (c * nrm[faces].mean(1)).sum(-1) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    return faces


def make_camera(H: int, W: int, origin=(0.0, 0.0, -2.0), focal_ratio: float = 0.8):
    """pinhole looking down +z; K, R (world->cam), T."""
    K = np.array([[focal_ratio * H, 0, W / 2], [0, focal_ratio * H, H / 2], [0, 0, 1]], dtype=np.float64)
    R = np.eye(3)
    T = -R @ np.asarray(origin, dtype=np.float64).reshape(3, 1)
    return K, R, T


def tilted_cam_R() -> torch.Tensor:
    """(1,3,3) world-to-camera rotation that is NOT aligned with the world axes (visualiser tests: camera-space normals and
    the light-probe inset, whose axes are undefined for a camera looking along the world's up axis)."""
    ax, ay = 0.35, -0.6
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    return torch.from_numpy((Rx @ Ry).astype(np.float32))[None]


def rays_within_bounds(H, W, K, R, T, bounds: np.ndarray):
    """Restates lib/utils/data_utils.py:827-845 (get_rays), :860-875 (get_full_near_far),
    :925-938 (get_rays_within_bounds) in numpy: unit directions, AABB near/far, box mask."""
    ray_o = -(R.T @ T).ravel()
    i, j = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    xy1 = np.stack([j, i, np.ones_like(i)], axis=2)
    pixel_camera = xy1 @ np.linalg.inv(K).T
    pixel_world = (pixel_camera - T.ravel()) @ R
    ray_d = pixel_world - ray_o[None, None]
    ray_d = ray_d / np.linalg.norm(ray_d, axis=2, keepdims=True)
    ray_o = np.broadcast_to(ray_o, ray_d.shape)
    ray_o = ray_o.reshape(-1, 3).astype(np.float32)
    ray_d = ray_d.reshape(-1, 3).astype(np.float32)
    norm_d = np.linalg.norm(ray_d, axis=-1, keepdims=True)
    viewdir = ray_d / norm_d
    viewdir[(viewdir < 1e-5) & (viewdir > -1e-10)] = 1e-5
    viewdir[(viewdir > -1e-5) & (viewdir < 1e-10)] = -1e-5
    tmin = (bounds[:1] - ray_o[:1]) / viewdir
    tmax = (bounds[1:2] - ray_o[:1]) / viewdir
    t1, t2 = np.minimum(tmin, tmax), np.maximum(tmin, tmax)
    near, far = np.max(t1, axis=-1), np.min(t2, axis=-1)
    mask = near < far
    near = (near / norm_d[..., 0])[mask] / norm_d[mask, 0]
    far = (far / norm_d[..., 0])[mask] / norm_d[mask, 0]
    return ray_o[mask], ray_d[mask], near.astype(np.float32), far.astype(np.float32), mask.reshape(H, W)


def make_skeleton(seed: int = 0):
    """N3 inputs (base_dataset.py:308-397): a 52-joint tree (parents in topological order, SMPL-H style), rest joints
    inside the body, per-frame axis-angle poses / Rh / Th, the big-pose transforms, and a triangulation of the template
    (convex hull of the Fibonacci sphere: a closed, consistently oriented mesh)."""
    from scipy.spatial import ConvexHull
    r = _rng(seed, 'skeleton')
    parents = np.zeros(N_BONES, dtype=np.int64)
    parents[0] = -1
    for j in range(1, N_BONES):
        parents[j] = r.integers(max(0, j - 6), j)
    tjoints = r.uniform(-0.25, 0.25, (N_BONES, 3)).astype(np.float32)
    poses = (r.standard_normal((N_BONES, 3)) * 0.25).astype(np.float32)
    big_poses = (r.standard_normal((N_BONES, 3)) * 0.15).astype(np.float32)
    Rh = np.array([0.1, -0.2, 0.15], dtype=np.float32)
    Th = np.array([0.03, -0.02, 0.05], dtype=np.float32)
    b = make_body(seed, posed=False)
    tv = b.tverts[0].numpy().astype(np.float64)
    hull = ConvexHull(tv / np.array([0.8, 0.7, 1.1]))          # on the sphere itself: every vertex is on the hull
    faces = hull.simplices.astype(np.int64)
    # outward orientation (pytorch3d normals follow the winding)
    c = np.cross(tv[faces[:, 1]] - tv[faces[:, 0]], tv[faces[:, 2]] - tv[faces[:, 0]])
    flip = (c * tv[faces].mean(1)).sum(-1) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    return dotdict(parents=parents, tjoints=tjoints, poses=poses, big_poses=big_poses, Rh=Rh, Th=Th, faces=faces,
                   tverts=b.tverts[0].numpy(), weights=b.weights[0].numpy())


def make_batch(H: int, W: int, seed: int = 0, posed: bool = True, n_novel_lights: int = 0,
               crop: int = 0, skin_noise: float = 2.0, cam_dist: float = 2.0, n_bones: int = None, n_verts: int = None,
               split_axis=None, split_offset=None, split_cos: float = 0.0, skin_sharpness: float = 6.0, crop_at=None) -> dotdict:
    """Full §8b batch on CPU. ``crop``>0 keeps only a centred crop x crop window of pixels.  ``cam_dist``: distance of the camera from
    the body's centre (2 m, the body then covers ~8 % of the frame; 0.96 m: ~35 %, a frame-filling subject).
    ``crop_at``: (row, column) of the window's top-left pixel instead of the centred one."""
    b = make_body(seed, posed, skin_noise=skin_noise, n_bones=n_bones, n_verts=n_verts, split_axis=split_axis, split_offset=split_offset, split_cos=split_cos,
                  skin_sharpness=skin_sharpness)
    K, R, T = make_camera(H, W, origin=(0.0, 0.0, -float(cam_dist)))
    ro, rd, near, far, mask = rays_within_bounds(H, W, K, R, T, b.wbounds[0].numpy().astype(np.float64))
    if crop:
        win = np.zeros((H, W), dtype=bool)
        y0, x0 = ((H - crop) // 2, (W - crop) // 2) if crop_at is None else (int(crop_at[0]), int(crop_at[1]))
        win[y0:y0 + crop, x0:x0 + crop] = True
        keep = win[mask]
        ro, rd, near, far = ro[keep], rd[keep], near[keep], far[keep]
        mask = mask & win
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None]
    b.ray_o, b.ray_d, b.near, b.far = f(ro), f(rd), f(near), f(far)
    b.mask_at_box = torch.from_numpy(mask.reshape(1, -1))
    b.meta = dotdict(H=torch.tensor([H]), W=torch.tensor([W]), frame_index=torch.tensor([0]), view_index=torch.tensor([0]))
    b.cam_K, b.cam_R, b.cam_T = f(K.astype(np.float32)), f(R.astype(np.float32)), f(T.astype(np.float32))
    if n_novel_lights:
        b.novel_lights = make_novel_lights(n_novel_lights, seed)
    return b


def sample_rays(batch: dotdict, n_target: int):
    """every stride-th in-box ray of `batch` (in place; rays are independent units, so the sample renders as one chunk and every
    ray gets the pixel it has in the whole frame up to the per-chunk box growth of the shadow rays).  Returns (batch, P, stride)."""
    P = batch.ray_o.shape[1]
    stride = max(1, P // max(1, n_target))
    if stride % 2 == 0:
        stride += 1          # an even stride can divide the image width: every sample would then sit in the same pixel column
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        batch[k] = batch[k][:, ::stride].contiguous()
    return batch, P, stride


def make_novel_lights(n: int, seed: int = 0, env_h: int = 16, env_w: int = 32) -> dotdict:
    lights = dotdict()
    for k in range(n):
        r = _rng(seed, f'light{k}')
        if k == n - 1 and n > 1:  # OLAT-style: one hot * 100 + ambient 0.25
            probe = np.full((env_h, env_w, 3), 0.25, dtype=np.float32)
            probe[4, 13] = 100.0
        else:
            probe = np.exp(r.standard_normal((env_h, env_w, 1)) * 1.0 - 1.0) * (0.5 + r.uniform(0, 1, (1, 1, 3)))
            probe = probe.astype(np.float32)
        lights[f'probe{k:02d}'] = dotdict(probe=torch.from_numpy(probe)[None])
    return lights


def to_device(batch, device):
    """the loader's hand-over (the reference's to_cuda(batch)); small per-frame constants the host needs again (the body's
    bounding box the renderer grows per chunk) keep a host mirror, so the render loop never reads them back from the device"""
    out = dotdict()
    if isinstance(batch.get('wbounds', None), torch.Tensor) and not batch['wbounds'].is_cuda:
        out['wbounds_host'] = batch['wbounds'].clone()
    for k, v in batch.items():
        if k in ('cam_K', 'cam_R', 'cam_T') and isinstance(v, torch.Tensor) and not v.is_cuda:
            out[k + '_host'] = v             # the ground pass generates its full-frame rays from the camera (host doubles)
        if k == 'meta' or k.endswith('_host') or k.startswith('wbounds_host'):
            out.setdefault(k, v)
        elif isinstance(v, torch.Tensor):
            out[k] = v.to(device)
        elif isinstance(v, dict):
            out[k] = to_device(v, device)
        else:
            out[k] = v
    if 'wbounds_host' in out and isinstance(out.get('wbounds', None), torch.Tensor):
        out['wbounds_host_version'] = out['wbounds']._version      # the mirror is valid for THIS state of the device tensor (Renderer._grow_bounds)
    return out
