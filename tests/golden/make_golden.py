#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --mode {ops,anisdf,sphere,relight,novel,...}
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --mode switches      # one child process per variant -> switches.npz

Imports /root/reference's hot-path modules on CPU with a stub import-finder for the third-party
packages the image lacks (SURVEY.md §8c recipe), loads the build-owned synthetic weights
(relightableavatar_amd.synthetic) into the reference nets with load_state_dict, runs the reference
functions on build-owned synthetic inputs and stores inputs-by-seed + reference outputs as small
.npz fixtures next to this script.  The reference never travels: only these fixtures are committed.
One process per mode, because the reference binds cfg values as default arguments at import time
(SURVEY.md §5 "config / flags" gotcha).
"""
import argparse
import json
import os
import sys
import types
import importlib.abc
import importlib.machinery
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np
import torch

ROOTS = ['cv2', 'termcolor', 'pdbr', 'easymocap', 'smplx', 'pytorch3d', 'h5py', 'imageio', 'mcubes', 'trimesh',
         'torch_scatter', 'easyvolcap', 'lpips', 'skimage', 'open3d', 'pyntcloud', 'ujson', 'ruamel', 'kornia']


class _Mod(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        m = MagicMock(name=f'{self.__name__}.{name}')
        setattr(self, name, m)
        return m


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    def create_module(self, spec):
        m = _Mod(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_reference():
    sys.meta_path.insert(0, _Finder())
    import termcolor
    termcolor.colored = lambda x, *a, **k: str(x)
    import pytorch3d.ops

    def knn_points(p1, p2, K=1, return_nn=False, return_sorted=True, **kw):
        # pytorch3d contract: exact squared-L2 K-NN, ascending.  (p-v)^2 form, chunked.
        ds, ids = [], []
        for i in range(0, p1.shape[1], 4096):
            d = ((p1[:, i:i + 4096, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
            dd, ii = d.topk(K, dim=-1, largest=False, sorted=True)
            ds.append(dd), ids.append(ii)
        if not ds:
            return p1.new_zeros(*p1.shape[:2], K), torch.zeros(*p1.shape[:2], K, dtype=torch.long), None
        return torch.cat(ds, 1), torch.cat(ids, 1), None
    pytorch3d.ops.knn_points = knn_points
    torch.cuda.synchronize = lambda *a, **k: None
    sys.path.insert(0, '/root/reference')
    os.chdir('/root/reference')
    from lib.config import cfg
    return cfg


def set_cfg(cfg, mode):
    cfg.n_bones = 52
    cfg.cond_dim = 156
    cfg.xyz_res, cfg.sdf_res, cfg.view_res = 10, 8, 4
    cfg.fix_material = 0
    cfg.vis_rendering_map = True
    cfg.geometry_pretrain = '/nonexistent'
    if mode in ('ops', 'relight', 'novel', 'ground', 'relight_smooth', 'novel_ground', 'fields'):
        cfg.relighting = True
        cfg.n_samples = 3
        cfg.render_chunk_size = 65536
        cfg.network_chunk_size = 65536
        cfg.dist_th = 0.125
        cfg.obj_lvis.dist_th = 0.125
        cfg.achro_light = True
    elif mode == 'sphere':
        cfg.n_samples = 3
        cfg.render_chunk_size = 65536
        cfg.network_chunk_size = 65536
        cfg.dist_th = 0.1
    elif mode in ('anisdf', 'anisdf128', 'fixmat'):
        cfg.n_samples = 128 if mode == 'anisdf128' else 64
        cfg.render_chunk_size = 8192
        cfg.dist_th = 0.1
        if mode == 'fixmat':
            cfg.fix_material = -1      # base_network.py:502: `fix_material >= 0 or always_fix_material` -> train_motion.poses[:, -1]
    if mode in ('novel', 'novel_ground'):
        cfg.vis_novel_light = True
        cfg.test_light = ['main']
    if mode in ('ground', 'novel_ground'):       # N1: ground-plane pass; two ground chunks so that the per-chunk bbox growth shows
        cfg.vis_ground_shading = True
        cfg.ground_normal = [0.0, -1.0, 0.0]
        cfg.ground_origin = [0.0, 0.45, 0.0]
        cfg.render_chunk_size = 384


# The hot path's configuration switches (sphere_tracing_renderer.py:36,275,295-300,516-538,720-757; relight_utils.py:563-566;
# relight_network.py:63-66), one relit frame of the same 10 x 10 pixel window per variant.  Keys with a dot address a sub-node.
SWITCH_VARIANTS = {
    'base': {},
    'no_dfss': {'no_dfss': True},
    'no_claybook': {'no_claybook': True},
    'no_visibility': {'no_visibility': True},
    'local_visibility': {'local_visibility': True},
    'lambert_only': {'lambert_only': True},
    'glossy_only': {'glossy_only': True},
    'linear': {'tonemapping_rendering': False},
    'only_visibility': {'only_visibility': True},
    'vis_lvis_map': {'vis_lvis_map': True},
    'vis_ldot_map': {'vis_ldot_map': True},
    'chromatic': {'achro_light': False},
    'material_params': {'albedo_multiplier': 2.0, 'shading_albedo': 0.5, 'fresnel_f0': 0.04, 'albedo_slope': 0.7, 'albedo_bias': 0.1,
                        'roughness_slope': 0.5, 'roughness_bias': 0.2},
    'trace_params': {'sphere_tracing.shadow_skip_iter': 2, 'sphere_tracing.tan_i_multiplier': 1.5, 'sphere_tracing.offset': 0.03,
                     'sphere_tracing.near_offset': 0.02, 'sphere_tracing.iter': 12,
                     'obj_lvis.iter': 6, 'obj_lvis.offset': 0.02, 'obj_lvis.near_offset': 0.03, 'obj_lvis.relax': 0.1, 'surf_sample_range': 0.01},
    'no_specular_vis': {'vis_specular_map': False, 'bg_brightness': 0.5},
    'no_geodesic_filter': {'use_geodesic_filter': False},
    'env_r': {'env_r': 3.0},        # lights on a sphere of 3 m: directions and the shadow rays' far end change
    'maps_only': {'vis_rendering_map': False, 'vis_specular_map': False},      # render_human's early return (:702-705): no shading at all
    # structural parameters: one material sample per hit (zval = 0.5, :608-609), five; light sets of other sizes (45 lights: not a multiple
    # of a wavefront; the learned map at 1x / 3x the probe's size); a single shadow iteration from iteration 0
    'one_sample': {'n_samples': 1},
    'five_samples': {'n_samples': 5, 'surf_sample_range': 0.02},
    'small_probe': {'env_h': 8, 'env_w': 16, 'envmap_upscale': 1},
    'odd_probe': {'env_h': 5, 'env_w': 9, 'envmap_upscale': 3},
    'one_shadow_iter': {'obj_lvis.iter': 1, 'sphere_tracing.shadow_skip_iter': 0},
    # another body model: 24 bones (SMPL; cond_dim 72) on a 5 023-vertex mesh — keys starting with @ are arguments of synthetic.make_batch
    'smpl24': {'n_bones': 24, 'cond_dim': 72, '@n_bones': 24, '@n_verts': 5023},
    # other synthetic weights (@weights_seed: synthetic.make_state_dict's seed), another body in its rest pose, another camera distance
    'other_weights': {'@weights_seed': 7, '@seed': 3, '@posed': False, '@cam_dist': 1.6},
    'all_shadowed': {'@weights_seed': 5, '@seed': 3, '@posed': False, '@cam_dist': 1.6},      # this field shadows the whole window: shade = rgb = 0
}
# the hard cases (round 6).  Every variant above renders a convex blob with near-initialisation weights: a shadow ray that leaves it towards a
# front-facing light never meets the body again, and the SDF net's high-frequency encoding columns are ~0.  These do not:
#  * split_body — a bone group (the cap of the template around @split_axis) is pulled 0.57 m out of the body (LBS only: the canonical
#    field stays one zero set): a horn over the shoulder with a gap under it; the key light (@env front: ~2/3 of the power in ~20 lights, on
#    the camera's side) throws its shadow onto the body.  The window lies in that cast shadow and across its penumbra.  obj_lvis.iter = 12:
#    with the default 4 iterations (offset 1 cm) a shadow ray is 10 cm long at most and no part ever shadows another at distance —
#    split_body_iter4 keeps the default for comparison;
#  * sharp_weights — trained-like weights (synthetic.SHARP_BANDS: live encoding columns up to 2^7, centimetre-scale surface detail,
#    |grad sdf| = 1.3 +- 0.5, a 1.5 cm residual deformation) on the base window, with the distance field on 3 000 points around the body;
#  * sharp_split — both, with white noise in the skinning logits (skin_noise 2.0: the world -> big-pose warp jumps between neighbours).
from relightableavatar_amd.synthetic import SPLIT_BODY_KW      # noqa: E402
SPLIT_BODY = dict({'@' + k: v for k, v in SPLIT_BODY_KW.items()}, **{'@env': 'front'})
SPLIT_WINDOW = {'@crop': 12, '@crop_at': [64, 46]}
SWITCH_VARIANTS.update({
    'split_body': dict(SPLIT_BODY, **SPLIT_WINDOW, **{'obj_lvis.iter': 12}),
    'split_body_iter4': dict(SPLIT_BODY, **SPLIT_WINDOW),
    'sharp_weights': {'@weights_kind': 'sharp'},
    'sharp_split': dict(SPLIT_BODY, **SPLIT_WINDOW, **{'obj_lvis.iter': 12, '@weights_kind': 'sharp', '@skin_noise': 2.0}),
})
# the same for the ground-plane pass (render_ground :463-548 + blend_output_): names start with g_, the frame is frame_ground.npz's
# (24 x 24, 10 x 10 window, two ground chunks) on the smooth body
GROUND_BASE = {'vis_ground_shading': True, 'ground_normal': [0.0, -1.0, 0.0], 'ground_origin': [0.0, 0.45, 0.0], 'render_chunk_size': 384}
SWITCH_VARIANTS.update({
    'g_base': {},
    'g_no_dfss': {'no_dfss': True},
    'g_vis_lvis_map': {'vis_lvis_map': True},
    'g_vis_ldot_map': {'vis_ldot_map': True},
    'g_linear': {'tonemapping_rendering': False},
    'g_local_visibility': {'local_visibility': True},
    'g_plain_ground': {'ground_attach_envmap': False, 'ground_albedo': [0.3, 0.2, 0.1], 'ground_shading_multiplier': 2.0},
    'g_env_r': {'env_r': 1.5},      # the light sphere's radius: the ground's distance fade (:497-505) and depth clip (:540) become active inside the frame
    'g_only_visibility': {'only_visibility': True},      # one-channel shade / spec maps of the ground blended against the human layer's (:516-519)
    'g_split_body': dict(SPLIT_BODY),      # the horn's and the body's shadows on the ground, metres long under the low key light
    'g_env_lvis': {'env_lvis.iter': 8, 'env_lvis.offset': 0.02, 'env_lvis.dist_th': 0.01, 'env_lvis.bbox_margin': 0.3, 'env_lvis.near_offset': 0.03},
})
# and for the volume path (base_renderer.py:17,72,120-121): names start with v_, AniSDF network, an 8 x 8 window, 64 samples unless overridden
SWITCH_VARIANTS.update({
    'v_bg': {'bg_brightness': 0.5},
    'v_clip': {'clip_near': 1.7, 'clip_far': 2.4},
    'v_s16_chunks': {'n_samples': 16, 'render_chunk_size': 24},
    'v_sharp_weights': {'@weights_kind': 'sharp'},      # round 6: the trained-like weights through the volume path (every sample a full query: normals + colour net)
})
# ... and through the sphere-tracing fast path of the AniSDF network (config 3: surface trace + full query + colour net, no relighting): s_
SWITCH_VARIANTS.update({
    's_sharp_weights': {'@weights_kind': 'sharp'},
})
VOLUME_H, VOLUME_CROP = 128, 8
# and the novel-light renderer's rotating-light sequence (novel_light_sphere_tracing.py:163-171, relight_utils.py:55-110): names start with
# n_; one probe with a full-resolution image, rotate_ratio * env_w headings re-shaded from one traced frame, four of them stored
SWITCH_VARIANTS.update({
    'n_rotate': {'vis_novel_light': True, 'vis_rotate_light': True, 'rotate_ratio': 1, 'test_light': ['main']},
    'n_only_visibility': {'vis_novel_light': True, 'only_visibility': True, 'test_light': ['main']},      # the cached cosines become 1 (:720-722, :758-759)
    'n_rotate_ground': dict(GROUND_BASE, **{'vis_novel_light': True, 'vis_rotate_light': True, 'rotate_ratio': 2, 'test_light': []}),
    # round 6: the hard-case body through the novel-light renderer — traced once under the learned map, re-shaded under a lognormal probe and
    # an OLAT-style one (a single light of 100 over an ambient 0.25): the key lights of ALL the frame's probes matter for the ONE trace
    'n_split_body': dict(SPLIT_BODY, **SPLIT_WINDOW, **{'vis_novel_light': True, 'test_light': ['main'], 'obj_lvis.iter': 12, '@n_novel_lights': 2}),
})
NOVEL_H, NOVEL_CROP, NOVEL_GROUND_H = 128, 6, 16
NOVEL_HEADINGS = (0, 5, 16, 31)


def novel_light_with_image(synthetic, env_image_w):
    """one synthetic probe + a full-resolution image of it (the dataset delivers both; rotate_envmap shifts both)"""
    lights = synthetic.make_novel_lights(1, 0)
    g = torch.Generator().manual_seed(5)
    lights['probe00'].image = torch.rand(1, env_image_w // 2, env_image_w, 3, generator=g) * 2.0
    return lights
for _k in [k for k in SWITCH_VARIANTS if k.startswith('g_')]:
    SWITCH_VARIANTS[_k] = dict(GROUND_BASE, **SWITCH_VARIANTS[_k])
SWITCH_H, SWITCH_CROP = 128, 10
GROUND_H, GROUND_CROP = 24, 10
SWITCH_KEYS = ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'surf_map', 'albedo_map', 'roughness_map', 'shade_map', 'spec_map')


def apply_overrides(cfg, overrides):
    for k, v in overrides.items():
        if k.startswith('@'):
            continue
        node = cfg
        parts = k.split('.')
        for q in parts[:-1]:
            node = node[q]
        node[parts[-1]] = v


def to_ref_batch(b):
    from lib.utils.base_utils import dotdict
    out = dotdict()
    for k, v in b.items():
        out[k] = to_ref_batch(v) if isinstance(v, dict) else v
    return out


def npz(path, **kw):
    arrs = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        arrs[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, path), **arrs)
    print('wrote', path, {k: a.shape for k, a in arrs.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', required=True, choices=['ops', 'anisdf', 'sphere', 'relight', 'novel', 'rays', 'ground', 'envmap', 'lbs',
                                                   'relight_smooth', 'novel_ground', 'anisdf128', 'fields', 'fixmat', 'visual', 'switches', 'switch'])
    ap.add_argument('--variant', default='base')
    ap.add_argument('--out', default='')
    ap.add_argument('--only', default='', help='--mode switches: regenerate only these variants (comma separated) inside the existing switches.npz')
    args = ap.parse_args()
    mode = args.mode
    if mode == 'switches':          # the reference binds cfg values as default arguments at import: one child process per variant
        import subprocess
        import tempfile
        import json
        merged = dict(H=np.asarray(SWITCH_H), crop=np.asarray(SWITCH_CROP), ground_H=np.asarray(GROUND_H), ground_crop=np.asarray(GROUND_CROP),
                      volume_H=np.asarray(VOLUME_H), volume_crop=np.asarray(VOLUME_CROP), novel_H=np.asarray(NOVEL_H), novel_crop=np.asarray(NOVEL_CROP),
                      novel_ground_H=np.asarray(NOVEL_GROUND_H), novel_headings=np.asarray(NOVEL_HEADINGS),
                      variants_json=np.asarray(json.dumps(SWITCH_VARIANTS)))
        only = [n for n in args.only.split(',') if n]
        if only:          # keep every other variant's arrays as they are in the committed file
            with np.load(os.path.join(HERE, 'switches.npz')) as z:
                for k in z.files:
                    if k != 'variants_json' and k.split('.')[0] not in only:
                        merged.setdefault(k, z[k])
        with tempfile.TemporaryDirectory() as tmp:
            for name in (only or SWITCH_VARIANTS):
                out = os.path.join(tmp, name + '.npz')
                subprocess.run([sys.executable, os.path.abspath(__file__), '--mode', 'switch', '--variant', name, '--out', out], check=True)
                with np.load(out) as z:
                    for k in z.files:
                        merged[f'{name}.{k}'] = z[k]
        np.savez_compressed(os.path.join(HERE, 'switches.npz'), **merged)
        print('wrote switches.npz', len(merged), 'arrays')
        return
    from relightableavatar_amd import synthetic
    from relightableavatar_amd.config import make_cfg
    cfg = install_reference()
    if mode == 'rays':
        gen_rays(synthetic)
        return
    if mode == 'envmap':
        gen_envmap(cfg, synthetic)
        return
    if mode == 'lbs':
        gen_lbs(synthetic)
        return
    if mode == 'visual':
        gen_visual(cfg, synthetic)
        return
    set_cfg(cfg, ('anisdf' if args.variant.startswith('v_') else ('sphere' if args.variant.startswith('s_') else 'relight')) if mode == 'switch' else mode)
    if mode == 'switch':
        cfg.vis_specular_map = True
        apply_overrides(cfg, SWITCH_VARIANTS[args.variant])
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    if mode == 'switch':
        gen_switch(cfg, synthetic, args.variant, args.out)
        return
    my_cfg = make_cfg({'ops': 'relight', 'anisdf': 'anisdf', 'sphere': 'sphere_tracing', 'relight': 'relight', 'novel': 'novel_light', 'ground': 'relight',
                       'relight_smooth': 'relight', 'novel_ground': 'novel_light', 'anisdf128': 'anisdf', 'fields': 'relight', 'fixmat': 'anisdf'}[mode])
    relight = mode in ('ops', 'relight', 'novel', 'ground', 'relight_smooth', 'novel_ground', 'fields')
    sd = synthetic.make_state_dict(0, relight=relight, cfg=my_cfg)
    if relight:
        from lib.networks.relight.relight_network import Network
    else:
        from lib.networks.deform.base_network import Network
    net = Network()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if 'embedder' not in m], missing
    net.eval()

    if mode == 'ops':
        gen_ops(net, cfg, synthetic)
        return
    if mode == 'fields':
        gen_fields(net, cfg, synthetic)
        return
    if mode == 'fixmat':
        gen_fixmat(net, cfg, synthetic)
        return
    from lib.networks.renderer import base_renderer, sphere_tracing_renderer
    if mode == 'anisdf':
        H, crop = 128, 24
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop))
        with torch.no_grad():
            out = base_renderer.Renderer(net).render(batch)
        npz('frame_anisdf.npz', H=H, crop=crop, n_samples=cfg.n_samples,
            **{k: out[k] for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'cpts_map', 'bpts_map', 'resd_map')})
    elif mode == 'anisdf128':      # config 2's sample count (base.yaml:78) on a small crop: the depth-major sample layout is S-dependent
        H, crop = 128, 12
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop))
        with torch.no_grad():
            out = base_renderer.Renderer(net).render(batch)
        npz('frame_anisdf128.npz', H=H, crop=crop, n_samples=cfg.n_samples,
            **{k: out[k] for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'cpts_map', 'bpts_map', 'resd_map')})
    elif mode == 'relight_smooth':  # the well-conditioned case: spatially smooth skinning field (the reference's sphere trace converges)
        H, crop = 128, 16
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop, skin_noise=0.0))
        cfg.vis_specular_map = True
        with torch.no_grad():
            out = sphere_tracing_renderer.Renderer(net).render(batch)
        npz('frame_relight_smooth.npz', H=H, crop=crop, skin_noise=0.0, wbounds_after=batch.wbounds,
            **{k: out[k] for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'surf_map', 'albedo_map', 'roughness_map',
                                   'shade_map', 'spec_map', 'cpts_map', 'bpts_map', 'resd_map',
                                   # render_human's per-hit leftovers (:616-650), in the reference's own (topk, unsorted) hit order
                                   'raw', 'volume_albedo', 'volume_roughness')})
    elif mode == 'novel_ground':    # the README's relight command (readme.md:64): vis_novel_light + vis_ground_shading, main + probes
        from lib.networks.renderer import novel_light_sphere_tracing
        H, crop = 24, 10
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop, n_novel_lights=2, skin_noise=0.0))
        m = batch.mask_at_box.reshape(1, -1)
        with torch.no_grad():
            out = novel_light_sphere_tracing.Renderer(net).render(batch)
        kw = dict(H=H, crop=crop, skin_noise=0.0, render_chunk_size=cfg.render_chunk_size, ground_normal=cfg.ground_normal,
                  ground_origin=cfg.ground_origin, wbounds_after=batch.wbounds)
        for name in out:
            if name == 'diff':
                continue
            for k in ('rgb_map', 'shade_map', 'spec_map', 'acc_map', 'albedo_map', 'norm_map', 'surf_map', 'depth_map', 'roughness_map'):
                if k in out[name]:
                    kw[f'{name}.{k}'] = out[name][k]
        npz('frame_novel_ground.npz', **kw)
    elif mode == 'sphere':
        H, crop = 128, 32
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop))
        with torch.no_grad():
            out = sphere_tracing_renderer.Renderer(net).render(batch)
        npz('frame_sphere.npz', H=H, crop=crop,
            **{k: out[k] for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'surf_map', 'cpts_map', 'bpts_map', 'resd_map')})
    elif mode == 'relight':
        H, crop = 128, 16
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop))
        cfg.vis_specular_map = True
        with torch.no_grad():
            out = sphere_tracing_renderer.Renderer(net).render(batch)
        npz('frame_relight.npz', H=H, crop=crop, wbounds_after=batch.wbounds,
            **{k: out[k] for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'surf_map', 'albedo_map', 'roughness_map',
                                   'shade_map', 'spec_map', 'cpts_map', 'bpts_map', 'resd_map')})
    elif mode == 'ground':
        H, crop = 24, 10
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop))
        with torch.no_grad():
            out = sphere_tracing_renderer.Renderer(net).render(batch)
        npz('frame_ground.npz', H=H, crop=crop, render_chunk_size=cfg.render_chunk_size, ground_normal=cfg.ground_normal,
            ground_origin=cfg.ground_origin, wbounds_after=batch.wbounds,
            **{k: out[k] for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'surf_map', 'albedo_map', 'roughness_map',
                                   'shade_map', 'spec_map', 'cpts_map', 'bpts_map')})
    elif mode == 'novel':
        from lib.networks.renderer import novel_light_sphere_tracing
        H, crop = 128, 12
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop, n_novel_lights=3))
        with torch.no_grad():
            out = novel_light_sphere_tracing.Renderer(net).render(batch)
        kw = dict(H=H, crop=crop)
        for name in out:
            if name == 'diff':
                continue
            for k in ('rgb_map', 'shade_map', 'spec_map', 'acc_map', 'lvis_map', 'ldot_map', 'albedo_map', 'norm_map'):
                if k in out[name]:
                    kw[f'{name}.{k}'] = out[name][k]
        npz('frame_novel.npz', **kw)


def gen_switch(cfg, synthetic, variant, out_path):
    """one relit frame of the reference under SWITCH_VARIANTS[variant] (cfg already carries the overrides; nothing of the reference's
    hot path is imported yet, so values bound as default arguments see them too)"""
    from relightableavatar_amd.config import make_cfg
    if variant == 'n_split_body':
        from lib.networks.renderer import novel_light_sphere_tracing
        from lib.networks.relight.relight_network import Network
        my_cfg = make_cfg('novel_light')
        bkw = {k[1:]: v for k, v in SWITCH_VARIANTS[variant].items() if k.startswith('@')}
        net = Network()
        missing, unexpected = net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=my_cfg, env=bkw.pop('env', 'back')), strict=False)
        assert not unexpected and not [m for m in missing if 'embedder' not in m], (missing, unexpected)
        net.eval()
        batch = to_ref_batch(synthetic.make_batch(SWITCH_H, SWITCH_H, **{**dict(seed=0, posed=True, crop=SWITCH_CROP, skin_noise=0.0), **bkw}))
        with torch.no_grad():
            out = novel_light_sphere_tracing.Renderer(net).render(batch)
        names = [k for k in out if k != 'diff']
        arrs = {'names': np.asarray(json.dumps(names))}
        for name in names:
            for k in ('rgb_map', 'shade_map', 'spec_map', 'albedo_map', 'acc_map', 'norm_map', 'surf_map'):
                if k in out[name]:
                    arrs[f'{name}/{k}'] = out[name][k].detach().cpu().numpy()
        np.savez_compressed(out_path, **arrs)
        print('switch', variant, names, {k: a.shape for k, a in arrs.items() if k.startswith('probe01')})
        return
    if variant.startswith('n_'):
        from lib.networks.renderer import novel_light_sphere_tracing
        from lib.networks.relight.relight_network import Network
        my_cfg = make_cfg('novel_light')
        cfg.env_image_w = 64
        net = Network()
        missing, unexpected = net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=my_cfg), strict=False)
        assert not unexpected and not [m for m in missing if 'embedder' not in m], (missing, unexpected)
        net.eval()
        ground = 'ground' in variant
        H = NOVEL_GROUND_H if ground else NOVEL_H
        batch = to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True, crop=NOVEL_CROP, skin_noise=0.0))
        batch.novel_lights = to_ref_batch(novel_light_with_image(synthetic, 64))
        with torch.no_grad():
            out = novel_light_sphere_tracing.Renderer(net).render(batch)
        names = [k for k in out if k != 'diff']
        arrs = {'names': np.asarray(json.dumps(names))}
        keep = (['main'] if 'main' in out else []) + [f'probe00-{j:04d}' for j in NOVEL_HEADINGS] if cfg.vis_rotate_light else names
        for name in keep:
            for k in ('rgb_map', 'shade_map', 'spec_map', 'albedo_map', 'acc_map'):
                if k in out[name]:
                    arrs[f'{name}/{k}'] = out[name][k].detach().cpu().numpy()
            arrs[f'{name}/probe'] = out[name].envmap.probe.detach().cpu().numpy()
        np.savez_compressed(out_path, **arrs)
        print('switch', variant, len(names), 'outputs;', {k: a.shape for k, a in arrs.items() if k.startswith('probe00-0005')})
        return
    if variant.startswith('s_'):
        my_cfg = make_cfg('sphere_tracing')
        sd = synthetic.make_state_dict(0, relight=False, cfg=my_cfg, kind=SWITCH_VARIANTS[variant].get('@weights_kind', 'init'))
        from lib.networks.deform.base_network import Network
        from lib.networks.renderer import sphere_tracing_renderer
        net = Network()
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected and not [m for m in missing if 'embedder' not in m], (missing, unexpected)
        net.eval()
        batch = to_ref_batch(synthetic.make_batch(SWITCH_H, SWITCH_H, seed=0, posed=True, crop=SWITCH_CROP, skin_noise=0.0))
        with torch.no_grad():
            out = sphere_tracing_renderer.Renderer(net).render(batch)
        arrs = {k: out[k].detach().cpu().numpy() for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'surf_map', 'cpts_map', 'resd_map')}
        np.savez_compressed(out_path, **arrs)
        print('switch', variant, {k: a.shape for k, a in arrs.items()})
        return
    if variant.startswith('v_'):
        my_cfg = make_cfg('anisdf', n_samples=64)
        sd = synthetic.make_state_dict(0, relight=False, cfg=my_cfg, kind=SWITCH_VARIANTS[variant].get('@weights_kind', 'init'))
        from lib.networks.deform.base_network import Network
        from lib.networks.renderer import base_renderer
        net = Network()
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected and not [m for m in missing if 'embedder' not in m], (missing, unexpected)
        net.eval()
        batch = to_ref_batch(synthetic.make_batch(VOLUME_H, VOLUME_H, seed=0, posed=True, crop=VOLUME_CROP, skin_noise=0.0))
        with torch.no_grad():
            out = base_renderer.Renderer(net).render(batch)
        arrs = {k: out[k].detach().cpu().numpy() for k in ('rgb_map', 'acc_map', 'depth_map', 'norm_map', 'cpts_map', 'resd_map')}
        np.savez_compressed(out_path, **arrs)
        print('switch', variant, {k: a.shape for k, a in arrs.items()})
        return
    my_cfg = make_cfg('relight')
    batch_kw = {k[1:]: v for k, v in SWITCH_VARIANTS[variant].items() if k.startswith('@')}
    for k, v in SWITCH_VARIANTS[variant].items():
        if k.startswith('@'):
            continue
        node = my_cfg
        parts = k.split('.')
        for q in parts[:-1]:
            node = node[q]
        node[parts[-1]] = v
    sd = synthetic.make_state_dict(batch_kw.pop('weights_seed', 0), relight=True, cfg=my_cfg, kind=batch_kw.pop('weights_kind', 'init'),
                                   env=batch_kw.pop('env', 'back'))
    from lib.networks.relight.relight_network import Network
    from lib.networks.renderer import sphere_tracing_renderer
    net = Network()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if 'embedder' not in m], missing
    net.eval()
    H, crop = (GROUND_H, GROUND_CROP) if variant.startswith('g_') else (SWITCH_H, SWITCH_CROP)
    batch = to_ref_batch(synthetic.make_batch(H, H, **{**dict(seed=0, posed=True, crop=crop, skin_noise=0.0), **batch_kw}))
    with torch.no_grad():
        out = sphere_tracing_renderer.Renderer(net).render(batch)
    arrs = {k: out[k].detach().cpu().numpy() for k in SWITCH_KEYS if k in out}
    if variant.startswith('g_'):
        arrs['wbounds_after'] = batch.wbounds.numpy()
    if variant in ('base', 'no_geodesic_filter', 'sharp_weights'):
        # the distance field itself on points all around the body (where the neighbour rule matters: between the arms and the trunk,
        # between the legs), the frame's window being a patch of the chest
        body = to_ref_batch(synthetic.make_body(0, posed=True, skin_noise=0.0))
        g = torch.Generator().manual_seed(91)
        wb = body.wbounds[0]
        x = wb[0] + (wb[1] - wb[0]) * torch.rand(3000, 3, generator=g)
        with torch.no_grad():
            arrs['hdq_x'] = x.numpy()
            arrs['hdq_sdf'] = net.inference_world_distance_field(x[None], body, smooth_transition=True, dist_th=cfg.dist_th)[0].numpy()
            arrs['hdq_sdf_coarse'] = net.world_to_bigpose(x[None], None, body, dist_th=cfg.dist_th).sdf_batch.mean(dim=-1)[0].numpy()
    np.savez_compressed(out_path, **arrs)
    print('switch', variant, {k: a.shape for k, a in arrs.items()})


def gen_fields(net, cfg, synthetic):
    """The Network methods the reference renderer binds for its ablation modes (sphere_tracing_renderer.py:955-961):
    inference_observed_distance_field (base_network.py:389-449, plain and filtered) and world_to_bigpose_transform /
    bigpose_to_world_transform (:338-363).  The two transform methods concatenate batch.Th[..., None] to batch.R, which only
    works for Th of shape (B,3) — with the dataset's (B,1,3) they raise — so they are called with Th squeezed."""
    g = torch.Generator().manual_seed(77)
    batch = to_ref_batch(synthetic.make_body(0, posed=True))
    bp = torch.nn.functional.normalize(torch.randn(400, 3, generator=g), dim=-1) * (0.3 + 0.25 * torch.rand(400, 1, generator=g)) * torch.tensor([0.8, 0.7, 1.1])
    wb = batch.wbounds[0]
    xw = wb[0] + (wb[1] - wb[0]) * torch.rand(300, 3, generator=g)
    with torch.no_grad():
        obs = net.inference_observed_distance_field(bp[None], batch)
        obs_f = net.inference_observed_distance_field(bp[None], batch, smooth_transition=True, filtering=True, dist_th=0.125)
        obs_fn = net.inference_observed_distance_field(bp[None], batch, smooth_transition=False, filtering=True, dist_th=0.125)
        b2 = to_ref_batch(synthetic.make_body(0, posed=True))
        b2.Th = b2.Th[:, 0]
        w2b = net.world_to_bigpose_transform(xw[None], b2)
        b2w = net.bigpose_to_world_transform(bp[None], b2)
        # both return their rows in the COMPACTION order of geodesic_knn (batch_aware_indexing = topk(sorted=False),
        # implementation-defined) and never scatter them back: store that order next to the rows
        from lib.utils.sample_utils import geodesic_knn
        from lib.utils.blend_utils import world_points_to_pose_points
        w2b_inds = geodesic_knn(world_points_to_pose_points(xw[None], b2.R, b2.Th), b2.pverts, b2.pnorm, b2.tverts, b2.tnorm, 3, 1e9)[2]
        b2w_inds = geodesic_knn(bp[None], b2.tverts, b2.tnorm, b2.tverts, b2.tnorm, 3, 1e9)[2]
    npz('fields.npz', obs_x=bp, obs_sdf=obs[0], obs_sdf_filtered=obs_f[0], obs_sdf_filtered_nosmooth=obs_fn[0], w2b_x=xw, w2b=w2b[0], b2w=b2w[0],
        w2b_inds=w2b_inds[0], b2w_inds=b2w_inds[0])


def gen_fixmat(net, cfg, synthetic):
    """AniSDF Network.forward with cfg.fix_material = -1: the colour net is conditioned on train_motion.poses[:, -1]
    (base_network.py:501-503), geometry on the current pose."""
    g = torch.Generator().manual_seed(5)
    batch = to_ref_batch(synthetic.make_body(0, posed=True))
    vid = torch.randint(0, 6890, (200,), generator=g)
    xw = (batch.pverts[0] @ batch.R[0].T + batch.Th[0])[vid] + 0.02 * torch.randn(200, 3, generator=g)
    v = torch.nn.functional.normalize(torch.randn(200, 3, generator=g), dim=-1)
    out = net(xw[None], v[None], 0.005, batch)
    npz('fixmat.npz', x=xw, v=v, raw=out.raw[0].detach(), fix_material=cfg.fix_material)


def gen_visual(cfg, synthetic):
    """N4, third item: Visualizer.generate_image (lib/visualizers/base_visualizer.py:54-231) for every output type of the hot
    path, fed with the maps of the reference's own relit frame (frame_relight_smooth.npz) — the images are pure functions of
    those maps, so the fixture only stores the images."""
    from lib.utils.base_utils import dotdict
    from lib.config.config import Output
    from lib.utils import relight_utils
    from lib.visualizers.base_visualizer import Visualizer
    ref = dict(np.load(os.path.join(HERE, 'frame_relight_smooth.npz')))
    H, crop = int(ref['H']), int(ref['crop'])
    b = synthetic.make_batch(H, H, seed=0, posed=True, crop=crop, skin_noise=float(ref['skin_noise']))
    b.cam_R = synthetic.tilted_cam_R()          # the synthetic camera looks along the world's z ("up" for gen_light_dir): degenerate probe axes
    batch = to_ref_batch(b)
    from relightableavatar_amd.config import make_cfg
    sd = synthetic.make_state_dict(0, relight=True, cfg=make_cfg('relight'))
    probe = torch.nn.functional.softplus(sd['global_env_map_'].expand(-1, -1, 3))[None]
    out = dotdict({k: torch.from_numpy(v) for k, v in ref.items() if k.endswith('_map')})
    out.envmap = dotdict(probe=probe)
    cfg.env_h, cfg.env_w = 16, 32
    _glx = relight_utils.gen_light_xyz           # its device argument defaults to 'cuda'
    relight_utils.gen_light_xyz = lambda h, w, r=1e2, device='cpu': _glx(h, w, r, device='cpu')
    kw = {}
    for t in (Output.Surface, Output.Residual, Output.Depth, Output.Alpha, Output.Normal, Output.Specular, Output.Albedo, Output.Roughness,
              Output.Shading, Output.Rendering):
        kw[f'img_{t.name}'] = Visualizer.generate_image(dotdict(out), batch, t)
    cfg.normalize_shading, cfg.store_alpha_channel, cfg.probe_size_ratio, cfg.tonemapping_albedo = True, False, 0.0, False
    for t in (Output.Shading, Output.Albedo, Output.Rendering):
        kw[f'alt_{t.name}'] = Visualizer.generate_image(dotdict(out), batch, t)
    kw['img_Envmap'] = Visualizer.generate_image(dotdict(out), batch, Output.Envmap)
    # the frame's camera has a d_x = 0 pixel column, whose depth is 0/0 (quirk 4) and makes the percentile NaN: a second Depth
    # image from the same map with the non-finite entries replaced by 1.7 exercises the stretch itself
    o2 = dotdict(out)
    o2.depth_map = torch.where(torch.isfinite(out.depth_map), out.depth_map, torch.full_like(out.depth_map, 1.7))
    kw['alt_Depth_finite'] = Visualizer.generate_image(o2, batch, Output.Depth)
    npz('visual.npz', **kw)


def gen_rays(synthetic):
    """N2: the reference's per-frame ray set-up (lib/utils/data_utils.py:925-938) on two cameras."""
    from lib.utils.data_utils import get_rays_within_bounds
    kw = {}
    b = synthetic.make_body(0, True)
    bounds = b.wbounds[0].numpy().astype(np.float32)          # get_bounds() returns float32 (base_dataset.py)
    # (a) the synthetic frontal camera (float64, as relightableavatar_amd.synthetic builds it)
    H = W = 96
    K, R, T = synthetic.make_camera(H, W)
    # (b) a rotated float32 camera, H != W (novel-view style intrinsics, pose_dataset.py:57-64)
    H2, W2 = 64, 80
    K2 = np.array([[H2 * 0.8, 0, W2 / 2], [0, H2 * 0.8, H2 / 2], [0, 0, 1]], dtype=np.float32)
    ax, ay = 0.35, -0.6
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    R2 = (Rx @ Ry).astype(np.float32)
    T2 = np.array([[0.1], [-0.05], [2.2]], dtype=np.float32)
    for tag, (h, w, k, r, t) in {'a': (H, W, K, R, T), 'b': (H2, W2, K2, R2, T2)}.items():
        ro, rd, near, far, mask = get_rays_within_bounds(h, w, k, r, t, bounds)
        kw.update({f'{tag}_H': h, f'{tag}_W': w, f'{tag}_K': np.asarray(k, np.float64), f'{tag}_R': np.asarray(r, np.float64),
                   f'{tag}_T': np.asarray(t, np.float64).reshape(3), f'{tag}_ray_o': ro, f'{tag}_ray_d': rd, f'{tag}_near': near,
                   f'{tag}_far': far, f'{tag}_mask': mask})
    npz('rays.npz', bounds=bounds, **kw)


def gen_lbs(synthetic):
    """N3: the reference's own per-frame SMPL-state functions on the synthetic skeleton (every 5th vertex kept to stay small):
    get_rigid_transformation_and_joints (data_utils.py:1026-1069), pose_points_to_tpose_points / tpose_points_to_pose_points /
    pose_points_to_world_points (blend_utils.py:264-313), get_bounds (data_utils.py:616-622)."""
    from lib.utils import data_utils, blend_utils
    sk = synthetic.make_skeleton(0)
    A, J = data_utils.get_rigid_transformation_and_joints(sk.poses, sk.tjoints, sk.parents)
    big_A, _ = data_utils.get_rigid_transformation_and_joints(sk.big_poses, sk.tjoints, sk.parents)
    sel = np.arange(0, sk.tverts.shape[0], 5)
    tv, w = torch.from_numpy(sk.tverts[sel])[None], torch.from_numpy(sk.weights[sel])[None]
    txyz = blend_utils.pose_points_to_tpose_points(tv, w, torch.from_numpy(big_A)[None])
    pxyz = blend_utils.tpose_points_to_pose_points(txyz, w, torch.from_numpy(A)[None])
    R = synthetic._rodrigues(sk.Rh.astype(np.float64)).astype(np.float32)
    wxyz = blend_utils.pose_points_to_world_points(pxyz, torch.from_numpy(R)[None], torch.from_numpy(sk.Th)[None])
    npz('lbs.npz', sel=sel, A=A, joints=J, big_A=big_A, txyz=txyz[0], pxyz=pxyz[0], wxyz=wxyz[0], R=R,
        pbounds=data_utils.get_bounds(pxyz[0].numpy().copy()), wbounds=data_utils.get_bounds(wxyz[0].numpy().copy()))


def gen_envmap(cfg, synthetic):
    """N4: rotate_envmap (probe + image) and add_light_probe of the reference (lib/utils/relight_utils.py:38-103)."""
    from lib.utils.base_utils import dotdict
    from lib.utils import relight_utils
    lights = synthetic.make_novel_lights(3, 0)
    g = torch.Generator().manual_seed(3)
    nl = dotdict()
    for k, v in lights.items():
        nl[k] = dotdict(probe=v.probe, image=torch.rand(1, 24, 48, 3, generator=g))
    kw = {}
    repeat = 4
    for index in (0, 5, 37, 128 + 77, 2 * 128 + 127):
        name, env = relight_utils.rotate_envmap(nl, index, repeat, 32, 48)
        kw[f'rot{index}_probe'] = env.probe[0]
        kw[f'rot{index}_image'] = env.image[0]
        kw[f'rot{index}_name'] = np.array(name)
    K, R, T = synthetic.make_camera(96, 96)
    ax, ay = 0.35, -0.6
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    cam_R = torch.from_numpy((Rx @ Ry).astype(np.float32))[None]
    H, W = 40, 64
    rgb = torch.rand(1, H * W, 3, generator=g)
    batch = dotdict(meta=dotdict(H=torch.tensor([H]), W=torch.tensor([W])), cam_R=cam_R)
    cfg.env_h, cfg.env_w, cfg.probe_size_ratio = 16, 32, 0.2
    _glx = relight_utils.gen_light_xyz           # its device argument defaults to 'cuda'
    relight_utils.gen_light_xyz = lambda h, w, r=1e2, device='cpu': _glx(h, w, r, device='cpu')
    out = relight_utils.add_light_probe(rgb.clone(), nl['probe00'].probe, batch, cfg)
    npz('envmap.npz', repeat=repeat, images=torch.stack([nl[k].image[0] for k in nl]), cam_R=cam_R[0], H=H, W=W, rgb_in=rgb[0],
        rgb_out=out[0], **kw)


def gen_ops(net, cfg, synthetic):
    """Stage-level goldens (relight network = superset of AniSDF stages)."""
    from lib.utils.base_utils import dotdict
    from lib.utils import relight_utils, net_utils, sample_utils
    from lib.networks import embedder
    from lib.networks.renderer import sphere_tracing_renderer as st
    g = torch.Generator().manual_seed(123)
    batch = to_ref_batch(synthetic.make_body(0, posed=True))
    # --- A: positional encoding
    x = (torch.rand(64, 3, generator=g) - 0.5) * 2.0
    kw = dict(pe_x=x)
    for L in (10, 8, 4):
        kw[f'pe{L}'] = embedder.PositionalEncoding(L)(x[None])[0]
    # --- B,C: MLPs on big-pose points
    bpts = (torch.rand(256, 3, generator=g) - 0.5) * 1.0
    cond = batch.poses.view(1, -1)
    with torch.no_grad():
        resd = net.residual_deformation_network(bpts[None], cond)[0]
        sdf, feat = net.signed_distance_network.sdf_feat((bpts + resd)[None])
        occ = net_utils.sdf_to_occ(sdf, net.signed_distance_network.beta)
        albedo = net.albedo_network(feat)
        rough = net.roughness_network(feat)
        view = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1)
        nrm = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1)
        cfix = batch.train_motion.poses[:, 0].view(1, 1, -1).expand(1, 256, -1)
        rgb = net.render_network(view[None], nrm[None], feat, cfix)
    kw.update(mlp_bpts=bpts, mlp_resd=resd, mlp_sdf=sdf[0], mlp_feat=feat[0], mlp_occ=occ[0], mlp_albedo=albedo[0],
              mlp_rough=rough[0], col_view=view, col_norm=nrm, col_rgb=rgb[0])
    # --- D,E,F: knn / warp / HDQ on world points around the body
    wb = batch.wbounds[0]
    xw = wb[0] + (wb[1] - wb[0]) * torch.rand(4096, 3, generator=g)
    xw = torch.cat([xw, batch.pverts[0, ::40] @ batch.R[0].T + batch.Th[0] + 0.01 * torch.randn(173, 3, generator=g)])
    ppts = (xw - batch.Th[0]) @ batch.R[0]
    with torch.no_grad():
        sdf_batch, nn_batch, inds, S, d2, nn, _ = sample_utils.geodesic_knn(ppts[None], batch.pverts, batch.pnorm, batch.tverts, batch.tnorm, 3, 0.125)
        ret = net.world_to_bigpose(xw[None], torch.ones_like(xw)[None] * torch.tensor([0.0, 0.6, 0.8]), batch, dist_th=0.125)
        hdq = net.inference_world_distance_field(xw[None], batch, smooth_transition=True, dist_th=0.125)
        hdq_nosmooth = net.inference_world_distance_field(xw[None], batch, smooth_transition=False, dist_th=0.125)
    fine = torch.zeros(xw.shape[0], dtype=torch.bool)
    fine[inds[0]] = True

    def scat(v):  # scatter compacted rows back to the full set so the fixture is order-free
        o = torch.zeros(xw.shape[0], *v.shape[2:], dtype=v.dtype)
        o[ret.inds[0]] = v[0]
        return o
    kw.update(hdq_x=xw, knn_sdf_batch=sdf_batch[0], knn_nn_batch=nn_batch[0], knn_fine=fine,
              warp_bpts=scat(ret.bpts), warp_tpts=scat(ret.tpts), warp_A_bw=scat(ret.A_bw), warp_big_A_bw=scat(ret.big_A_bw),
              warp_bvds=scat(ret.bvds), warp_d2=scat(ret.d2), warp_nn=scat(ret.nn), hdq_sdf=hdq[0], hdq_sdf_nosmooth=hdq_nosmooth[0])
    # --- K: full forward (normals, raw 17)
    xs = xw[fine][:300]
    out = net(xs[None], None, 0.005, batch)
    _, geo = net.forward_geometry(xs[None], None, 0.005, batch)
    kw.update(fwd_x=xs, fwd_raw=out.raw[0].detach(), fwd_inds=geo.inds[0], fwd_norm_c=geo.norm[0].detach(), fwd_sdf_c=geo.sdf[0].detach())
    # --- L: light geometry, envmap sampling, microfacet, srgb
    xyz, area = relight_utils.gen_light_xyz(16, 32, 10, device='cpu')
    dirs = torch.nn.functional.normalize(torch.randn(500, 3, generator=g), dim=-1)
    probe = net.global_env_map.detach()
    kw.update(light_xyz=xyz, light_area=area, light_sharp=net.light_sharp, env_probe=probe, env_dirs=dirs,
              env_sample=relight_utils.sample_envmap_image(probe[None], dirs[None])[0])
    N, L = 40, 512
    p2l = torch.randn(L, N, 3, generator=g)   # (L,P,3) like surf2light
    p2c = torch.randn(N, 3, generator=g)
    nn_ = torch.randn(N, 3, generator=g)
    alb = torch.rand(N, 3, generator=g)
    rgh = torch.rand(N, 1, generator=g) * 0.9 + 0.09
    rgh[:4] = 0.09
    brdf = net.microfacet(p2l.clone().view(1, 16, 32, N, 3), p2c.clone()[None], nn_.clone()[None], alb.clone()[None], rgh.clone()[None]).view(L, N, 3)
    lin = torch.cat([torch.linspace(-0.1, 1.2, 200), torch.tensor([0.0, 0.0031308, 0.0031309, 1.0])])
    kw.update(mf_p2l=p2l, mf_p2c=p2c, mf_n=nn_, mf_albedo=alb, mf_rough=rgh, mf_brdf=brdf, srgb_in=lin, srgb_out=relight_utils.linear2srgb(lin))
    # --- volume rendering + aabb
    raw = torch.rand(50, 5, 7, generator=g)
    al = torch.rand(50, 5, generator=g)
    al[:5] = 0
    w_, m_, a_ = net_utils.volume_rendering(raw[None], al[None])
    ro = (torch.rand(300, 3, generator=g) - 0.5)
    rd = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1)
    rd[:3, 0] = 0.0
    rd[3:6, 1] = 1e-9
    rd[6:9, 2] = -1e-17
    bb = torch.tensor([[[-0.6, -0.5, -0.7], [0.6, 0.5, 0.7]]])
    nr, fr_ = net_utils.get_near_far_aabb(bb, ro[None], rd[None].clone(), return_raw=True)
    kw.update(vr_raw=raw, vr_alpha=al, vr_weights=w_[0], vr_map=m_[0], vr_acc=a_[0], aabb_o=ro, aabb_d=rd, aabb_bounds=bb[0], aabb_near=nr[0], aabb_far=fr_[0])
    # --- G: sphere tracing (surface + shadow) through the real HDQ
    b2 = synthetic.make_batch(128, 128, seed=0, posed=True, crop=20)
    ray_o, ray_d, near, far = b2.ray_o, b2.ray_d, b2.near, b2.far
    dec = lambda x, **k: net.inference_world_distance_field(x, batch, smooth_transition=True, **k)
    with torch.no_grad():
        surf, edge, occ_, st_, ot_ = st.sphere_tracing(ray_o, ray_d, near, far, dec, None, None)
        # shadow-style trace with per-ray tan_i
        tan_i = (9.0 + 20 * torch.rand(1, ray_o.shape[1], 1, generator=g))
        o2 = surf + 0.0
        d2_ = torch.nn.functional.normalize(torch.randn(1, ray_o.shape[1], 3, generator=g), dim=-1)
        surf2, edge2, occ2, st2, ot2 = st.sphere_tracing(o2, d2_, torch.full_like(near, 0.02), torch.full_like(near, 0.8), dec, None, None,
                                                         iter=4, offset=0.01, relax=0.0, tan_i=tan_i, soft_shadow=True, dist_th=0.125)
    kw.update(st_o=ray_o[0], st_d=ray_d[0], st_near=near[0], st_far=far[0], st_surf=surf[0], st_occ=occ_[0], st_st=st_[0], st_ot=ot_[0],
              sh_o=o2[0], sh_d=d2_[0], sh_tan_i=tan_i[0], sh_occ=occ2[0], sh_ot=ot2[0])
    # --- J: light visibility on a handful of surface points
    hit = (1 - occ_[0, :, 0]) > 0
    sp = surf[0][hit][:24]
    with torch.no_grad():
        fw = net(sp[None], None, 0.005, batch).raw[0]
    nrm_s = fw[:, 13:16]
    nrm_s = torch.where(nrm_s.sum(-1, keepdim=True) == 0, torch.ones_like(nrm_s), nrm_s)
    nrm_s = net_utils.normalize(nrm_s)
    acc_s = (1 - occ_[0, :, 0])[hit][:24]
    bbox = batch.wbounds.clone()
    bbox[:, 0] -= 0.25
    bbox[:, 1] += 0.25
    shadow_dec = lambda o, d, n, f, *a, **k: st.sphere_tracing(o, d, n, f, dec, None, None, *a, **k)
    with torch.no_grad():
        lvis, ldot = st.light_visibility(sp[None], nrm_s[None], acc_s[None], net.light_xyz, net.light_sharp, shadow_dec, bbox, **cfg.obj_lvis)
    kw.update(lv_surf=sp, lv_norm=nrm_s, lv_acc=acc_s, lv_bbox=bbox[0], lv_lvis=lvis[0].reshape(512, -1), lv_ldot=ldot[0].reshape(512, -1))
    npz('ops.npz', **kw)


if __name__ == '__main__':
    main()
