"""Resolved hot-path constants.

The reference reads a global yacs tree (lib/config/config.py:34-425 + configs/base.yaml +
configs/mobile_stage/xuzhen_12v_geo.yaml, merged by update_cfg config.py:487-519).  The
config *system* is out of scope; what the hot path consumes is the
flat set of keys below, with the reference's key names and default values.  ``make_cfg``
reproduces the mode merges for the BASELINE experiments.
"""
from .base_utils import dotdict

_ACTIVE = None


def set_active_cfg(cfg):
    """the reference has one global cfg; plugin classes take no arguments and read the active one."""
    global _ACTIVE
    _ACTIVE = cfg


def active_cfg():
    global _ACTIVE
    if _ACTIVE is None:
        _ACTIVE = default_cfg()
    return _ACTIVE


# Switches of the reference's hot path that this build does NOT reproduce — each either cannot run in the reference release itself or needs
# a third-party CUDA extension.  A cfg carrying one of them away from its default is refused (make_renderer / make_network), never ignored.
UNSUPPORTED = {
    'bruteforce_st': (False, 'render_bruteforce_human (sphere_tracing_renderer.py:787-940) needs lib/networks/relight/nerfactor_network.py, '
                             'which the reference release does not contain'),
    'smpl_distance': (False, 'the SMPL mesh distance (base_network.py:417-427) needs the bvh_distance_queries CUDA extension'),
    'ablate_hdq_mode': ('hdq', "the world / can / curve tracing modes (sphere_tracing_renderer.py:141-151) call world_to_bigpose_transform, "
                               "which raises on the dataset's Th of shape (B, 1, 3); the operators exist: ra_observed_sdf, ra_bigpose_transform"),
    'check_bound_sdf': (False, 'debug colour map of |sdf| at termination (sphere_tracing_renderer.py:577-587, needs easyvolcap)'),
    'geometry_normal': (False, 'part of render_bruteforce_human'),
    'geometry_visibility': (False, 'part of render_bruteforce_human'),
    'zero_roughness': (False, 'part of render_bruteforce_human'),
}


def check_supported(cfg):
    for k, (default, why) in UNSUPPORTED.items():
        if k in cfg and cfg[k] != default:
            raise NotImplementedError(f'cfg.{k} = {cfg[k]!r} is not supported: {why}')


def default_cfg() -> dotdict:
    c = dotdict()
    # plugin selection (lib/networks/make_network.py:4-7, renderer/make_renderer.py:5-8)
    c.network_module = 'relightableavatar_amd.networks.deform.base_network'
    c.renderer_module = 'relightableavatar_amd.renderer.base_renderer'
    # network shape (configs/base.yaml:47-49, config.py:224-231,441,465-466)
    c.n_bones = 52
    c.cond_dim = 156
    c.xyz_res = 10
    c.sdf_res = 8
    c.view_res = 4
    c.feat_dim = 256
    c.resd_limit = 0.05
    c.sdf_beta_init_value = 0.1
    c.blend_radius = 0.075       # config.py:191
    c.sample_vert_cnt = 3        # config.py:192
    c.use_geodesic_filter = True
    c.lambertian = False
    # thresholds / sampling (base.yaml:77-97)
    c.dist_th = 0.1
    c.n_samples = 128
    c.perturb = 1.0
    c.clip_near = 0.02
    c.clip_far = 10.0
    c.bg_brightness = 0.0
    c.render_chunk_size = 8192
    c.volume_chunk_rays = 65536  # not in the reference: rays per volume-path launch sequence on a 288 GB device (the image does not depend on it); 0 = render_chunk_size
    c.sphere_chunk_rays = 262144  # not in the reference: rays per launch sequence of the sphere-tracing renderers (consecutive render chunks merged, the shadow rays of every ray clipped against its own chunk's box: the image does not depend on it; a 1024 x 1024 frame runs ONE 16-iteration surface loop instead of three); 0 = render_chunk_size
    c.ground_chunk_rays = 262144  # not in the reference: pixels per launch sequence of the ground-plane pass (consecutive render chunks merged, every pixel clipped against its own chunk's box: the image does not depend on it); 0 = render_chunk_size
    c.network_chunk_size = 262144
    c.fix_material = 0           # xuzhen_12v_geo.yaml:25
    c.always_fix_material = True
    # sphere tracing (config.py:116-124)
    c.sphere_tracing = dotdict(iter=16, tan_i=1000.0, relax=0.0, offset=0.02, eps=1e-8,
                               near_offset=0.01, shadow_skip_iter=1, tan_i_multiplier=1.0)
    c.obj_lvis = dotdict(iter=4, offset=0.01, relax=0.0, near_offset=0.02, dist_th=0.05)   # config.py:127-132
    c.env_lvis = dotdict(iter=16, offset=0.01, relax=0.0, near_offset=0.02, bbox_margin=0.25, dist_th=0.005)
    c.no_claybook = False
    c.no_dfss = False
    c.no_visibility = False
    c.local_visibility = False
    c.surf_sample_range = 0.005  # config.py:76
    # relighting (config.py:84-113,405-419)
    c.relighting = False
    c.achro_light = False
    c.envmap_upscale = 2
    c.envmap_init_intensity = 0.2
    c.env_h = 16
    c.env_w = 32
    c.env_r = 10.0
    c.fresnel_f0 = 0.02
    c.lambert_only = False
    c.glossy_only = False
    c.albedo_slope = 1.0
    c.albedo_bias = 0.0
    c.roughness_slope = 0.90
    c.roughness_bias = 0.09
    c.relight_network_width = 128
    c.relight_network_depth = 2
    c.albedo_multiplier = 1.0
    c.shading_albedo = 0.8
    c.tonemapping_rendering = True
    c.only_visibility = False
    c.rgb_as_albedo = False
    # visualisation switches that change what render() returns
    c.vis_rendering_map = True
    c.vis_shading_map = False
    c.vis_specular_map = False
    c.vis_novel_light = False
    c.vis_ground_shading = False
    # ground-plane pass (config.py:104-107, 45, 353; render_ground sphere_tracing_renderer.py:463-548)
    c.ground_normal = [0.0, 0.0, 1.0]
    c.ground_origin = [0.0, 0.0, 0.0]
    c.ground_albedo = [0.05, 0.05, 0.05]
    c.ground_attach_envmap = True
    c.ground_shading_multiplier = 1.0
    c.vis_lvis_map = False
    c.vis_ldot_map = False
    c.replace_light = ''
    c.test_light = ['main']
    c.vis_rotate_light = False
    c.rotate_ratio = 4            # config.py:350
    c.probe_size_ratio = 0.2      # config.py:354
    c.env_image_w = 2048
    # visualiser normalisations (config.py:41-46,384,398,416)
    c.normalize_shading = False
    c.normalize_specular = True
    c.min_clip = 1.0
    c.vis_median_depth = False
    c.store_alpha_channel = True
    c.tonemapping_albedo = True
    # build-side knobs (not in the reference)
    c.mlp_dtype = 'f16'          # element type of the fused MLP kernels: 'f16' or 'bf16' (fp32 accumulate either way)
    c.ret_raw = True             # sphere-tracing renderer: also return render_human's per-hit raw / volume_albedo / volume_roughness (lazily)
    c.novel_light_timing = True  # novel-light renderer: bracket the main pass with device syncs to fill `diff` like the reference (:107-112)
    c.query_skip = True          # rays that did not move since their last distance query are not queried again (exact; False = the reference's schedule)
    # arithmetic of the distance queries inside tracing loops (include/relightableavatar.h ra_config.trace_precision): 1 = the surface
    # trace in compensated arithmetic (f16 hi + lo pairs, 3 MFMAs per k-step: as accurate as the reference's fp32), shadow rays plain;
    # 0 = plain 16-bit operands everywhere; 2 = compensated everywhere (validation)
    c.trace_precision = 1
    # with trace_precision 1: shadow rays towards the frame's key lights (each holds >= this fraction of a probe's power, four times the mean share of 512
    # lights; the 48 strongest at most) are traced in compensated arithmetic too (ra_config.key_light_share; 0 = off, round 5's behaviour)
    c.key_light_share = 0.0078
    c.k4_batch_slots = 0         # full queries per forward+backward launch pair (bounds the 4.9 KB/slot activation tape); 0 = 1 Mi
    return c


def make_cfg(mode: str = 'anisdf', **overrides) -> dotdict:
    """mode: 'anisdf' (volume), 'sphere_tracing', 'relight', 'novel_light'.

    Mirrors update_cfg (config.py:487-519): sphere_tracing_cfg (base.yaml:132-136),
    relighting_cfg (base.yaml:138-201 + xuzhen_12v_geo.yaml:46-58), novel_light_cfg
    (base.yaml:193-195).
    """
    c = default_cfg()
    if mode in ('sphere_tracing', 'relight', 'novel_light'):
        c.n_samples = 3
        c.render_chunk_size = 65536
        c.network_chunk_size = 1048576
        c.renderer_module = 'relightableavatar_amd.renderer.sphere_tracing_renderer'
    if mode in ('relight', 'novel_light'):
        c.relighting = True
        c.dist_th = 0.125
        c.obj_lvis.dist_th = 0.125
        c.achro_light = True
        c.network_module = 'relightableavatar_amd.networks.relight.relight_network'
    if mode == 'novel_light':
        c.vis_novel_light = True
        c.renderer_module = 'relightableavatar_amd.renderer.novel_light_sphere_tracing'
    elif mode not in ('anisdf', 'sphere_tracing', 'relight'):
        raise ValueError(f'unknown mode {mode}')
    for k, v in overrides.items():
        if isinstance(v, dict) and isinstance(c.get(k), dict):
            c[k].update(v)
        else:
            c[k] = v
    return c
