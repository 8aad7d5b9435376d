"""ctypes binding of the C ABI (include/relightableavatar.h).

The product path has no CPU fallback: if the shared library is missing, or no HIP device is
visible when a context is created, this raises — it never routes through the oracle.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RA_LIB_PATH') or os.path.join(_HERE, 'librelightableavatar_hip.so')    # override: kernel experiments (tools/)
_lib = None
ABI_VERSION = 8          # RA_ABI_VERSION of include/relightableavatar.h


class RaError(RuntimeError):
    pass


class ra_config(C.Structure):
    _fields_ = [('xyz_res', C.c_int), ('sdf_res', C.c_int), ('view_res', C.c_int), ('n_bones', C.c_int), ('relight', C.c_int),
                ('resd_limit', C.c_float), ('blend_radius', C.c_float), ('albedo_slope', C.c_float), ('albedo_bias', C.c_float),
                ('roughness_slope', C.c_float), ('roughness_bias', C.c_float), ('fresnel_f0', C.c_float),
                ('shading_albedo', C.c_float), ('albedo_multiplier', C.c_float), ('lambert_only', C.c_int),
                ('glossy_only', C.c_int), ('tonemapping', C.c_int), ('bg_brightness', C.c_float), ('mlp_f16', C.c_int), ('query_skip', C.c_int),
                ('k4_batch_slots', C.c_int), ('trace_precision', C.c_int), ('clip_near', C.c_float), ('clip_far', C.c_float),
                ('only_visibility', C.c_int), ('vis_shade_map', C.c_int), ('use_geodesic_filter', C.c_int), ('key_light_share', C.c_float)]


class ra_frame(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('R', 'Th', 'poses', 'cond_fix', 'A', 'big_A', 'pverts', 'pnorm', 'tverts', 'weights')] + \
               [('n_verts', C.c_int)]


class ra_trace_params(C.Structure):
    _fields_ = [('iters', C.c_int), ('tan_i', C.c_float), ('tan_i_multiplier', C.c_float), ('relax', C.c_float),
                ('offset', C.c_float), ('eps', C.c_float), ('shadow_skip_iter', C.c_int), ('clay_book', C.c_int),
                ('soft_shadow', C.c_int), ('dist_th', C.c_float)]


RENDER_OUT_KEYS = ('rgb', 'acc', 'depth', 'surf', 'norm', 'albedo', 'roughness', 'shade', 'spec', 'cpts', 'bpts', 'resd',
                   'ray_o', 'lvis', 'ldot', 'raw', 'volume_albedo', 'volume_roughness')


class ra_render_out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in RENDER_OUT_KEYS]


class ra_sphere_params(C.Structure):
    _fields_ = [('surface', ra_trace_params), ('shadow', ra_trace_params), ('shadow_near_offset', C.c_float),
                ('dist_th', C.c_float), ('surf_sample_range', C.c_float), ('n_samples', C.c_int), ('relighting', C.c_int),
                ('no_visibility', C.c_int), ('local_visibility', C.c_int), ('premultiply', C.c_int),
                ('n_boxes', C.c_int), ('boxes', C.POINTER(C.c_float)), ('box_start', C.POINTER(C.c_int))]


class ra_ground_params(C.Structure):
    _fields_ = [('normal', C.c_float * 3), ('origin', C.c_float * 3), ('albedo', C.c_float * 3), ('attach_envmap', C.c_int),
                ('env_r', C.c_float), ('shading_multiplier', C.c_float), ('shadow', ra_trace_params), ('shadow_near_offset', C.c_float),
                ('no_visibility', C.c_int), ('local_visibility', C.c_int), ('n_boxes', C.c_int), ('boxes', C.POINTER(C.c_float)),
                ('box_start', C.POINTER(C.c_int))]


GROUND_OUT_KEYS = ('rgb', 'surf', 'albedo', 'shade', 'spec', 'depth', 'lvis', 'ldot')


class ra_ground_out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in GROUND_OUT_KEYS]


class ra_pose_in(C.Structure):
    _fields_ = [(k, C.POINTER(C.c_float)) for k in ('poses', 'tjoints', 'big_A', 'Rh', 'Th')] + \
               [('parents', C.POINTER(C.c_int)), ('faces', C.POINTER(C.c_int)), ('n_bones', C.c_int), ('n_faces', C.c_int), ('n_verts', C.c_int),
                ('tverts', C.c_void_p), ('weights', C.c_void_p), ('bounds_padding', C.c_float)]


POSE_OUT_KEYS = ('A', 'joints', 'tpose', 'pverts', 'wverts', 'pnorm', 'R', 'pbounds', 'wbounds', 'poses', 'Th')


class ra_pose_out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in POSE_OUT_KEYS]


class ra_image_params(C.Structure):
    _fields_ = [('type', C.c_int), ('H', C.c_int), ('W', C.c_int), ('bg_brightness', C.c_float), ('normalize', C.c_int), ('tonemap', C.c_int),
                ('min_clip', C.c_float), ('cam_R', C.c_float * 9), ('tbounds', C.c_float * 6)]


class ra_counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ('n_coarse', 'n_fine_sdf', 'n_fine_full', 'n_shadow_rays', 'n_hit_pixels', 'n_shaded', 'n_fine_sdf_wide', 'n_fine_sdf_comp')]


# every symbol include/relightableavatar.h declares
SYMBOLS = {
    'ra_last_error': (C.c_char_p, []),
    'ra_abi_version': (C.c_int, []),
    'ra_default_config': (C.c_int, [C.POINTER(ra_config)]),
    'ra_shard_plan': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 7),
    'ra_ctx_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    'ra_ctx_destroy': (C.c_int, [C.c_void_p]),
    'ra_set_config': (C.c_int, [C.c_void_p, C.POINTER(ra_config)]),
    'ra_set_weight': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    'ra_finalize_weights': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ra_set_frame': (C.c_int, [C.c_void_p, C.POINTER(ra_frame), C.c_void_p]),
    'ra_hdq_sdf': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    'ra_observed_sdf': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'ra_bigpose_transform': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'ra_raw_channels': (C.c_int, [C.c_void_p]),
    'ra_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    'ra_sphere_trace': (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int, C.POINTER(ra_trace_params)] + [C.c_void_p] * 5),
    'ra_render_sphere_chunk': (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(ra_sphere_params), C.POINTER(ra_render_out), C.c_void_p]),
    'ra_render_volume_chunk': (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.POINTER(ra_render_out), C.c_void_p]),
    'ra_render_ground_chunk': (C.c_int, [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(ra_ground_params), C.POINTER(ra_ground_out), C.c_void_p]),
    'ra_blend_ground': (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ra_reshade': (C.c_int, [C.c_void_p] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4),
    'ra_k3cc_enabled': (C.c_int, [C.c_void_p]),
    'ra_begin_render': (C.c_int, [C.c_void_p]),
    'ra_debug_key_lights': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ra_scatter_rows': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    'ra_gather_rays': (C.c_int, [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 9),
    'ra_set_key_probes': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'ra_reshade_ground': (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] +
                          [C.c_void_p] * 5),
    'ra_get_counters': (C.c_int, [C.c_void_p, C.POINTER(ra_counters), C.c_void_p]),
    'ra_reset_counters': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ra_get_mlp_time': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p]),
    'ra_get_kernel_time': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p]),
    'ra_enable_timing': (C.c_int, [C.c_void_p, C.c_int]),
    'ra_set_knn_mode': (C.c_int, [C.c_void_p, C.c_int]),
    'ra_gate_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    'ra_gate_destroy': (C.c_int, [C.c_void_p]),
    'ra_set_gate': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ra_pose_frame': (C.c_int, [C.c_void_p, C.POINTER(ra_pose_in), C.POINTER(ra_pose_out), C.c_void_p]),
    'ra_grow_bounds': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    'ra_shift_envmap': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    'ra_add_light_probe': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p]),
    'ra_map_to_image': (C.c_int, [C.c_void_p, C.POINTER(ra_image_params)] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ra_gen_rays': (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_float), C.c_void_p] + [C.c_void_p] * 5 +
                    [C.POINTER(C.c_int), C.c_void_p, C.c_void_p]),
    'ra_debug_mlp': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ra_debug_full': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ra_debug_aabb': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]),
    'ra_debug_lvis': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(ra_trace_params), C.c_float,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    'ra_debug_brdf': (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'ra_debug_bvh_ids': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    'ra_debug_hdq': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 7 + [C.POINTER(C.c_int), C.c_void_p]),
}


def build(verbose: bool = False) -> str:
    """compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j8']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RaError('building the HIP extension failed:\n' + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RaError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                      f'(make -C relightableavatar_amd/csrc). There is no CPU fallback for the render path.')
    # torch first: its wheel bundles its own libamdhip64 / libhsa-runtime64.  Loaded after ours, the process ends up with two HIP runtimes
    # and ours sees no device ("no ROCm-capable device is detected" in ra_ctx_create when build() preceded the first torch import);
    # loaded before, our library's libamdhip64.so.7 resolves to the copy that is already there.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)      # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if L.ra_abi_version() != ABI_VERSION:
        raise RaError('ABI version mismatch')
    _lib = L
    return L


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib().ra_last_error()
        raise RaError(f'{what}: {msg.decode() if msg else "unknown error"}')
