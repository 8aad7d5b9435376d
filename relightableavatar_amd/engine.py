"""Engine: thin Python owner of one ra_ctx (the C ABI context).

PyTorch is plumbing here: it owns device memory (torch tensors whose data_ptr() is handed to the
library) and the stream.  All arithmetic of the render path happens in the HIP library.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (check, ra_config, ra_counters, ra_frame, ra_ground_out, ra_ground_params, ra_pose_in, ra_pose_out, ra_render_out,
                   ra_sphere_params, ra_trace_params)
from .base_utils import dotdict


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class RaysPending:
    """rays generated on the device whose count has not been read yet (Engine.gen_rays_async)"""

    def __init__(self, bufs, host, mask_host, event, H, W, keep):
        self._bufs, self._host, self._mask_host, self._event, self.H, self.W, self._keep = bufs, host, mask_host, event, H, W, keep

    def ready(self):
        return self._event.query()

    def result(self):
        """ray_o, ray_d (P,3), near, far (P,), mask_at_box (H,W) bool on the device; wbounds_host (1,2,3) and — if requested —
        mask_host (H*W,) bool on the host.  Waits for the generating kernels only (an event), not for the stream."""
        self._event.synchronize()
        P = int(self._host[:1].view(torch.int32).item())
        ro, rd, near, far, mask = self._bufs
        # the buffers were allocated on the issuing stream; whoever consumes them on another stream must keep the caching allocator
        # from recycling the blocks under it (advisor, round 4)
        cur = torch.cuda.current_stream(ro.device)
        for b in self._bufs:
            b.record_stream(cur)
        out = dotdict(ray_o=ro[:P], ray_d=rd[:P], near=near[:P], far=far[:P], mask_at_box=mask.view(self.H, self.W).bool(),
                      wbounds_host=self._host[1:7].clone().reshape(1, 2, 3))
        if self._mask_host is not None:
            out.mask_host = self._mask_host.bool()
        return out


class Engine:
    def __init__(self, cfg, device=None, relight=False):
        if not torch.cuda.is_available():
            raise _lib.RaError('relightableavatar_amd needs a HIP device (MI355X); there is no CPU fallback for the render path')
        self.lib = _lib.lib()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.cfg = cfg
        self.ctx = C.c_void_p()
        check(self.lib.ra_ctx_create(C.byref(self.ctx), self.device.index or 0), 'ra_ctx_create')
        self.relight = bool(relight)      # RelightableAvatar heads (17-ch raw) vs AniSDF colour net (16-ch raw)
        c = ra_config(xyz_res=cfg.xyz_res, sdf_res=cfg.sdf_res, view_res=cfg.view_res, n_bones=cfg.n_bones, relight=int(self.relight),
                      resd_limit=cfg.resd_limit, blend_radius=cfg.blend_radius, albedo_slope=cfg.albedo_slope,
                      albedo_bias=cfg.albedo_bias, roughness_slope=cfg.roughness_slope, roughness_bias=cfg.roughness_bias,
                      fresnel_f0=cfg.fresnel_f0, shading_albedo=cfg.shading_albedo, albedo_multiplier=cfg.albedo_multiplier,
                      lambert_only=int(cfg.lambert_only), glossy_only=int(cfg.glossy_only),
                      tonemapping=int(cfg.tonemapping_rendering), bg_brightness=cfg.bg_brightness,
                      mlp_f16=int(cfg.mlp_dtype == 'f16'), query_skip=int(cfg.get('query_skip', True)),
                      k4_batch_slots=int(cfg.get('k4_batch_slots', 0)), trace_precision=int(cfg.get('trace_precision', 1)),
                      clip_near=float(cfg.get('clip_near', 0.02)), clip_far=float(cfg.get('clip_far', 10.0)),
                      only_visibility=int(bool(cfg.get('only_visibility', False))),
                      vis_shade_map=2 if cfg.get('vis_ldot_map', False) else (1 if cfg.get('vis_lvis_map', False) else 0),
                      use_geodesic_filter=int(bool(cfg.get('use_geodesic_filter', True))),
                      key_light_share=float(cfg.get('key_light_share', 0.0078)))
        assert cfg.mlp_dtype in ('f16', 'bf16')
        check(self.lib.ra_set_config(self.ctx, C.byref(c)), 'ra_set_config')
        self._frame_key = None
        self._frame_refs = None
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, 'ctx', None) and self.ctx.value:
                self.lib.ra_ctx_destroy(self.ctx)
                self.ctx = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_state_dict(self, sd: dict):
        """hand every tensor of the reference state_dict to the packer (host fp32)."""
        for k, v in sd.items():
            if not isinstance(v, torch.Tensor) or not v.dtype.is_floating_point:
                continue
            h = v.detach().to('cpu', torch.float32).contiguous()
            check(self.lib.ra_set_weight(self.ctx, k.encode(), C.c_void_p(h.data_ptr()), h.numel()), f'ra_set_weight({k})')
        check(self.lib.ra_finalize_weights(self.ctx, self.stream), 'ra_finalize_weights')
        self._frame_key = None

    FRAME_KEYS = ('R', 'Th', 'poses', 'A', 'big_A', 'pverts', 'pnorm', 'tverts', 'weights')

    def _cond_fix_source(self, batch):
        """the pose the colour net is conditioned on in eval mode (base_network.py:501-503): with
        `cfg.fix_material >= 0 or cfg.always_fix_material` it is train_motion.poses[:, cfg.fix_material] (so -1 selects the last
        training pose) and a batch without train_motion is an error, exactly as in the reference; otherwise the current pose.
        The relight network has no pose-conditioned head (relight_network.py:91-104)."""
        if self.relight:
            return None
        c = self.cfg
        if c.fix_material >= 0 or c.always_fix_material:
            tm = batch.get('train_motion', None) if hasattr(batch, 'get') else None
            if tm is None or 'poses' not in tm:
                raise ValueError('cfg.fix_material / cfg.always_fix_material need batch.train_motion.poses (base_network.py:502-503)')
            return tm['poses']
        return batch['poses']

    def set_frame(self, batch, force=False):
        """upload the SMPL frame state (batch keys of SURVEY.md section 8b).  The upload is skipped only when every tensor of the
        frame is the SAME live object (held by a strong reference, so its address cannot be recycled for another frame) at the
        same in-place version as in the previous call, and the fixed-material index is unchanged."""
        src = [batch[k] for k in self.FRAME_KEYS]
        cf_src = self._cond_fix_source(batch)
        refs = src + [cf_src]
        key = tuple(None if t is None else t._version for t in refs) + (int(self.cfg.fix_material), bool(self.cfg.always_fix_material))
        if (not force and self._frame_key == key and self._frame_refs is not None and len(self._frame_refs) == len(refs)
                and all(a is b for a, b in zip(self._frame_refs, refs))):
            return
        d = self.device
        t = {k: _f32(batch[k][0], d) for k in self.FRAME_KEYS}
        cond_fix = None
        if cf_src is not None:
            cond_fix = t['poses'] if cf_src is batch['poses'] else _f32(cf_src[0, self.cfg.fix_material], d)
        fr = ra_frame(R=_ptr(t['R']), Th=_ptr(t['Th']), poses=_ptr(t['poses']), cond_fix=_ptr(cond_fix), A=_ptr(t['A']),
                      big_A=_ptr(t['big_A']), pverts=_ptr(t['pverts']), pnorm=_ptr(t['pnorm']), tverts=_ptr(t['tverts']),
                      weights=_ptr(t['weights']), n_verts=int(t['pverts'].shape[0]))
        assert t['weights'].shape[-1] == self.cfg.n_bones, 'batch.weights does not match cfg.n_bones'
        check(self.lib.ra_set_frame(self.ctx, C.byref(fr), self.stream), 'ra_set_frame')
        self._keep = [t, cond_fix]   # the library copies asynchronously on the stream
        self._frame_key, self._frame_refs = key, refs

    # ------------------------------------------------------------------ operators
    def hdq_sdf(self, x: torch.Tensor, dist_th: float, smooth: bool) -> torch.Tensor:
        x = _f32(x.reshape(-1, 3), self.device)
        out = torch.empty(x.shape[0], device=self.device, dtype=torch.float32)
        check(self.lib.ra_hdq_sdf(self.ctx, _ptr(x), x.shape[0], dist_th, int(smooth), _ptr(out), self.stream), 'ra_hdq_sdf')
        return out

    def observed_sdf(self, bpts: torch.Tensor) -> torch.Tensor:
        """SDF(bpts + resd(bpts)) on big-pose points through the production distance-query kernel (no coarse level)."""
        bpts = _f32(bpts.reshape(-1, 3), self.device)
        out = torch.empty(bpts.shape[0], device=self.device, dtype=torch.float32)
        check(self.lib.ra_observed_sdf(self.ctx, _ptr(bpts), bpts.shape[0], _ptr(out), self.stream), 'ra_observed_sdf')
        return out

    def bigpose_transform(self, x: torch.Tensor, R: torch.Tensor, Th: torch.Tensor, invert=False) -> torch.Tensor:
        """(n,4,4) world -> big-pose transforms (or their affine inverses) of the points x against the current frame."""
        x = _f32(x.reshape(-1, 3), self.device)
        R, Th = _f32(R.reshape(3, 3), self.device), _f32(Th.reshape(3), self.device)
        out = torch.empty(x.shape[0], 4, 4, device=self.device, dtype=torch.float32)
        check(self.lib.ra_bigpose_transform(self.ctx, _ptr(x), x.shape[0], _ptr(R), _ptr(Th), int(invert), _ptr(out), self.stream),
              'ra_bigpose_transform')
        return out

    def forward(self, x: torch.Tensor, v, dist_th: float) -> torch.Tensor:
        x = _f32(x.reshape(-1, 3), self.device)
        v = None if v is None else _f32(v.reshape(-1, 3), self.device)
        C_ = self.lib.ra_raw_channels(self.ctx)
        raw = torch.empty(x.shape[0], C_, device=self.device, dtype=torch.float32)
        check(self.lib.ra_forward(self.ctx, _ptr(x), _ptr(v), x.shape[0], dist_th, _ptr(raw), self.stream), 'ra_forward')
        return raw

    def trace_params(self, tc, dist_th, soft, iters=None) -> ra_trace_params:
        st = self.cfg.sphere_tracing
        return ra_trace_params(iters=int(tc.iter if iters is None else iters), tan_i=st.tan_i, tan_i_multiplier=st.tan_i_multiplier,
                               relax=tc.relax, offset=tc.offset, eps=st.eps, shadow_skip_iter=st.shadow_skip_iter,
                               clay_book=int(not self.cfg.no_claybook), soft_shadow=int(soft), dist_th=dist_th)

    def sphere_trace(self, ray_o, ray_d, near, far, params: ra_trace_params, tan_i=None):
        d = self.device
        ray_o, ray_d = _f32(ray_o.reshape(-1, 3), d), _f32(ray_d.reshape(-1, 3), d)
        near, far = _f32(near.reshape(-1), d), _f32(far.reshape(-1), d)
        tan_i = None if tan_i is None else _f32(tan_i.reshape(-1), d)
        n = ray_o.shape[0]
        surf = torch.empty(n, 3, device=d)
        occ, st, ot = (torch.empty(n, device=d) for _ in range(3))
        check(self.lib.ra_sphere_trace(self.ctx, _ptr(ray_o), _ptr(ray_d), _ptr(near), _ptr(far), _ptr(tan_i), n, C.byref(params),
                                       _ptr(surf), _ptr(occ), _ptr(st), _ptr(ot), self.stream), 'ra_sphere_trace')
        return surf, occ, st, ot

    def sphere_params(self) -> ra_sphere_params:
        c = self.cfg
        return ra_sphere_params(surface=self.trace_params(c.sphere_tracing, c.dist_th, False),
                                shadow=self.trace_params(c.obj_lvis, c.obj_lvis.dist_th, not c.no_dfss),
                                shadow_near_offset=c.obj_lvis.near_offset, dist_th=c.dist_th,
                                surf_sample_range=c.surf_sample_range, n_samples=c.n_samples, relighting=int(c.relighting),
                                no_visibility=int(c.no_visibility), local_visibility=int(c.local_visibility), premultiply=1)

    def render_sphere_chunk(self, ray_o, ray_d, near, far, bbox6, probe, params, outs: dict, boxes=None, box_start=None):
        """ray tensors: contiguous fp32 device views (P,3)/(P,); outs: name -> tensor view or missing.
        boxes / box_start: several of the reference's chunks in this one call, the shadow rays of ray r clipped against the box of its own chunk."""
        P = ray_o.shape[0]
        ro = ra_render_out(**{k: _ptr(outs.get(k)) for k in _lib.RENDER_OUT_KEYS})
        bb = (C.c_float * 6)(*[float(v) for v in bbox6]) if bbox6 is not None else None
        ph, pw = (probe.shape[0], probe.shape[1]) if probe is not None else (0, 0)
        if boxes is not None and len(boxes) > 1:
            flat = (C.c_float * (6 * len(boxes)))(*[float(v) for b in boxes for v in b])
            starts = (C.c_int * (len(boxes) + 1))(*[int(v) for v in box_start])
            params.n_boxes, params.boxes, params.box_start = len(boxes), C.cast(flat, C.POINTER(C.c_float)), C.cast(starts, C.POINTER(C.c_int))
        else:
            params.n_boxes, params.boxes, params.box_start = 0, None, None
        check(self.lib.ra_render_sphere_chunk(self.ctx, _ptr(ray_o), _ptr(ray_d), _ptr(near), _ptr(far), P, bb, _ptr(probe), ph, pw,
                                              C.byref(params), C.byref(ro), self.stream), 'ra_render_sphere_chunk')

    def ground_params(self) -> ra_ground_params:
        c = self.cfg
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
        return ra_ground_params(normal=f3(c.ground_normal), origin=f3(c.ground_origin), albedo=f3(c.ground_albedo),
                                attach_envmap=int(c.ground_attach_envmap), env_r=float(c.env_r),
                                shading_multiplier=float(c.ground_shading_multiplier),
                                shadow=self.trace_params(c.env_lvis, c.env_lvis.dist_th, not c.no_dfss),
                                shadow_near_offset=c.env_lvis.near_offset, no_visibility=int(c.no_visibility),
                                local_visibility=int(c.local_visibility))

    def render_ground_chunk(self, ray_o, ray_d, acc, bbox6, probe, params, outs: dict, boxes=None, box_start=None):
        """N1: one chunk of full-frame rays against the ground plane (render_ground); outs: name -> (P,3)/(P,) views.
        boxes / box_start: several of the reference's chunks in this one call, pixel r clipped against the box of its own chunk."""
        P = ray_o.shape[0]
        go = ra_ground_out(**{k: _ptr(outs.get(k)) for k in _lib.GROUND_OUT_KEYS})
        bb = (C.c_float * 6)(*[float(v) for v in bbox6])
        if boxes is not None and len(boxes) > 1:
            flat = (C.c_float * (6 * len(boxes)))(*[float(v) for b in boxes for v in b])
            starts = (C.c_int * (len(boxes) + 1))(*[int(v) for v in box_start])
            params.n_boxes, params.boxes, params.box_start = len(boxes), C.cast(flat, C.POINTER(C.c_float)), C.cast(starts, C.POINTER(C.c_int))
        else:
            params.n_boxes, params.boxes, params.box_start = 0, None, None
        check(self.lib.ra_render_ground_chunk(self.ctx, _ptr(ray_o), _ptr(ray_d), _ptr(acc), P, bb, _ptr(probe), probe.shape[0],
                                              probe.shape[1], C.byref(params), C.byref(go), self.stream), 'ra_render_ground_chunk')

    def debug_key_lights(self, n_lights):
        """(bool mask, share) of the frame's key lights (test hook)"""
        key = torch.empty(n_lights, dtype=torch.uint8, device=self.device)
        share = torch.empty(n_lights, dtype=torch.float32, device=self.device)
        check(self.lib.ra_debug_key_lights(self.ctx, _ptr(key), _ptr(share), self.stream), 'ra_debug_key_lights')
        return key.bool(), share

    def begin_render(self):
        """one top-level render starts: its chunk calls are numbered from here (launch-variant hints; ra_begin_render)"""
        check(self.lib.ra_begin_render(self.ctx), 'ra_begin_render')

    def k3cc_enabled(self) -> bool:
        """the cooperative small-launch distance kernel passed its on-device self-test and is in use (ra_k3cc_enabled)"""
        return bool(self.lib.ra_k3cc_enabled(self.ctx))

    def set_key_probes(self, probe_sets):
        """name the frame's key lights (ra_config.key_light_share) from every probe its cached visibility will be shaded with:
        probe_sets is a list of (n, h, w, 3) / (h, w, 3) tensors (one entry per probe size); an empty list returns to per-call key lights."""
        if not probe_sets:
            check(self.lib.ra_set_key_probes(self.ctx, None, 0, 0, 0, 0, self.stream), 'ra_set_key_probes')
            return
        for i, pr in enumerate(probe_sets):
            pr = _f32(pr if pr.ndim == 4 else pr[None], self.device)
            self._keep.append(pr)
            check(self.lib.ra_set_key_probes(self.ctx, _ptr(pr), pr.shape[0], pr.shape[1], pr.shape[2], int(i > 0), self.stream), 'ra_set_key_probes')

    def reshade_ground(self, ray_d, albedo_map, lvis, ldot, probes, images=None, attach_envmap=True):
        """novel_light_sphere_tracing.render_ground (:70-99) for all probes at once: probes (n,h,w,3), optional images
        (n,ih,iw,3) -> rgb, albedo, shade, spec each (n,P,3)."""
        d = self.device
        ray_d = _f32(ray_d.reshape(-1, 3), d)
        P = ray_d.shape[0]
        albedo_map = None if albedo_map is None else _f32(albedo_map.reshape(P, 3), d)
        lvis, ldot = _f32(lvis.reshape(P, -1), d), _f32(ldot.reshape(P, -1), d)
        probes = _f32(probes, d)
        n, ph, pw = probes.shape[0], probes.shape[1], probes.shape[2]
        images = None if images is None else _f32(images, d)
        ih, iw = (images.shape[1], images.shape[2]) if images is not None else (0, 0)
        outs = [torch.empty(n, P, 3, device=d) for _ in range(4)]
        check(self.lib.ra_reshade_ground(self.ctx, _ptr(ray_d), _ptr(albedo_map), _ptr(lvis), _ptr(ldot), P, _ptr(probes), n, ph, pw,
                                         _ptr(images), ih, iw, int(bool(attach_envmap)), *[_ptr(o) for o in outs], self.stream),
              'ra_reshade_ground')
        return tuple(outs)

    def blend_ground(self, ground, human, inds, acc):
        """blend_output_'s alpha_blend for one map: ground (F,C)/(F,) or None, human (P,C)/(P,) or None, inds (P) int64, acc (F)."""
        ref = ground if ground is not None else human
        flat = ref.ndim == 1
        C_ = 1 if flat else ref.shape[-1]
        F_ = acc.shape[0]
        g = None if ground is None else _f32(ground, self.device).reshape(F_, C_)
        h = None if human is None else _f32(human, self.device).reshape(-1, C_)
        P = 0 if h is None else h.shape[0]
        dst = torch.empty(F_, C_, device=self.device)
        check(self.lib.ra_blend_ground(self.ctx, _ptr(g), _ptr(h), _ptr(inds), _ptr(acc), F_, P, C_, _ptr(dst), self.stream), 'ra_blend_ground')
        return dst[:, 0] if flat else dst

    def render_volume_chunk(self, ray_o, ray_d, near, far, n_samples, dist_th, outs: dict):
        P = ray_o.shape[0]
        ro = ra_render_out(**{k: _ptr(outs.get(k)) for k in _lib.RENDER_OUT_KEYS})
        check(self.lib.ra_render_volume_chunk(self.ctx, _ptr(ray_o), _ptr(ray_d), _ptr(near), _ptr(far), P, n_samples, dist_th,
                                              C.byref(ro), self.stream), 'ra_render_volume_chunk')

    def reshade(self, ray_o, surf, norm, albedo, rough, lvis, ldot, probes, want_spec=True):
        """probes (n,h,w,3) -> rgb, shade, spec each (n,P,3)."""
        d = self.device
        a = [_f32(t, d) for t in (ray_o.reshape(-1, 3), surf.reshape(-1, 3), norm.reshape(-1, 3), albedo.reshape(-1, 3), rough.reshape(-1))]
        P = a[0].shape[0]
        lvis, ldot = _f32(lvis.reshape(P, -1), d), _f32(ldot.reshape(P, -1), d)
        probes = _f32(probes, d)
        n, ph, pw = probes.shape[0], probes.shape[1], probes.shape[2]
        rgb, shade = torch.empty(n, P, 3, device=d), torch.empty(n, P, 3, device=d)
        spec = torch.empty(n, P, 3, device=d) if want_spec else None
        check(self.lib.ra_reshade(self.ctx, *[_ptr(t) for t in a], _ptr(lvis), _ptr(ldot), P, _ptr(probes), n, ph, pw,
                                  _ptr(rgb), _ptr(shade), _ptr(spec), self.stream), 'ra_reshade')
        return rgb, shade, spec

    # ------------------------------------------------------------------ measurement
    def counters(self) -> dotdict:
        c = ra_counters()
        check(self.lib.ra_get_counters(self.ctx, C.byref(c), self.stream), 'ra_get_counters')
        return dotdict({k: int(getattr(c, k)) for k, _ in ra_counters._fields_})

    def reset_counters(self):
        check(self.lib.ra_reset_counters(self.ctx, self.stream), 'ra_reset_counters')

    def set_gate(self, gate):
        """attach (or, with None, detach) a pipeline.Gate: contexts sharing one run their light-visibility stages one after the other"""
        check(self.lib.ra_set_gate(self.ctx, gate.handle if gate is not None else C.c_void_p()), 'ra_set_gate')
        self._gate = gate        # keeps it alive as long as the context refers to it

    def set_knn_mode(self, use_bvh=True):
        check(self.lib.ra_set_knn_mode(self.ctx, int(use_bvh)), 'ra_set_knn_mode')
        self._frame_key = None

    def gen_rays(self, H, W, K, R, T, bounds, count=None):
        """N2: rays of an H x W pinhole view culled against the body's bounding box, on the device.
        K, R (3,3), T (3,) or (3,1): anything numpy can read; bounds (2,3).  Returns the reference's
        get_rays_within_bounds outputs as device tensors: ray_o, ray_d (P,3), near, far (P,), mask_at_box (H,W) bool.
        count: the number of in-box rays if the caller knows it (H * W for an unbounded box): skips the read-back and its synchronisation."""
        bufs, cnt = self._gen_rays(H, W, K, R, T, bounds, host_count=count is None)
        P = cnt if count is None else int(count)
        ro, rd, near, far, mask = bufs
        return dotdict(ray_o=ro[:P], ray_d=rd[:P], near=near[:P], far=far[:P], mask_at_box=mask.view(int(H), int(W)).bool())

    def _gen_rays(self, H, W, K, R, T, bounds, host_count, count_dev=None):
        import numpy as np
        Kd = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(9))
        Rd = np.ascontiguousarray(np.asarray(R, dtype=np.float64).reshape(9))
        Td = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(3))
        on_dev = isinstance(bounds, torch.Tensor) and bounds.is_cuda
        if on_dev:           # a box that is still being computed on the stream (pose_frame's wbounds): read on the device, no round trip
            bdev = _f32(bounds.reshape(6), self.device)
            bd = None
        else:
            bd = np.ascontiguousarray(np.asarray(bounds.detach().cpu() if isinstance(bounds, torch.Tensor) else bounds, dtype=np.float32).reshape(6))
        n = int(H) * int(W)
        d = self.device
        ro, rd = torch.empty(n, 3, device=d), torch.empty(n, 3, device=d)
        near, far = torch.empty(n, device=d), torch.empty(n, device=d)
        mask = torch.empty(n, device=d, dtype=torch.uint8)
        cnt = C.c_int(0)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        check(self.lib.ra_gen_rays(self.ctx, int(H), int(W), dp(Kd), dp(Rd), dp(Td), None if bd is None else bd.ctypes.data_as(C.POINTER(C.c_float)),
                                   _ptr(bdev) if on_dev else None, _ptr(ro), _ptr(rd), _ptr(near), _ptr(far), _ptr(mask),
                                   C.byref(cnt) if host_count else None, None if count_dev is None else _ptr(count_dev), self.stream), 'ra_gen_rays')
        return (ro, rd, near, far, mask), cnt.value

    def gen_rays_async(self, H, W, K, R, T, bounds, mask_to_host=False):
        """N2 for an animation loop: the rays of a frame whose box (`bounds`: a device tensor, e.g. pose_frame's wbounds) may still be on its
        way.  Nothing waits: the ray buffers have the frame's capacity (H * W), the in-box count and the box travel to pinned host memory
        behind an event, and `.result()` — called a pipeline turn later, when the kernels are long done — trims the buffers.  `mask_to_host`
        also brings mask_at_box to the host (the shard plan of an N-rank job is host work: shard.make_plan)."""
        n = int(H) * int(W)
        d = self.device
        count_dev = torch.empty(1, device=d, dtype=torch.int32)
        bufs, _ = self._gen_rays(H, W, K, R, T, bounds, host_count=False, count_dev=count_dev)
        bdev = _f32(bounds.reshape(6), d) if isinstance(bounds, torch.Tensor) else torch.as_tensor(bounds, dtype=torch.float32, device=d).reshape(6)
        host = torch.empty(8, dtype=torch.float32, pin_memory=True)             # [count (int bits) | 6 box floats]
        host[:1].view(torch.int32).copy_(count_dev, non_blocking=True)
        host[1:7].copy_(bdev, non_blocking=True)
        mask_host = None
        if mask_to_host:
            mask_host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            mask_host.copy_(bufs[4], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(d))
        return RaysPending(bufs, host, mask_host, ev, int(H), int(W), (count_dev, bdev))

    def pose_frame(self, poses, tjoints, parents, tverts, weights, big_A, faces, Rh, Th, padding=0.05):
        """N3: per-frame body state on the device (base_dataset.py:308-397).  Small inputs (poses, tjoints (J,3), parents (J),
        big_A (J,4,4), Rh, Th, faces (F,3)) are read from host memory; tverts (N,3) and weights (N,J) are device tensors
        (moved if needed).  Returns device tensors A (J,4,4), joints, tverts (T pose), pverts, wverts, pnorm, R, pbounds, wbounds, and the
        device copies of poses (J,3) / Th (3).  Asynchronous: nothing waits for the stream (the host inputs are staged in pinned memory)."""
        import numpy as np
        h32 = lambda a: np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
        hp, hj, hb, hr, ht = h32(poses).reshape(-1, 3), h32(tjoints).reshape(-1, 3), h32(big_A).reshape(-1, 16), h32(Rh).reshape(3), h32(Th).reshape(3)
        par = np.ascontiguousarray(parents.cpu().numpy() if isinstance(parents, torch.Tensor) else parents, dtype=np.int32)
        fc = np.ascontiguousarray(faces.cpu().numpy() if isinstance(faces, torch.Tensor) else faces, dtype=np.int32).reshape(-1, 3)
        d = self.device
        tv, w = _f32(tverts.reshape(-1, 3) if isinstance(tverts, torch.Tensor) else torch.as_tensor(tverts).reshape(-1, 3), d), \
            _f32(weights if isinstance(weights, torch.Tensor) else torch.as_tensor(weights), d)
        w = w.reshape(tv.shape[0], -1)
        J, N = hp.shape[0], tv.shape[0]
        o = dotdict(A=torch.empty(J, 4, 4, device=d), joints=torch.empty(J, 3, device=d), tverts=torch.empty(N, 3, device=d),
                    pverts=torch.empty(N, 3, device=d), wverts=torch.empty(N, 3, device=d), pnorm=torch.empty(N, 3, device=d),
                    R=torch.empty(3, 3, device=d), pbounds=torch.empty(2, 3, device=d), wbounds=torch.empty(2, 3, device=d),
                    poses=torch.empty(J, 3, device=d), Th=torch.empty(3, device=d))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        pin = ra_pose_in(poses=fp(hp), tjoints=fp(hj), big_A=fp(hb), Rh=fp(hr), Th=fp(ht), parents=par.ctypes.data_as(C.POINTER(C.c_int)),
                         faces=fc.ctypes.data_as(C.POINTER(C.c_int)), n_bones=J, n_faces=fc.shape[0], n_verts=N, tverts=_ptr(tv), weights=_ptr(w),
                         bounds_padding=float(padding))
        pout = ra_pose_out(A=_ptr(o.A), joints=_ptr(o.joints), tpose=_ptr(o.tverts), pverts=_ptr(o.pverts), wverts=_ptr(o.wverts),
                           pnorm=_ptr(o.pnorm), R=_ptr(o.R), pbounds=_ptr(o.pbounds), wbounds=_ptr(o.wbounds), poses=_ptr(o.poses), Th=_ptr(o.Th))
        check(self.lib.ra_pose_frame(self.ctx, C.byref(pin), C.byref(pout), self.stream), 'ra_pose_frame')
        return o

    def grow_bounds(self, wbounds, margin):
        """batch.wbounds (1,2,3) -= / += margin in place, one launch (the reference's per-chunk quirk, sphere_tracing_renderer.py:1020-1022)"""
        assert wbounds.is_cuda and wbounds.dtype == torch.float32 and wbounds.is_contiguous() and wbounds.numel() == 6
        check(self.lib.ra_grow_bounds(self.ctx, _ptr(wbounds), float(margin), self.stream), 'ra_grow_bounds')
        torch.autograd.graph.increment_version(wbounds)      # an in-place write torch did not see: keep its version counter honest
        return wbounds

    def shift_envmap(self, image, shift):
        """N4: rotate_envmap's shift_image: (H,W,C) or (1,H,W,C) -> same shape, shifted `shift` pixels with wrap-around."""
        x = _f32(image, self.device)
        lead = x.shape[:-3]
        x3 = x.reshape(x.shape[-3:])
        out = torch.empty_like(x3)
        check(self.lib.ra_shift_envmap(self.ctx, _ptr(x3), x3.shape[0], x3.shape[1], x3.shape[2], float(shift), _ptr(out), self.stream),
              'ra_shift_envmap')
        return out.reshape(*lead, *x3.shape)

    def add_light_probe(self, rgb, probe, H, W, cam_R, uH, uW):
        """N4: add_light_probe: rgb (..., H*W, 3) gets the probe inset in its top-left uH x uW pixels (in place on a device copy)."""
        import numpy as np
        out = _f32(rgb, self.device).clone()
        pr = _f32(probe[0] if probe.ndim == 4 else probe, self.device)
        R = np.ascontiguousarray(np.asarray(cam_R.detach().cpu() if isinstance(cam_R, torch.Tensor) else cam_R, dtype=np.float32).reshape(9))
        check(self.lib.ra_add_light_probe(self.ctx, _ptr(out), int(H), int(W), _ptr(pr), pr.shape[0], pr.shape[1],
                                          R.ctypes.data_as(C.POINTER(C.c_float)), int(uH), int(uW), self.stream), 'ra_add_light_probe')
        return out

    def enable_timing(self, on=True):
        check(self.lib.ra_enable_timing(self.ctx, int(on)), 'ra_enable_timing')

    def mlp_time(self):
        ms, n = C.c_float(), C.c_int()
        check(self.lib.ra_get_mlp_time(self.ctx, C.byref(ms), C.byref(n), self.stream), 'ra_get_mlp_time')
        return float(ms.value), int(n.value)

    def kernel_time(self, kind):
        """(ms, launches) of one kernel family since the last reset: 0 = fused distance query (K3, every width), 1 = full query (K4),
        2 = the 8-wave K3 only (launches that fill the chip), 3 = the 2- / 4-wave K3."""
        ms, n = C.c_float(), C.c_int()
        check(self.lib.ra_get_kernel_time(self.ctx, int(kind), C.byref(ms), C.byref(n), self.stream), 'ra_get_kernel_time')
        return float(ms.value), int(n.value)

    # ------------------------------------------------------------------ test hooks
    def debug_mlp(self, bpts, want_feat=True):
        d = self.device
        bpts = _f32(bpts.reshape(-1, 3), d)
        n = bpts.shape[0]
        resd, sdf = torch.empty(n, 3, device=d), torch.empty(n, device=d)
        feat = torch.empty(n, 256, device=d) if want_feat else None
        check(self.lib.ra_debug_mlp(self.ctx, _ptr(bpts), n, _ptr(resd), _ptr(sdf), _ptr(feat), self.stream), 'ra_debug_mlp')
        return resd, sdf, feat

    def debug_full(self, bpts):
        d = self.device
        bpts = _f32(bpts.reshape(-1, 3), d)
        n = bpts.shape[0]
        grad, sdf, feat = torch.empty(n, 3, device=d), torch.empty(n, device=d), torch.empty(n, 256, device=d)
        raw = torch.zeros(n, self.lib.ra_raw_channels(self.ctx), device=d)
        check(self.lib.ra_debug_full(self.ctx, _ptr(bpts), n, _ptr(grad), _ptr(sdf), _ptr(feat), _ptr(raw), self.stream), 'ra_debug_full')
        return grad, sdf, feat, raw

    def debug_aabb(self, o, d, bbox6):
        dv = self.device
        o, d = _f32(o.reshape(-1, 3), dv), _f32(d.reshape(-1, 3), dv)
        n = o.shape[0]
        near, far = torch.empty(n, device=dv), torch.empty(n, device=dv)
        bb = (C.c_float * 6)(*[float(v) for v in bbox6])
        check(self.lib.ra_debug_aabb(self.ctx, _ptr(o), _ptr(d), n, bb, _ptr(near), _ptr(far), self.stream), 'ra_debug_aabb')
        return near, far

    def debug_lvis(self, surf, norm, acc, bbox6, lvis_cfg=None):
        """light_visibility on given surface points: (n,512) lvis, ldot"""
        dv = self.device
        c = self.cfg
        lv = c.obj_lvis if lvis_cfg is None else lvis_cfg
        surf, norm, acc = _f32(surf.reshape(-1, 3), dv), _f32(norm.reshape(-1, 3), dv), _f32(acc.reshape(-1), dv)
        n = surf.shape[0]
        L = c.env_h * c.env_w
        lvis, ldot = torch.empty(n, L, device=dv), torch.empty(n, L, device=dv)
        bb = (C.c_float * 6)(*[float(v) for v in bbox6])
        p = self.trace_params(lv, lv.dist_th, not c.no_dfss)
        check(self.lib.ra_debug_lvis(self.ctx, _ptr(surf), _ptr(norm), _ptr(acc), n, bb, C.byref(p), float(lv.near_offset), _ptr(lvis), _ptr(ldot),
                                     self.stream), 'ra_debug_lvis')
        return lvis, ldot

    def debug_brdf(self, p2l, p2c, normal, albedo, rough):
        """p2l (L,N,3), others per point -> (L,N,3)"""
        dv = self.device
        L, N = p2l.shape[0], p2l.shape[1]
        a = [_f32(t, dv) for t in (p2l.reshape(L * N, 3), p2c.reshape(N, 3), normal.reshape(N, 3), albedo.reshape(N, 3), rough.reshape(N))]
        out = torch.empty(L, N, 3, device=dv)
        check(self.lib.ra_debug_brdf(self.ctx, *[_ptr(t) for t in a], L, N, _ptr(out), self.stream), 'ra_debug_brdf')
        return out

    def debug_bvh_ids(self):
        """vertex ids of the frame's box structure in leaf order (numpy int32; padding 0x7fffffff), empty without a structure"""
        import numpy as np
        cap = 32 * 1024
        ids = np.empty(cap, dtype=np.int32)
        n = C.c_int()
        check(self.lib.ra_debug_bvh_ids(self.ctx, ids.ctypes.data_as(C.c_void_p), cap, C.byref(n), self.stream), 'ra_debug_bvh_ids')
        return ids[:n.value].copy()

    def debug_hdq(self, x, dist_th):
        d = self.device
        x = _f32(x.reshape(-1, 3), d)
        n = x.shape[0]
        o = dotdict(sdf_coarse=torch.empty(n, device=d), sdf_batch=torch.empty(n, 3, device=d),
                    nn_batch=torch.empty(n, 3, device=d, dtype=torch.int32), d2=torch.empty(n, 3, device=d),
                    bpts=torch.empty(n, 3, device=d), tpts=torch.empty(n, 3, device=d), mats=torch.empty(n, 24, device=d))
        cnt = C.c_int()
        check(self.lib.ra_debug_hdq(self.ctx, _ptr(x), n, dist_th, _ptr(o.sdf_coarse), _ptr(o.sdf_batch), _ptr(o.nn_batch), _ptr(o.d2),
                                    _ptr(o.bpts), _ptr(o.tpts), _ptr(o.mats), C.byref(cnt), self.stream), 'ra_debug_hdq')
        o.fine_count = int(cnt.value)
        return o
