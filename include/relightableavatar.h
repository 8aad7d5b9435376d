/*
 * relightableavatar.h — C ABI of the MI355X-native render hot path (librelightableavatar_hip.so).
 *
 * The reference (zju3dv/RelightableAvatar) is pure Python and has no FFI of its own; the drop-in
 * boundary is its plugin API: make_network / make_renderer / Renderer.render(batch)
 * (lib/networks/make_network.py:4-7, lib/networks/renderer/make_renderer.py:5-8).  The Python
 * classes in relightableavatar_amd/{networks,renderer} keep that API and bind these entry points
 * with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions:
 *  - plain pointers and sizes only; every float* is fp32; "dev" = device (HBM) pointer,
 *    "host" = host pointer.  stream is a hipStream_t passed as void* (torch's current stream);
 *    every launch goes on it, no hidden synchronisation except where stated.
 *  - every function returns 0 on success, non-zero on error; ra_last_error() gives the message
 *    (thread-local).  The library never calls abort().
 *  - all mutable state lives in the opaque ra_ctx (packed weights, frame state, scratch arenas,
 *    work counters).  One ctx per device; a ctx is not thread-safe, distinct ctxs are independent.
 *  - batch size B is 1 (base.yaml:74 test.batch_size 1); arrays are passed without the batch dim.
 */
#ifndef RELIGHTABLEAVATAR_H
#define RELIGHTABLEAVATAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RA_ABI_VERSION 8
#define RA_N_LIGHTS_MAX 512 /* env_h * env_w = 16 * 32 (lib/config/config.py:111-112) */

typedef struct ra_ctx ra_ctx;

const char* ra_last_error(void);
int ra_abi_version(void);

/* ---- lifetime -------------------------------------------------------------------------- */
int ra_ctx_create(ra_ctx** out, int device);
int ra_ctx_destroy(ra_ctx* ctx);

/* ---- constants the reference reads from its global cfg (lib/config/config.py) ------------ */
typedef struct ra_config {
    int   xyz_res, sdf_res, view_res;         /* 10, 8, 4   (configs/base.yaml:47-49)            */
    int   n_bones;                            /* 52  -> cond_dim = 3 * n_bones                    */
    int   relight;                            /* 1: RelightableAvatar heads, 0: AniSDF colour net */
    float resd_limit;                         /* 0.05 (config.py:224)                             */
    float blend_radius;                       /* 0.075 (config.py:191)                            */
    float albedo_slope, albedo_bias;          /* 1.0, 0.0  (config.py:407-408)                    */
    float roughness_slope, roughness_bias;    /* 0.9, 0.09 (config.py:409-410)                    */
    float fresnel_f0;                         /* 0.02 (config.py:89)                              */
    float shading_albedo;                     /* 0.8  (config.py:394)                             */
    float albedo_multiplier;                  /* 1.0                                              */
    int   lambert_only, glossy_only;          /* ablation switches (config.py:48-49)              */
    int   tonemapping;                        /* cfg.tonemapping_rendering                        */
    float bg_brightness;                      /* 0.0                                              */
    int   mlp_f16;                            /* element type of the fused MLP kernels: 1 = IEEE half, 0 = bfloat16 (same MFMA rate) */
    int   query_skip;                         /* 1 (default): rays that did not move since their last distance query / shadow rays whose
                                                 visibility reached 0 keep their last distance instead of being queried again — exact
                                                 (frames are bit-identical with 0), the reference re-queries them (sphere_tracing_renderer.py:144-205) */
    int   k4_batch_slots;                     /* full queries per forward+backward launch pair (bounds the activation tape: 4.9 KB per slot);
                                                 0 = default (1 Mi slots = 5 GB) */
    int   trace_precision;                    /* arithmetic of the distance queries INSIDE tracing loops.  The reference computes everything in
                                                 fp32; f16 MFMA operands leave the distance 5e-5 rms off, which flips the phase of the surface trace's
                                                 limit cycles on ~1 % of the pixels (sphere_tracing_renderer.py:176-197: hit test, sign-change
                                                 interpolation).  1 (default): the surface trace (16 iterations, 2 % of a relit frame's fine queries)
                                                 runs in COMPENSATED arithmetic — f16 hi + lo operand pairs, three MFMAs per k-step, fp32 accumulate:
                                                 1.3e-7 rms from a float64 evaluation, as good as fp32 itself — the shadow rays stay on plain 16-bit
                                                 operands; 0: plain operands everywhere (round 3's behaviour); 2: compensated everywhere, also
                                                 ra_hdq_sdf / ra_observed_sdf and the shadow rays (validation; 3x the MFMA work) */
    float clip_near, clip_far;                /* 0.02, 10.0: the volume renderer's near.clip(min=clip_near), far.clip(max=clip_far)
                                                 (base_renderer.py:120-121; config.py clip_near / clip_far), applied by ra_render_volume_chunk */
    int   only_visibility;                    /* cfg.only_visibility (sphere_tracing_renderer.py:516-519, 720-723; debugging option): the cosine
                                                 of every light is 1 and the probe's radiance the mean of its channels — in the shading of
                                                 ra_render_sphere_chunk and ra_render_ground_chunk (the three channels of `shade` are then equal:
                                                 the reference's map has one) */
    int   vis_shade_map;                      /* what the `shade` output of ra_render_sphere_chunk / ra_render_ground_chunk holds: 0 the shading
                                                 (:749-751), 1 the mean light visibility (cfg.vis_lvis_map, :537, :756), 2 the mean cosine
                                                 (cfg.vis_ldot_map, :538, :757; the reference applies it after vis_lvis_map, so it wins) */
    int   use_geodesic_filter;                /* 1 (default, cfg.use_geodesic_filter): geodesic_knn (sample_utils.py:103-162) — per-neighbour
                                                 signed distances, neighbours farther than dist_th from the closest one ON THE CANONICAL BODY
                                                 replaced by it; 0: knn_with_filter (:164-194) — distance sqrt(mean d^2) with the sign of
                                                 max_k sign((x - v_k) . n_k), the three neighbours as found */
    float key_light_share;                    /* 0.0078 (default: four times the mean share of 512 lights); with trace_precision 1: the
                                                 light-visibility rays towards the frame's KEY LIGHTS are traced in compensated arithmetic like the
                                                 surface trace.  A light is a key light when it holds at least this fraction of a probe's power
                                                 (radiance x solid angle) — and at least 4 / L — under any of the frame's probes; the 48 lights with
                                                 the largest share at most, which bounds the tier's cost at ~10 % of the shadow rays.  A DFSS penumbra value is d * sharp / (2 t) (sphere_tracing_renderer.py:157-179): it
                                                 amplifies the 5e-5 distance error of plain f16 operands up to 500 x per light.  Summed over 512 lights
                                                 of comparable power the errors average out; under a key light that holds most of the power they do
                                                 not (the reference-made hard cases of tests/golden/switches.npz: max |err| 1.2e-2 .. 4.4e-2).  The
                                                 key lights are 0-3 % of the lights (a sun, a window; none under an overcast sky).  0: no key-light tier (round 5's behaviour) */
} ra_config;
/* A zero-initialised ra_config is NOT the default configuration (trace_precision 0 = plain operands, clip_far 0, ...): start from
 * ra_default_config() — the values documented above — and override.  ra_set_config rejects trace_precision outside 0..2,
 * clip_far <= clip_near (or NaN), vis_shade_map outside 0..2 and a negative key_light_share. */
int ra_default_config(ra_config* out);
int ra_set_config(ra_ctx* ctx, const ra_config* cfg);

/* ---- weights: one call per state_dict entry (SURVEY.md section 8b "weights on disk") ------
 * name is the reference's state_dict key, data a HOST fp32 array of `numel` elements in
 * torch's row-major layout.  ra_finalize_weights folds weight_norm (W = g*v/|v|, net_utils.py:
 * 1326-1327), folds 1/sqrt(2) of the SDF skip (net_utils.py:1345-1346), pads to MFMA tiles,
 * converts to bf16 fragment order and uploads.  Replaces load_network()+nn.Module parameters
 * (lib/utils/net_utils.py:1514-1584). */
int ra_set_weight(ra_ctx* ctx, const char* name, const float* host_data, size_t numel);
int ra_finalize_weights(ra_ctx* ctx, void* stream);
/* 1 when the cooperative 4-wave distance kernel for launches of <= 8 Ki points (K3CC, csrc/ra_k3cc.hpp) passed its self-test against the
 * plain compensated kernel on this device at ra_finalize_weights (bit for bit; tested once per process and device) and is in use; 0: the
 * context launches K3C's 4-wave tiles instead (slower small launches, same results) — or has no weights yet. */
int ra_k3cc_enabled(const ra_ctx* ctx);

/* ---- per-frame SMPL state: the batch keys world_to_bigpose consumes
 * (lib/networks/deform/base_network.py:238-336; schema lib/datasets/base_dataset.py:337-397).
 * All dev pointers.  pverts, weights, A, big_A, poses and cond_fix are consumed by ra_set_frame (async on stream: vertex blend,
 * BVH build, bias folds); R, Th, pnorm and tverts are read IN PLACE by every later query of the frame, so the caller keeps them
 * alive and unchanged until the next ra_set_frame (the Python binding holds references).  cond_fix = train_motion.poses[:,
 * fix_material] (base_network.py:501-503), may be NULL for the relight network. */
typedef struct ra_frame {
    const float* R;        /* 3x3   */
    const float* Th;       /* 3     */
    const float* poses;    /* n_bones*3 (cond)            */
    const float* cond_fix; /* n_bones*3 or NULL            */
    const float* A;        /* n_bones x 4 x 4             */
    const float* big_A;    /* n_bones x 4 x 4             */
    const float* pverts;   /* n_verts x 3                 */
    const float* pnorm;    /* n_verts x 3                 */
    const float* tverts;   /* n_verts x 3                 */
    const float* weights;  /* n_verts x n_bones           */
    int n_verts;
} ra_frame;
int ra_set_frame(ra_ctx* ctx, const ra_frame* frame, void* stream);

/* ---- operators -------------------------------------------------------------------------- */
/* Network.inference_world_distance_field (base_network.py:365-387): hierarchical distance
 * query. x: n x 3 world points -> sdf: n. */
int ra_hdq_sdf(ra_ctx* ctx, const float* x_dev, int n, float dist_th, int smooth_transition,
               float* sdf_dev, void* stream);

/* Network.inference_observed_distance_field with filtering=False (base_network.py:389-449): x are big-pose points,
 * sdf = SDF(x + resd(x, pose)) -> sdf: n.  Runs the production distance-query kernel (K3) without the coarse level.  The
 * filtered variant is the hierarchical query above on a frame whose posed body IS the template (pverts = tverts, A = I). */
int ra_observed_sdf(ra_ctx* ctx, const float* bpts_dev, int n, float* sdf_dev, void* stream);

/* Network.world_to_bigpose_transform / bigpose_to_world_transform (base_network.py:338-363): per point the 4x4
 * w2b = big_A_bw @ affine_inverse(A_bw) @ affine_inverse([R | Th]) with the blended bone transforms of the point's 3 nearest
 * vertices of the ctx's CURRENT frame (no distance filtering); invert != 0 returns affine_inverse(w2b) instead.  R (9) and
 * Th (3) are device pointers to the world <- pose transform used in the composition: the backward variant searches the
 * template (caller sets a frame with pverts = tverts, R = I, Th = 0) but composes with the real frame's R, Th.
 * affine_inverse transposes the 3x3 block (blend_utils.py:11-15) also for blended, non-rigid matrices, as the reference does.
 * out: n x 16 row-major. */
int ra_bigpose_transform(ra_ctx* ctx, const float* x_dev, int n, const float* R_dev, const float* Th_dev, int invert,
                         float* out_dev, void* stream);

/* Network.forward in eval mode (relight_network.py:91-104 / base_network.py:496-515).
 * x, v: n x 3 (v may be NULL for the relight network); raw: n x ra_raw_channels(), zeros for
 * points farther than dist_th from the body.  Channels: relight [cpts3,bpts3,resd3,albedo3,
 * roughness1,norm3,occ1] = 17; AniSDF [cpts3,bpts3,resd3,norm3,rgb3,occ1] = 16. */
int ra_raw_channels(const ra_ctx* ctx);
int ra_forward(ra_ctx* ctx, const float* x_dev, const float* v_dev, int n, float dist_th,
               float* raw_dev, void* stream);

/* sphere_tracing (sphere_tracing_renderer.py:20-216, mode 'hdq') over n rays.
 * tan_i_dev: per-ray sharpness (soft shadow) or NULL -> scalar tan_i.  Outputs may be NULL. */
typedef struct ra_trace_params {
    int   iters;            /* 16 surface / 4 shadow                                   */
    float tan_i;            /* cfg.sphere_tracing.tan_i = 1000 (hard)                  */
    float tan_i_multiplier; /* 1                                                       */
    float relax, offset, eps;
    int   shadow_skip_iter; /* 1                                                       */
    int   clay_book;        /* !cfg.no_claybook                                        */
    int   soft_shadow;
    float dist_th;          /* HDQ threshold forwarded to the distance query           */
} ra_trace_params;
int ra_sphere_trace(ra_ctx* ctx, const float* ray_o, const float* ray_d, const float* near_,
                    const float* far_, const float* tan_i_dev, int n, const ra_trace_params* p,
                    float* surf, float* occ, float* st, float* ot, void* stream);

/* ---- renderers: one chunk of rays (chunkify, net_utils.py:291-359) ------------------------ */
typedef struct ra_render_out { /* all dev, all optional (NULL = not wanted); P rays        */
    float* rgb;        /* P x 3                                                         */
    float* acc;        /* P                                                             */
    float* depth;      /* P                                                             */
    float* surf;       /* P x 3                                                         */
    float* norm;       /* P x 3                                                         */
    float* albedo;     /* P x 3   (relight)                                             */
    float* roughness;  /* P       (relight)                                             */
    float* shade;      /* P x 3   (relight)                                             */
    float* spec;       /* P x 3   (relight, cfg.vis_specular_map)                        */
    float* cpts;       /* P x 3                                                         */
    float* bpts;       /* P x 3                                                         */
    float* resd;       /* P x 3                                                         */
    float* ray_o;      /* P x 3   origins of hit rays, zeros elsewhere                  */
    float* lvis;       /* P x 512 (cfg.vis_novel_light)                                 */
    float* ldot;       /* P x 512 (cfg.vis_novel_light)                                 */
    /* render_human's per-hit leftovers (sphere_tracing_renderer.py:616-650), as full-ray maps (zeros on misses, NOT premultiplied);
     * the caller compacts them to the hit rays */
    float* raw;               /* P x (n_samples * C): the network's raw channels of the n_samples surface samples (C = ra_raw_channels) */
    float* volume_albedo;     /* P x 3   composited albedo, clipped, before cfg.albedo_multiplier (relight)                 */
    float* volume_roughness;  /* P       composited roughness, clipped (relight)                                            */
} ra_render_out;

typedef struct ra_sphere_params {
    ra_trace_params surface;   /* cfg.sphere_tracing                                     */
    ra_trace_params shadow;    /* cfg.obj_lvis (iter 4, offset .01, dist_th .125)         */
    float shadow_near_offset;  /* cfg.obj_lvis.near_offset = 0.02                        */
    float dist_th;             /* cfg.dist_th for the material query                     */
    float surf_sample_range;   /* 0.005                                                  */
    int   n_samples;           /* 3                                                      */
    int   relighting;          /* cfg.relighting                                         */
    int   no_visibility, local_visibility;
    int   premultiply;         /* alpha_output_ (sphere_tracing_renderer.py:454-460,1113) */
    /* Several of the reference's render chunks in ONE call (the reference's chunks bound ITS memory; rays are independent): ray r of this
     * call belongs to chunk j with box_start[j] <= r < box_start[j + 1] and its shadow rays are clipped against boxes[6 j .. 6 j + 5] — the
     * box the reference's in-place growth (:1020-1022) had reached at that chunk, computed by the caller with the same float arithmetic.
     * Nothing else of render_human depends on the box, so the pixels are those of chunk-by-chunk rendering, bit for bit, and a frame of
     * several chunks runs ONE 16-iteration surface loop instead of one per chunk.  n_boxes <= 1: every ray uses `bbox`.  At most 32 boxes;
     * box_start[0] = 0, box_start[n_boxes] = P, ascending (host arrays). */
    int   n_boxes;
    const float* boxes;
    const int*   box_start;
} ra_sphere_params;

/* sphere_tracing_renderer.Renderer.get_pixel_value -> render_human (:551-784, :981-1039) for one
 * chunk.  bbox: 6 floats (min xyz, max xyz) = batch.wbounds AFTER the caller applied the
 * per-chunk margin growth (quirk: :1020-1022).  probe: env_h2 x env_w2 x 3 linear radiance
 * (the learned 32x64 map or a novel 16x32 probe); light_xyz/area/sharp: 512 lights. */
int ra_render_sphere_chunk(ra_ctx* ctx, const float* ray_o, const float* ray_d, const float* near_,
                           const float* far_, int P, const float* bbox_host6,
                           const float* probe_dev, int probe_h, int probe_w,
                           const ra_sphere_params* p, const ra_render_out* out, void* stream);

/* One rank's rays out of the frame's (pixels dealt in 8 x 8 tiles): out_*[i] = *[idx[i]] for the four per-ray arrays of a
 * batch (ray_o / ray_d: 3 floats per ray; near / far: 1), idx: n int64 ray indices (device).  No context: any device, any time.
 * Replaces the four index_select launches of the host framework in the per-frame path of a sharded job. */
int ra_gather_rays(int device, const long long* idx_dev, int n, const float* ray_o, const float* ray_d, const float* near_, const float* far_,
                   float* out_o, float* out_d, float* out_near, float* out_far, void* stream);

/* ... and the un-interleave after the frame all_gather: dst[dst_idx[i]] = src[src_idx[i]] for n rows of C floats (rank r, slot j -> the
 * j-th pixel owned by r; the index vectors are the shard plan's, ra_shard_plan). */
int ra_scatter_rows(int device, const float* src_dev, const long long* src_idx_dev, const long long* dst_idx_dev, long long n, int C,
                    float* dst_dev, void* stream);

/* Marks the start of one top-level render (Renderer.render: every chunk call of one frame follows).  The library numbers a frame's render
 * calls from ra_set_frame to match each with the counts the same call found in an earlier frame (launch-variant hints: a pure speed aid);
 * a caller that renders the SAME frame state again and again without ra_set_frame (a cached frame, a turntable of cameras) calls this
 * instead, or its calls run out of numbered slots after 64 and lose their hints.  Optional; never changes results. */
int ra_begin_render(ra_ctx* ctx);

/* The key lights of the current frame (ra_config.key_light_share), named by the caller: n probes of probe_h x probe_w x 3 (device) — every
 * probe the frame's cached visibility will be shaded with.  ra_render_sphere_chunk / ra_render_ground_chunk derive the key lights from the
 * probe they shade with; a renderer that traces once and re-shades under OTHER probes afterwards (novel_light_sphere_tracing.py:163-213:
 * ra_reshade, ra_reshade_ground) calls this before the frame's render calls, once per probe size (accumulate = 1 adds to the flags of the
 * call before), and with n = 0 after them (back to per-call key lights).  No counterpart in the reference, whose arithmetic is fp32 throughout. */
int ra_set_key_probes(ra_ctx* ctx, const float* probes_dev, int n, int probe_h, int probe_w, int accumulate, void* stream);

/* base_renderer.Renderer.get_pixel_value (base_renderer.py:53-113): uniform samples,
 * Network.forward per sample, alpha compositing.  near / far as the dataset delivers them: the renderer's clip
 * (base_renderer.py:120-121) is applied here (ra_config.clip_near / clip_far).  Every array of `out` is written for every ray. */
int ra_render_volume_chunk(ra_ctx* ctx, const float* ray_o, const float* ray_d, const float* near_,
                           const float* far_, int P, int n_samples, float dist_th,
                           const ra_render_out* out, void* stream);

/* novel_light_sphere_tracing.render_human (:21-66): re-shade cached maps under n_probes probes
 * in one pass.  probes: n_probes x h x w x 3. Outputs n_probes x P x 3 each (may be NULL). */
int ra_reshade(ra_ctx* ctx, const float* ray_o, const float* surf, const float* norm,
               const float* albedo, const float* roughness, const float* lvis, const float* ldot,
               int P, const float* probes_dev, int n_probes, int probe_h, int probe_w,
               float* rgb, float* shade, float* spec, void* stream);

/* novel_light_sphere_tracing.render_ground (:70-99): re-shade the ground layer of the main pass under n_probes probes from its
 * cached per-light visibility and cosine (ra_ground_out.lvis / .ldot of ALL frame pixels, P x 512 each): Lambert ground,
 * rgb = linear2srgb(albedo / pi * sum_l lvis ldot area L_probe(l)), shade = sum / pi, spec = shade / 20 (no shading_albedo, no
 * ground_shading_multiplier here).  albedo: attach_envmap != 0 -> the probe's image (images: n_probes x ih x iw x 3, or NULL ->
 * the probe itself) sampled along ray_d (:79-83), else albedo_map (P x 3).  Outputs n_probes x P x 3 each, any may be NULL. */
int ra_reshade_ground(ra_ctx* ctx, const float* ray_d, const float* albedo_map, const float* lvis, const float* ldot, int P,
                      const float* probes_dev, int n_probes, int probe_h, int probe_w, const float* images_dev, int image_h, int image_w,
                      int attach_envmap, float* rgb, float* albedo, float* shade, float* spec, void* stream);

/* ---- measurement ------------------------------------------------------------------------- */
typedef struct ra_counters {       /* cumulative since ra_reset_counters; read with a sync  */
    uint64_t n_coarse;             /* 3-NN queries (K1)                                       */
    uint64_t n_fine_sdf;           /* resd+SDF forward, sdf only  (F_sdf  = 1 901 568 FLOP)   */
    uint64_t n_fine_full;          /* resd+SDF forward+tangents+heads (geometry points)       */
    uint64_t n_shadow_rays;
    uint64_t n_hit_pixels;
    uint64_t n_shaded;             /* pixel x probe shading evaluations                       */
    uint64_t n_fine_sdf_wide;      /* the part of n_fine_sdf computed by the 8-wave distance kernel (launches that fill the chip) */
    uint64_t n_fine_sdf_comp;      /* the part of n_fine_sdf computed in compensated arithmetic (ra_config.trace_precision): 3 x F_sdf executed */
} ra_counters;
int ra_get_counters(ra_ctx* ctx, ra_counters* out, void* stream); /* synchronises stream */
int ra_reset_counters(ra_ctx* ctx, void* stream);

/* time (ms) spent in the fused MLP kernel launches since the last reset, measured with HIP
 * events on `stream`; n_launches receives the launch count.  Synchronises. */
int ra_get_mlp_time(ra_ctx* ctx, float* ms, int* n_launches, void* stream);
/* the same for one kernel family: kind 0 = fused distance query (K3, every width), 1 = full query with normals / material / colour (K4),
 * 2 = the 8-wave distance query only (the launches that fill the chip: the frame's dominant kernel), 3 = the 2- / 4-wave distance query,
 * 4 = the compensated distance query (K3C and its cooperative small-launch variant K3CC; not part of kind 0) */
int ra_get_kernel_time(ra_ctx* ctx, int kind, float* ms, int* n_launches, void* stream);
int ra_enable_timing(ra_ctx* ctx, int on);
/* ---- frames in flight --------------------------------------------------------------------------------------------
 * Throughput aid with no counterpart in the reference (its frame loop is sequential: lib/evaluators, run.py).  Several contexts on
 * several HIP streams render DIFFERENT frames at once; contexts that share a gate run their light-visibility stages (the frame's
 * large distance-query launches, sphere_tracing_renderer.py:265-344) one after the other in submission order, while everything
 * else of the next frame (pose, box structure, the 16 latency-bound surface-tracing iterations, normals, shading) runs beside the
 * previous frame's stage.  Frames are bit-identical with and without.  A gate is used from ONE host thread; it must outlive the
 * contexts attached to it (detach with gate = NULL). */
typedef struct ra_gate ra_gate;
int ra_gate_create(ra_gate** out, int device);
int ra_gate_destroy(ra_gate* gate);
int ra_set_gate(ra_ctx* ctx, ra_gate* gate);

/* ---- multi-GPU: the deal of a frame's in-box rays to the ranks of one node (no counterpart in the reference,
 * whose inference is single-process) ----------------------------------------------------------------------------------------------
 * HOST function, no device work, no ctx: relightableavatar_amd/shard.py make_plan's per-frame part in one pass over the mask.
 * mask: H x W bytes (batch.mask_at_box, row-major, non-zero = in box); P = number of in-box rays (= set bytes, checked); the in-box rays
 * are the mask's pixels in row-major order (lib/utils/data_utils.py:925-938).  Pixels are dealt in 8 x 8 tiles: ground = 0 -> the tiles
 * that hold in-box pixels round robin in raster order; ground != 0 -> fixed diagonal stripes (tile_y + tile_x) % world over the whole
 * frame (the sharded ground-plane pass needs a rank's in-box pixels to be a subset of its frame pixels).
 * Outputs (host, caller-allocated, e.g. inside one pinned staging block): owner[P] (nullable) rank of every ray; counts[world];
 * *n_max = max(counts); order[P] = rays grouped by owner, ascending inside (rank r's shard is order[offs[r] : offs[r + 1]]);
 * src[P]: item j of `order` is row src[j] = rank * n_max + position of the all_gather's (world * n_max)-row output, i.e.
 * full[order] = gathered[src]; inds[P] (nullable; needs ground_pos[H * W] = position of every frame pixel in its owner's full-frame
 * pixel list): ground_pos of the rays in `order` order.  edges[n_edges] (nullable): ascending ray indices (chunk boundaries of the
 * unsharded ray list, chunkify's rule net_utils.py:323); chunk_pos[r * n_edges + e] = how many of rank r's rays lie before ray
 * edges[e].  world <= 256. */
int ra_shard_plan(const unsigned char* mask, int H, int W, int world, int ground, long long P, const long long* ground_pos,
                  const long long* edges, int n_edges, unsigned char* owner, long long* order, long long* src, long long* inds,
                  long long* counts, long long* chunk_pos, long long* n_max);

/* 1 (default): exact 3-NN through the per-frame vertex BVH; 0: brute force over all vertices (validation path).
 * Takes effect at the next ra_set_frame. Both return identical neighbours. */
int ra_set_knn_mode(ra_ctx* ctx, int use_bvh);

/* ---- N1: ground-plane pass --------------------------------------------------------------------
 * replaces render_ground (lib/networks/renderer/sphere_tracing_renderer.py:463-548) for one chunk of full-frame rays:
 * ray/plane hit (moller_trumbore on one triangle of the plane, mesh_utils.py:710-738), DFSS shadows of the avatar onto
 * the plane (light_visibility :265-344 with cfg.env_lvis), Lambert ground lit by the probe, distance fade (:497-505). */
typedef struct ra_ground_params {
    float normal[3];            /* cfg.ground_normal (normalised inside) */
    float origin[3];            /* cfg.ground_origin */
    float albedo[3];            /* cfg.ground_albedo, used when attach_envmap == 0 */
    int attach_envmap;          /* cfg.ground_attach_envmap: albedo = probe sampled along the view ray */
    float env_r;                /* cfg.env_r */
    float shading_multiplier;   /* cfg.ground_shading_multiplier */
    ra_trace_params shadow;     /* cfg.env_lvis as a trace-parameter block */
    float shadow_near_offset;   /* cfg.env_lvis.near_offset */
    int no_visibility, local_visibility;
    /* Several of the reference's render chunks in ONE call (their launches merged; pixels identical): pixel r of the call belongs to chunk j
     * with box_start[j] <= r < box_start[j + 1] and its shadow rays are clipped against boxes[6 j .. 6 j + 5], the box the reference's
     * in-place growth (sphere_tracing_renderer.py:1054-1056) had reached at that chunk — computed by the caller, in the reference's float
     * arithmetic.  n_boxes = 0 or 1: the call is one chunk and uses the bbox argument.  At most 32 boxes per call. */
    int n_boxes;
    const float* boxes;         /* host, n_boxes x 6 */
    const int* box_start;       /* host, n_boxes + 1 ascending pixel indices, box_start[0] = 0, box_start[n_boxes] = P */
} ra_ground_params;

typedef struct ra_ground_out {  /* device buffers with P rows, any may be NULL */
    void* rgb;      /* (P,3) */
    void* surf;     /* (P,3) */
    void* albedo;   /* (P,3) */
    void* shade;    /* (P,3) shade_map (multiplier applied) */
    void* spec;     /* (P,3) spec_map = shade / 20 */
    void* depth;    /* (P)   t clipped to +-env_r */
    void* lvis;     /* (P,512) visibility after the distance fade (render_ground :505), cfg.vis_novel_light (:541-543) */
    void* ldot;     /* (P,512) ground normal . light direction, not clamped (:504) */
} ra_ground_out;

/* ray_o, ray_d: (P,3); acc: (P) = 1 - human acc (pixels with acc <= 0 are not traced and come back as zeros);
 * bbox: 6 host floats (the human box grown by the caller, get_ground_value :1054-1056); probe: (ph,pw,3) device. */
int ra_render_ground_chunk(ra_ctx* ctx, const float* ray_o, const float* ray_d, const float* acc, int P, const float* bbox,
                           const float* probe, int ph, int pw, const ra_ground_params* p, const ra_ground_out* out, void* stream);

/* blend_output_ / alpha_blend (sphere_tracing_renderer.py:396-451) for one map of C channels:
 *   dst[f] = ground[f] * acc[f]                      for all F full-frame pixels (ground == NULL: zeros, the acc_map case)
 *   dst[inds[j]] += human[j] * (1 - acc[inds[j]])    for the P in-box rays (human == NULL: skipped, the ground-only keys)
 * acc: (F) = 1 - human acc scattered to the frame; inds: (P) int64 frame index of every in-box ray. */
int ra_blend_ground(ra_ctx* ctx, const float* ground, const float* human, const long long* inds, const float* acc, int F, int P, int C,
                    float* dst, void* stream);

/* ---- N2: ray generation + bounding-box culling on the device ------------------------------------
 * replaces lib/utils/data_utils.py:827-845 (get_rays), :860-875 (get_full_near_far), :925-938 (get_rays_within_bounds),
 * called per frame by lib/datasets/pose_dataset.py:53-68 on the CPU.
 * K, R: 9 doubles row-major, T: 3 doubles (host); bounds: 6 floats (host: min xyz, max xyz = batch.wbounds).
 * Device outputs with capacity H*W rays: ray_o, ray_d (n,3) f32; near, far (n) f32 — the in-box rays in row-major
 * pixel order, exactly the reference's boolean-mask order; mask_at_box: H*W uint8.  *n_rays receives the count
 * (synchronises the stream); n_rays = NULL: no read-back, no synchronisation (the caller knows the count, e.g. H*W for an unbounded box).
 * bounds_dev (6 device floats, takes precedence over the host `bounds`, which may then be NULL): the box of a frame whose body state
 * is still being computed on the stream (ra_pose_frame's wbounds) — no host round trip between N3 and N2.  n_rays_dev (device int,
 * nullable) receives the count too: an animation loop copies it to pinned memory behind an event and reads it a pipeline turn later
 * (relightableavatar_amd/engine.py gen_rays_async), so generating the rays of frame f + 1 never waits for frame f.  Directions are computed in fp64 and rounded once (the reference computes them in the
 * camera's dtype and casts to float32); near/far follow the reference's float32 arithmetic operation by operation. */
int ra_gen_rays(ra_ctx* ctx, int H, int W, const double* K, const double* R, const double* T, const float* bounds, const float* bounds_dev,
                void* ray_o, void* ray_d, void* near, void* far, void* mask_at_box, int* n_rays, int* n_rays_dev, void* stream);

/* ---- N3: per-frame body state on the device -----------------------------------------------------
 * replaces lib/datasets/base_dataset.py:308-397 (get_lbs_params with cfg.use_geometry, get_blend), which runs on the CPU
 * per frame: bone transforms from the axis-angle poses (net_utils.py:1164-1183 / data_utils.py:1004-1069), template ->
 * T pose -> posed -> world vertices (blend_utils.py:212-218,264-313), vertex normals (pytorch3d Meshes.verts_normals),
 * bounds (data_utils.py:616-622).
 * Host inputs: poses, tjoints (J,3) f32, parents (J) int32 (topological order, parents[0] unused), big_A (J,16) f32,
 * Rh, Th (3) f32, faces (F,3) int32 (its vertex -> corner list is cached by content hash).
 * Device inputs: tverts (N,3), weights (N,J).  Device outputs (any may be NULL): A (J,16), joints (J,3), tpose (N,3),
 * pverts (N,3), wverts (N,3), pnorm (N,3), R (9), pbounds (6), wbounds (6).
 * ASYNCHRONOUS: the host inputs are copied into a pinned staging ring of the context before the call returns (the caller may reuse its
 * arrays at once), uploaded with one asynchronous copy, and the bone transforms (Rodrigues + the chain of 4 x 4 products, float64) run
 * on the device; nothing waits for the stream except the first call with a new `faces` array (its vertex -> corner list is built on the
 * host and uploaded synchronously, once per mesh). */
typedef struct ra_pose_in {
    const float *poses, *tjoints, *big_A, *Rh, *Th;
    const int* parents;
    const int* faces;
    int n_bones, n_faces, n_verts;
    const void *tverts, *weights;
    float bounds_padding;           /* get_bounds(padding=0.05) */
} ra_pose_in;
typedef struct ra_pose_out {
    void *A, *joints, *tpose, *pverts, *wverts, *pnorm, *R, *pbounds, *wbounds;
    void *poses, *Th;               /* device copies of the uploaded poses (J,3) and Th (3): the batch keys `poses` / `Th` without a second upload */
} ra_pose_out;
int ra_pose_frame(ra_ctx* ctx, const ra_pose_in* in, const ra_pose_out* out, void* stream);

/* The reference grows batch.wbounds IN PLACE by cfg.env_lvis.bbox_margin once per render chunk (sphere_tracing_renderer.py:1020-1022,
 * :1054-1056: bbox[:, 0] -= m; bbox[:, 1] += m).  wbounds: the batch's device tensor, 2 x 3 floats (min | max); one launch. */
int ra_grow_bounds(ra_ctx* ctx, float* wbounds_dev, float margin, void* stream);

/* ---- N4: environment-map rotation and the light-probe inset -------------------------------------
 * ra_shift_envmap: rotate_envmap's shift_image (lib/utils/relight_utils.py:69-85): out[y][x] = bilinear sample of img at
 * x + 0.5 + shift (wrapped modulo W; grid_sample align_corners=False, border padding), img/out: (H,W,C) device fp32.
 * ra_add_light_probe: add_light_probe (relight_utils.py:38-54 with gen_light_dir :9-35): overwrites the top-left uH x uW
 * pixels of rgb (H,W,3) with the probe seen along the camera's horizontal heading; cam_R: 9 host floats (world-to-camera). */
int ra_shift_envmap(ra_ctx* ctx, const float* img, int H, int W, int C, float shift, float* out, void* stream);
int ra_add_light_probe(ra_ctx* ctx, float* rgb, int H, int W, const float* probe, int ph, int pw, const float* cam_R, int uH, int uW,
                       void* stream);

/* ---- N4, third item: map -> image normalisations of the reference visualiser -------------------------------------
 * Visualizer.generate_image (lib/visualizers/base_visualizer.py:54-201) for one output type (lib/config/config.py:364-378): the
 * per-type normalisation of the rendered map, then the scatter of the P in-box rays into the H x W image over bg_brightness
 * (:188-195) and, if alpha != NULL, the alpha plane (:201-208; the caller concatenates it after the light-probe inset).
 *   SURFACE   a = cpts_map | surf_map (P,3): (a - tbounds[0]) / (tbounds[1] - tbounds[0]) * acc
 *   RESIDUAL  a = cpts_map, b = bpts_map:     acc * (a - b) / (the int(0.005 * 3P)-th largest value of a - b)
 *   DEPTH     a = depth_map (P):              clip((a - lo) / (hi - lo), 0, 1), lo / hi = the int(0.01 * P)-th smallest / largest
 *                                             depth among rays with acc != 0, lo clipped to min_clip (fewer such rays than the
 *                                             rank: the rank is clamped to their count, where the reference's topk raises)
 *   ALPHA     acc;   ROUGHNESS a (P);   RENDERING a (P,3);   ALBEDO a (P,3), linear2srgb if tonemap
 *   NORMAL    a = norm_map (P,3):             (normalize(a) @ cam_R^T, y and z flipped) * 0.5 + 0.5, times acc
 *   SHADING / SPECULAR a (P,3):               if normalize: a / (the int(0.005 * 3P)-th largest value)
 * pix: frame pixel of every ray (NULL: the maps are full-frame, P == H*W).  image: H*W x 3. */
enum { RA_IMG_SURFACE = 3, RA_IMG_RESIDUAL = 4, RA_IMG_DEPTH = 5, RA_IMG_ALPHA = 6, RA_IMG_NORMAL = 7, RA_IMG_SPECULAR = 8,
       RA_IMG_ALBEDO = 9, RA_IMG_ROUGHNESS = 10, RA_IMG_SHADING = 11, RA_IMG_RENDERING = 12 };      /* values of the Output enum */
typedef struct ra_image_params {
    int type, H, W;
    float bg_brightness;    /* cfg.bg_brightness */
    int normalize;          /* cfg.normalize_shading / cfg.normalize_specular */
    int tonemap;            /* cfg.tonemapping_albedo */
    float min_clip;         /* cfg.min_clip */
    float cam_R[9];         /* batch.cam_R (world -> camera), NORMAL */
    float tbounds[6];       /* batch.tbounds (big-pose box), SURFACE */
} ra_image_params;
int ra_map_to_image(ra_ctx* ctx, const ra_image_params* p, const float* a_dev, const float* b_dev, const float* acc_dev,
                    const long long* pix_dev, int P, float* image_dev, float* alpha_dev, void* stream);

/* ---- test hooks: stage outputs for the parity tests (tests/test_gpu_*.py); not used by renderers ---- */
/* resd + sdf MLPs on given big-pose points: resd n x 3, sdf n, feat n x 256 (any may be NULL) */
/* the current frame's key lights (ra_config.key_light_share): n_lights flags and every light's largest share of a probe's power */
int ra_debug_key_lights(ra_ctx* ctx, unsigned char* key_dev, float* share_dev, void* stream);
int ra_debug_mlp(ra_ctx* ctx, const float* bpts_dev, int n, float* resd, float* sdf, float* feat, void* stream);
/* full kernel with identity warp: d sdf/d bpts n x 3, sdf n, feat n x 256, raw n x C */
int ra_debug_full(ra_ctx* ctx, const float* bpts_dev, int n, float* grad, float* sdf, float* feat, float* raw, void* stream);
/* the vertex ids of the current frame's box structure in leaf order (32 per leaf; the padding of the last leaf is 0x7fffffff) to a HOST array
 * of `capacity` ints; *n_out = 32 x leaves (0 without a structure: brute-force mode or a mesh beyond its limit).  Synchronises the stream. */
int ra_debug_bvh_ids(ra_ctx* ctx, int* ids_host, int capacity, int* n_out, void* stream);
/* coarse level: per-point coarse sdf n, sdf_batch n x 3, nn_batch n x 3 (int32), filtered d2 n x 3, and for fine points
 * bpts/tpts n x 3, blended (A|big_A) rows n x 24 (zeros elsewhere); fine_count_host receives the count (synchronises) */
int ra_debug_hdq(ra_ctx* ctx, const float* x_dev, int n, float dist_th, float* sdf_coarse, float* sdf_batch, int* nn_batch,
                 float* d2, float* bpts, float* tpts, float* mats, int* fine_count_host, void* stream);

/* get_near_far_aabb (net_utils.py:1683-1712, return_raw) on n rays: the slab test the shadow-ray generator runs inline */
int ra_debug_aabb(ra_ctx* ctx, const float* o_dev, const float* d_dev, int n, const float* bbox_host6, float* near_dev, float* far_dev, void* stream);
/* light_visibility (sphere_tracing_renderer.py:265-344) for n surface points: surf, norm n x 3, acc n -> lvis, ldot n x 512 */
int ra_debug_lvis(ra_ctx* ctx, const float* surf_dev, const float* norm_dev, const float* acc_dev, int n, const float* bbox_host6,
                  const ra_trace_params* shadow, float near_offset, float* lvis_dev, float* ldot_dev, void* stream);
/* Microfacet.__call__ (relight_utils.py:484-577): pts2l L x N x 3, pts2c / normal / albedo N x 3, rough N -> brdf L x N x 3 */
int ra_debug_brdf(ra_ctx* ctx, const float* p2l_dev, const float* p2c_dev, const float* normal_dev, const float* albedo_dev,
                  const float* rough_dev, int L, int N, float* brdf_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RELIGHTABLEAVATAR_H */
