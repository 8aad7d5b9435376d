// Launchers of the non-MLP kernels (ra_hdq.hip, ra_trace.hip).
#pragma once
#include "ra_common.hpp"

// where the query points of a pass come from
struct RaySet {
    int mode;             // 0: x[i];  1: o[i] + t[i] d[i];  2: o[pix[i]] + t[i] ldir[light[i]]
    const float* x;       // mode 0: n x 3
    const float* o;       // mode 1: n x 3; mode 2: n_pix x 3
    const float* d;       // mode 1: n x 3
    const float* t;       // modes 1,2: n
    const int* pix;       // mode 2
    const int* light;     // mode 2
    const float* ldir;    // mode 2: n_lights x 3 (unit)
    const int* n_dev;     // optional device-side count (<= n launched); nullptr -> n
    const unsigned char* skip;   // optional: ray i did not move since its last query -> its sdf[i] is still valid, do not re-query
    // optional: the three nearest vertices every query of this pass finds are written here (3 ints per ray), and — with hint_valid — read
    // first: the neighbours a ray found one tracing iteration ago give its bounds at the new point at once (3 distance evaluations instead
    // of a seed search + a 32-candidate leaf scan dense in insertions, in EVERY wave of a split workgroup); any vertex is a valid
    // candidate, so the result stays the exact 3-NN
    int* nn_hint;
    int hint_valid;
    // optional, for a loop's FIRST pass: start from another pass's neighbours — query i reads the triplet hint_src[3 * hint_src_index[i]]
    // (the shadow rays of a hit pixel start next to the point whose neighbours the surface trace found last).  Takes precedence over
    // nn_hint as the source; nn_hint is still written.
    const int* hint_src;
    const int* hint_src_index;
};

struct HdqOut {
    float* sdf;           // n: coarse signed distance (smpl_sdf after the abs rule)
    int* fine_count;      // device counter, ZERO on entry (a fresh one per pass: ra_api.cpp next_fine_counter)
    int* fine_idx;        // slot -> point
    float* bpts;          // slot x 3
    float* mats;          // slot x 24 or nullptr
    // optional second fine list (the key-light tier, ra_config.key_light_share): fine points of rays towards a light with key[light] != 0
    // (RaySet mode 2) are compacted into fine_idx2 / bpts2 through fine_count2 instead — the list the compensated kernel answers
    const unsigned char* key;   // nullable: n_lights flags
    int* fine_count2;
    int* fine_idx2;
    float* bpts2;
    float* raw_zero;      // nullable: n x raw_C rows that non-fine points zero (Network.forward's zeros outside dist_th)
    int raw_C;
    // debug (nullable): per point
    float* dbg_sdf_batch; // n x 3
    int* dbg_nn_batch;    // n x 3
    float* dbg_d2;        // n x 3 (filtered)
    float* dbg_tpts;      // slot-ordered? no: per point n x 3 (only fine points written)
    float* dbg_bpts;      // n x 3
    float* dbg_mats;      // n x 24
    DevCounters* counters; // nullable
};

void launch_vert_blend(const float* weights, const float* A, const float* big_A, int n_verts, int n_bones, float* vertA,
                       hipStream_t s);
void launch_pack_verts(const float* pverts, int n_verts, float4* out, hipStream_t s);
void launch_fold_bias(const float* W, int ld, int col0, int ncond, const float* cond, const float* bias, float* out,
                      hipStream_t s);   // out[r] = bias[r] + sum_c W[r*ld + col0 + c] * cond[c], r < 256
// per-frame build of the vertex box structure (bbox, Morton codes, ranks, boxes: two launches); n_verts <= 16384
// order: scratch of n_verts ints (the vertices' Morton order, recomputed every frame)
void launch_bvh_build(const float4* pverts4, int n_verts, int* order, float* leaves, float4* sbox, int n_leaves, int n_supers, hipStream_t s);
int bvh_leaf_count(int n_verts);
int bvh_super_count(int n_leaves);
void launch_hdq_coarse(const FrameState& fr, const RaySet& rs, int n, float th, float blend_radius, const HdqOut& out,
                       hipStream_t s, bool geodesic = true);

struct TraceState {       // SoA per ray
    float *t, *d0, *dt, *st, *ot, *cd, *occ, *off, *rlx;
    const float *near_, *far_;
    const float* tan_i;   // per ray (mode 1) or per light (mode 2) or nullptr -> scalar
    const int* light;     // mode 2
    unsigned char* stuck; // ray did not move in the last update (t clamped at far / near): the next query would repeat the last one
};
void launch_trace_init(const TraceState& ts, int n, const int* n_dev, const ra_trace_params& p, hipStream_t s);

void launch_trace_update(const TraceState& ts, const float* sdf, int n, const int* n_dev, int iter,
                         const ra_trace_params& p, hipStream_t s);

// surface pass epilogue: surf/depth/acc, hit compaction (hit_count: zeroed here unless counter_is_zero), slot_of_ray[r] = -1 for every ray
void launch_surface_finish(const float* ray_o, const float* ray_d, const float* st, const float* occ, int P, float* surf,
                           float* depth, float* acc, int* hit_idx, int* hit_count, hipStream_t s, int* slot_of_ray = nullptr,
                           bool counter_is_zero = false);
// slot_of_ray[hit_idx[k]] = k for the (possibly re-ordered) hit list
void launch_slot_index(const int* hit_idx, const int* hit_count, int P, int* slot_of_ray, hipStream_t s);
// every requested output map of a chunk in ONE launch: per ray either its hit slot's values (optionally times acc) or zeros
struct MapJob { const float* src; float* dst; int C; int premul; int src_full; };
constexpr int RA_MAX_MAP_JOBS = 16;
struct EmitMaps { MapJob job[RA_MAX_MAP_JOBS]; long long end[RA_MAX_MAP_JOBS]; int n_jobs; const int* slot_of_ray; const float* acc; const int* perm; int P; };
void launch_emit_maps(const EmitMaps& m, hipStream_t s);
// 3 (S) samples per hit pixel: x = surf + z * view
void launch_surface_samples(const float* surf, const float* ray_d, const int* hit_idx, const int* hit_count, int P, int S,
                            float range, float* x, float* v, int* n_out, hipStream_t s);
// per hit pixel: composite S samples of raw (C channels, last = occ), normalise, split
struct SurfaceMaps {      // per hit slot
    float *cpts, *bpts, *resd, *norm, *albedo, *rough, *rgb;
    float* valbedo;       // nullable: clipped albedo before cfg.albedo_multiplier (ret.volume_albedo, :646)
};
void launch_surface_composite(const float* raw, int C, int S, const int* hit_count, int P, int relight, const ra_config& cfg,
                              const SurfaceMaps& m, hipStream_t s);

// shadow rays
constexpr int RA_MAX_BOXES = 32;
struct ShadowGen {
    const float* surf;    // P x 3 (full ray indexing)
    const float* norm;    // hit-slot x 3
    const float* acc;     // P
    const int* hit_idx;   // slot -> ray
    const int* hit_count;
    const float* ldir;    // L x 3 unit light directions
    float bbox[6];
    // several render chunks of the reference in ONE launch (the ground pass): ray r belongs to chunk j with box_start[j] <= r < box_start[j + 1]
    // and is clipped against boxes[j] — the box the reference's in-place growth (sphere_tracing_renderer.py:1054-1056) had reached at that
    // chunk, computed by the caller with the same float arithmetic.  n_boxes <= 1: every ray uses bbox.
    int n_boxes;
    float boxes[RA_MAX_BOXES][6];
    int box_start[RA_MAX_BOXES + 1];
    const int* perm;      // nullable: the rays were re-ordered (Morton sort) — ray r is the caller's ray perm[r], which is what box_start counts
    float near_offset;
    int L;
    int no_visibility, local_visibility;
    int split_wide_groups;   // a group of 64 hit slots wider than 15 cm emits its rays half by half (the human layer; see shadow_gen_kernel)
    // out
    float* lvis;          // slot x L
    float* ldot;          // slot x L
    int* ray_pix;         // ray -> full ray index (origin = surf[pix])
    int* ray_light;
    int* ray_slot;        // ray -> slot*L + light (where occ lands)
    float *near_, *far_;
    int* ray_count;
};
// key[l] = light l holds at least the fraction max(share, 4 / L) of a probe's power under any of the frame's probes — the kmax lights with
// the largest such share at most.  smax (L floats): every light's largest share so far; accumulate: the n probes join those of earlier calls
void launch_key_lights(const float* probes, int n, int ph, int pw, const float* ldir, const float* area, int L, float share, int kmax,
                       int accumulate, float* smax, unsigned char* key, hipStream_t s);
void launch_gather_shard_rays(const long long* idx, int n, const float* ro, const float* rd, const float* nr, const float* fr, float* so, float* sd,
                              float* sn, float* sf, hipStream_t s);
void launch_scatter_rows(const float* src, const long long* src_idx, const long long* dst_idx, long long n, int C, float* dst, hipStream_t s);
void launch_shadow_gen(const ShadowGen& g, int P, hipStream_t s, bool counter_is_zero = false);
void launch_debug_aabb(const float* o, const float* d, int n, const float* bbox6, float* nr, float* fr, hipStream_t s);
void launch_debug_brdf(const float* p2l, const float* p2c, const float* nrm, const float* alb, const float* rough, int L, int N, const ra_config& cfg,
                       float* out, hipStream_t s);
// reorder the hit list so that 64 consecutive hit pixels have neighbouring surface points (radix sort by Morton key)
size_t sort_hits_temp_bytes(int P);
int launch_sort_hits(const float* surf, const float* acc, int P, const float* bbox_min, unsigned* keys_in, unsigned* keys_out,
                     int* vals_in, int* hit_idx_out, void* temp, size_t temp_bytes, hipStream_t s);
void launch_shadow_scatter(const float* occ, const int* ray_slot, const int* ray_count, int max_rays, float* lvis, hipStream_t s);

struct ShadeIn {
    const float* ray_o;   // indexed by ray (idx != null) or by slot
    const float* surf;    // same indexing
    const int* idx;       // slot -> ray or nullptr (identity)
    const int* count;     // device count or nullptr -> n
    int n;
    const float *norm, *albedo, *rough;   // per slot
    const float *lvis, *ldot;             // slot x L
    const float *light_xyz, *light_area;  // L x 3, L
    int L;
    const float* probes;  // n_probes x h x w x 3
    int n_probes, ph, pw;
    int want_spec;
    float *rgb, *shade, *spec;            // n_probes x n x 3 (slot indexed)
};
void launch_shade(const ShadeIn& in, const ra_config& cfg, hipStream_t s);

// scatter hit-slot maps into full-ray outputs (zeros elsewhere), optional premultiplication by acc
// src_full: src is indexed by ray (like dst) instead of by hit slot
void launch_scatter_maps(const int* hit_idx, const int* hit_count, int P, int premultiply, const float* acc_full,
                         const float* src, int C, float* dst, int src_full, const int* perm, hipStream_t s);
// Morton-sort the primary rays of a chunk by their entry point and gather them into so/sd/sn/sf; perm[i] = caller index of sorted ray i
int launch_sort_rays(const float* ro, const float* rd, const float* nr, const float* fr, int P, const float* bbox_min, unsigned* keys_in,
                     unsigned* keys_out, int* vals_in, int* perm, void* temp, size_t temp_bytes, float* so, float* sd, float* sn, float* sf,
                     hipStream_t s, float near_min = -3.0e38f, float far_max = 3.0e38f);
void launch_accumulate(const int* count, unsigned long long* dst, hipStream_t s);

// volume path
void launch_volume_samples(const float* ray_o, const float* ray_d, const float* near_, const float* far_, int P, int S,
                           float* x, float* v, hipStream_t s);
void launch_volume_composite(const float* raw, int C, const float* near_, const float* far_, int P, int S, float bg,
                             const ra_render_out& out, const int* perm, hipStream_t s);
void launch_fill(float* p, size_t n, float v, hipStream_t s);
void launch_iota(int* idx, int n, int* count, hipStream_t s);      // idx[i] = i, *count = n
// per point: w2b = big_A_bw @ affine_inverse(A_bw) @ affine_inverse([R|Th]) (or its affine_inverse) from the blended rows of the coarse level
void launch_bigpose_compose(const float* mats, const float* d2, int n, float blend_radius, const float* R, const float* Th, int invert, float* out,
                            hipStream_t s);
void launch_light_dirs(const float* xyz, int L, float* ldir, hipStream_t s);

// N2: ray generation + AABB culling (ra_trace.hip)
struct RayCam { double Kinv[9]; double R[9]; double T[3]; double o[3]; float bmin[3]; float bmax[3]; int H, W; const float* bdev; };   // bdev: the box as 6 device floats (takes precedence)
size_t gen_rays_temp_bytes(int n_pixels);
int launch_gen_rays(const RayCam& cam, unsigned char* mask, int* pix_idx, int* count_dev, void* temp, size_t temp_bytes,
                    float* ray_o, float* ray_d, float* near, float* far, hipStream_t s);

// N1: ground-plane pass (ra_trace.hip)
struct GroundIn {
    const float *ray_o, *ray_d, *acc;     // P
    int P;
    float n[3], orig[3], albedo[3];
    int attach_envmap;
    float env_r, shading_multiplier;
};
// t, surf, clipped depth for every pixel; hit list = pixels with acc > 0; per-slot normal array filled with n
void launch_ground_hit(const GroundIn& g, float* t, float* surf, float* depth, float* norm_slots, int* hit_idx, int* hit_count, hipStream_t s);
struct GroundShade {
    GroundIn g;
    const float* t; const float* surf;
    const int* hit_idx; const int* hit_count;
    const float* lvis;                    // slot x L
    const float *ldir, *light_area; int L;
    const float* probe; int ph, pw;
    float *rgb, *albedo, *shade, *spec;   // P x 3 (full indexing), nullable
    float *lvis_out, *ldot_out;           // P x L (full indexing, pre-zeroed), nullable: what the novel-light re-shade consumes
};
void launch_ground_shade(const GroundShade& in, const ra_config& cfg, hipStream_t s);
struct GroundReshade {                    // novel_light_sphere_tracing.render_ground (:70-99)
    const float *ray_d, *albedo_map;      // P x 3
    const float *lvis, *ldot;             // P x L
    const float *ldir, *light_area; int L;
    const float* probes; int n_probes, ph, pw;
    const float* images; int ih, iw;      // nullable
    int attach_envmap, P;
    float *rgb, *albedo, *shade, *spec;   // n_probes x P x 3, nullable
};
void launch_ground_reshade(const GroundReshade& in, hipStream_t s);

// N4: envmap rotation + light-probe inset (ra_trace.hip)
void launch_grow_bounds(float* wbounds6, float margin, hipStream_t s);
void launch_shift_envmap(const float* img, int H, int W, int C, float shift, float* out, hipStream_t s);
struct ProbeInset { float axes[9]; int H, W, uH, uW, ph, pw; };     // axes: columns = right, -front, -down (gen_light_dir)
void launch_light_probe(const ProbeInset& p, const float* probe, float* rgb, hipStream_t s);
void launch_blend_ground(const float* ground, const float* human, const long long* inds, const float* acc, int F, int P, int C, float* dst,
                         hipStream_t s);

// N3: per-frame body state (ra_trace.hip)
void launch_bone_transforms(const float* staged, int J, float* A, float* joints, float* RT, hipStream_t s);
void launch_lbs_verts(const float* tverts, const float* weights, const float* A, const float* big_A, const float* R, const float* Th,
                      int n_verts, int n_bones, float* tpose, float* pverts, float* wverts, hipStream_t s);
void launch_vert_normals(const float* verts, const int* faces, const int* adj_start, const int* adj, int n_verts, float* normals, hipStream_t s);
void launch_bounds(const float* pts, int n, float padding, float* bounds6, hipStream_t s);


// N4: map -> image normalisations (ra_image.hip)
struct ImageJob {
    int type, P;
    const float *a, *b, *acc;             // main map (P x 3 or P), second map (Residual: bpts), acc (P, nullable -> 1)
    const long long* pix;                 // pixel of every ray (nullable -> identity: full-frame maps)
    const float* stats;                   // device: [k-th smallest, k-th largest] where the type needs a percentile
    float cam_R[9], tbounds[6];
    float min_clip, bg;
    int normalize, tonemap;
    float *image, *alpha;                 // H*W x 3, H*W (nullable)
};
size_t image_sort_temp_bytes(long long n);
int launch_percentiles(const float* vals, long long n, const float* acc_flags, int k, float* scratch_a, float* scratch_b, unsigned char* flag,
                       int* count_dev, void* temp, size_t temp_bytes, float* stats, hipStream_t s);
void launch_diff(const float* a, const float* b, long long n, float* o, hipStream_t s);
void launch_compose_image(const ImageJob& j, long long n_pixels, hipStream_t s);
