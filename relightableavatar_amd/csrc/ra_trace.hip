// K5-K8: ray-state kernels around the distance query (gfx950) — all HBM-streaming, SoA, one
// thread per ray / pixel-light pair / pixel; compaction by wave ballot + one atomic per wave.
//
//   reference: sphere_tracing        lib/networks/renderer/sphere_tracing_renderer.py:103-216
//              light_visibility      :265-344     get_near_far_aabb  lib/utils/net_utils.py:1683-1712
//              render_human          :551-784     evaluate_shade     :369-376
//              Microfacet / safe_divide / sample_envmap_image / linear2srgb
//                                    lib/utils/relight_utils.py:484-633,106-127,179-192
//              volume_rendering      lib/utils/net_utils.py:970-999
//              base_renderer         lib/networks/renderer/base_renderer.py:15-113
#include "ra_kernels.hpp"
#include <hipcub/hipcub.hpp>

namespace {

constexpr int TPB = 256;
constexpr float PI_F = 3.14159265358979323846f;

__device__ __forceinline__ int live_count(const int* n_dev, int n) { return n_dev ? min(*n_dev, n) : n; }

// torch.linspace(0, 1, S)[s]: step = 1/(S-1); first half start + s*step, second half end - (S-1-s)*step
__device__ __forceinline__ float linspace01(int s, int S) {
    if (S == 1) return 0.f;
    const float step = 1.f / (float)(S - 1);
    return (s < S / 2) ? (float)s * step : 1.f - (float)(S - 1 - s) * step;
}

// ------------------------------------------------------------------------------------------ trace
__global__ void trace_init_kernel(TraceState ts, int n_launch, const int* n_dev, float offset, float relax) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= live_count(n_dev, n_launch)) return;
    const float nr = ts.near_[i], fr = ts.far_[i];
    ts.t[i] = nr;
    ts.d0[i] = 1e9f;
    ts.occ[i] = 1.f;
    if (ts.stuck) ts.stuck[i] = 0;
    if (ts.dt) ts.dt[i] = 1e9f;
    if (ts.st) ts.st[i] = fr;
    if (ts.ot) ts.ot[i] = fr;
    if (ts.cd) ts.cd[i] = 1e9f;
    if (ts.off) ts.off[i] = offset;
    if (ts.rlx) ts.rlx[i] = relax;
}

template <bool SOFT>
__global__ void trace_update_kernel(TraceState ts, const float* __restrict__ sdf, int n_launch, const int* n_dev, int iter,
                                    ra_trace_params p) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= live_count(n_dev, n_launch)) return;
    const float d1 = sdf[i];
    const float d0 = ts.d0[i];
    float t = ts.t[i];
    const float t_prev = t;
    float occ = ts.occ[i];
    const float nr = ts.near_[i], fr = ts.far_[i];
    float tan_i = p.tan_i;                                   // hard shadow (sphere_tracing_renderer.py:107-110)
    if (SOFT) {
        const float ti = ts.tan_i ? (ts.light ? ts.tan_i[ts.light[i]] : ts.tan_i[i]) : p.tan_i;
        tan_i = p.tan_i_multiplier * ti;
    }
    const float tan2 = (1.f / tan_i) * 2.f;
    float off = ts.off ? ts.off[i] : p.offset;
    float rlx = ts.rlx ? ts.rlx[i] : p.relax;
    float ot = ts.ot ? ts.ot[i] : 0.f;
    if (SOFT && p.clay_book && iter >= p.shadow_skip_iter) {    // :157-172
        const float dx0 = d0 + rlx * d0 + off;
        const float dx1 = d1 + rlx * d1 + off;
        const float dy = (dx1 * dx1) / (2.f * dx0);
        const float dx = (sqrtf(dx1 * dx1 - dy * dy) - off) / (1.f + rlx);
        const float den = fmaxf(fmaxf(t - dy, nr), p.eps);
        const float cls = fmaxf(dx, 0.f) / den / tan2;
        const bool msk = (cls < occ) && (dy < t) && (dx1 > 0.f) && (dx0 > 0.f) && (dx > 0.f) && (dy > 0.f) && (dy < dx0);
        if (msk) { ot = t - dy; occ = cls; }
    }
    if (iter >= p.shadow_skip_iter) {                           // :175-179
        const float cls = fmaxf(d1, 0.f) / fmaxf(fmaxf(t, nr), p.eps) / tan2;
        if (cls < occ) { ot = t; occ = cls; }
    }
    if (!SOFT) {                                                // :182-197
        const float d1u = fabsf(d1), d0u = fabsf(d0);
        float st = ts.st[i], cd = ts.cd[i];
        const float dt = ts.dt[i];
        const float s0 = (d0 > 0.f) ? 1.f : ((d0 < 0.f) ? -1.f : 0.f);
        const float s1 = (d1 > 0.f) ? 1.f : ((d1 < 0.f) ? -1.f : 0.f);
        if (s0 != s1) {
            st = t - dt * fminf(fmaxf(d1u / (d0u + d1u + p.eps), 0.f), 1.f);
            off = 0.f;
            rlx = 0.f;
        }
        if (d1u < cd) { cd = d1u; st = t; }
        ts.st[i] = st;
        ts.cd[i] = cd;
        ts.off[i] = off;
        ts.rlx[i] = rlx;
    }
    const float dtn = d1 + rlx * d1 + off;                      // :200-205
    t = t + dtn;
    t = fminf(t, fr);
    t = fmaxf(t, nr);
    if (ts.dt) ts.dt[i] = dtn;
    ts.t[i] = t;
    // a ray clamped at far (or near) queries the same point again: the distance is already known, the state machine
    // still runs on it (exact), only the query is skipped
    // A shadow ray whose visibility reached 0 is decided (occ = min(occ, cls >= 0) and only occ is read back): no more queries.
    if (ts.stuck) ts.stuck[i] = (t == t_prev || (SOFT && occ == 0.f)) ? 1 : 0;
    ts.d0[i] = d1;
    ts.occ[i] = occ;
    if (ts.ot) ts.ot[i] = ot;
}

// ------------------------------------------------------------------------------------------ surface
__global__ void surface_finish_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ st,
                                      const float* __restrict__ occ, int P, float* __restrict__ surf, float* __restrict__ depth,
                                      float* __restrict__ acc, int* __restrict__ hit_idx, int* __restrict__ hit_count, int* __restrict__ slot_of_ray) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    bool hit = false;
    if (i < P) {
        if (slot_of_ray) slot_of_ray[i] = -1;
        const float s = st[i];
        const float sx = ro[3 * i] + s * rd[3 * i], sy = ro[3 * i + 1] + s * rd[3 * i + 1], sz = ro[3 * i + 2] + s * rd[3 * i + 2];
        surf[3 * i] = sx; surf[3 * i + 1] = sy; surf[3 * i + 2] = sz;
        depth[i] = (sx - ro[3 * i]) / rd[3 * i];                // :574 (x component only)
        const float a = 1.f - occ[i];                           // :575
        acc[i] = a;
        hit = a > 0.f;                                          // :590
    }
    const unsigned long long m = __ballot(hit);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(hit_count, __popcll(m));
    base = __shfl(base, 0);
    if (hit) hit_idx[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

__device__ __forceinline__ unsigned expand10t(unsigned v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// sort key of a ray: Morton code of its surface point on a 1/256 m grid inside the box (hit) or
// 0xffffffff (miss) -> after the sort the first hit_count values are the hit rays in an order where
// 64 consecutive pixels see neighbouring surface points (coherent waves for the shadow trace).
__global__ void hit_keys_kernel(const float* __restrict__ surf, const float* __restrict__ acc, int P, float bx, float by, float bz,
                                unsigned* __restrict__ keys, int* __restrict__ vals) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= P) return;
    unsigned k = 0xffffffffu;
    if (acc[i] > 0.f) {
        const unsigned qx = (unsigned)fminf(fmaxf((surf[3 * i] - bx) * 256.f, 0.f), 1023.f);
        const unsigned qy = (unsigned)fminf(fmaxf((surf[3 * i + 1] - by) * 256.f, 0.f), 1023.f);
        const unsigned qz = (unsigned)fminf(fmaxf((surf[3 * i + 2] - bz) * 256.f, 0.f), 1023.f);
        k = (expand10t(qx) << 2) | (expand10t(qy) << 1) | expand10t(qz);
    }
    keys[i] = k;
    vals[i] = i;
}

// sort key of a primary ray: Morton code of its entry point (o + near d) on a 1/256 m grid inside the box, so that
// the 64 rays of a wave start (and stay) close together -> compact box sweeps in the coarse level
__global__ void ray_keys_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ nr, int P,
                                float bx, float by, float bz, const float* __restrict__ origin_dev, unsigned* __restrict__ keys,
                                int* __restrict__ vals) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= P) return;
    if (origin_dev) { bx = origin_dev[0]; by = origin_dev[1]; bz = origin_dev[2]; }
    const float t = nr[i];
    const unsigned qx = (unsigned)fminf(fmaxf((ro[3 * i] + t * rd[3 * i] - bx) * 256.f, 0.f), 1023.f);
    const unsigned qy = (unsigned)fminf(fmaxf((ro[3 * i + 1] + t * rd[3 * i + 1] - by) * 256.f, 0.f), 1023.f);
    const unsigned qz = (unsigned)fminf(fmaxf((ro[3 * i + 2] + t * rd[3 * i + 2] - bz) * 256.f, 0.f), 1023.f);
    keys[i] = (expand10t(qx) << 2) | (expand10t(qy) << 1) | expand10t(qz);
    vals[i] = i;
}

// lower corner of the rays' entry points (one workgroup): origin of the Morton grid when the caller gives no box
__global__ __launch_bounds__(1024) void entry_min_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ nr,
                                                         int P, float* __restrict__ out) {
    __shared__ float sm_[16][3];
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f};
    for (int i = threadIdx.x; i < P; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) lo[c] = fminf(lo[c], ro[3 * i + c] + nr[i] * rd[3 * i + c]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) sm_[threadIdx.x >> 6][c] = lo[c];
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = sm_[0][threadIdx.x];
        for (int w = 1; w < 16; ++w) a = fminf(a, sm_[w][threadIdx.x]);
        out[threadIdx.x] = a;
    }
}

__global__ void gather_rays_kernel(const int* __restrict__ perm, int P, const float* __restrict__ ro, const float* __restrict__ rd,
                                   const float* __restrict__ nr, const float* __restrict__ fr, float* __restrict__ so, float* __restrict__ sd,
                                   float* __restrict__ sn, float* __restrict__ sf, float near_min, float far_max) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= P) return;
    const int r = perm[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { so[3 * i + c] = ro[3 * r + c]; sd[3 * i + c] = rd[3 * r + c]; }
    sn[i] = fmaxf(nr[r], near_min);     // the volume renderer's near.clip(min=clip_near) / far.clip(max=clip_far) (base_renderer.py:120-121);
    sf[i] = fminf(fr[r], far_max);      // -inf / +inf elsewhere
}

// a shard's rays out of the frame's (relightableavatar_amd/shard.py shard_batch): one launch instead of four index kernels of the host framework
__global__ void gather_shard_rays_kernel(const long long* __restrict__ idx, int n, const float* __restrict__ ro, const float* __restrict__ rd,
                                         const float* __restrict__ nr, const float* __restrict__ fr, float* __restrict__ so, float* __restrict__ sd,
                                         float* __restrict__ sn, float* __restrict__ sf) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const long long r = idx[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { so[3 * i + c] = ro[3 * r + c]; sd[3 * i + c] = rd[3 * r + c]; }
    sn[i] = nr[r];
    sf[i] = fr[r];
}

// the un-interleave after the frame all_gather: dst[dst_idx[i]] = src[src_idx[i]], rows of C floats (shard.py _exchange)
__global__ void scatter_rows_kernel(const float* __restrict__ src, const long long* __restrict__ src_idx, const long long* __restrict__ dst_idx,
                                    long long n, int C, float* __restrict__ dst) {
    const long long k = (long long)blockIdx.x * TPB + threadIdx.x;
    if (k >= n * C) return;
    const long long i = k / C;
    const int c = (int)(k - i * C);
    dst[dst_idx[i] * C + c] = src[src_idx[i] * C + c];
}

__global__ void surface_samples_kernel(const float* __restrict__ surf, const float* __restrict__ rd, const int* __restrict__ hit_idx,
                                       const int* __restrict__ hit_count, int S, float range, float* __restrict__ x,
                                       float* __restrict__ v, int* __restrict__ n_out) {
    const int nh = *hit_count;
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k == 0) *n_out = nh * S;
    if (k >= nh * S) return;
    const int h = k / S, s = k - h * S;
    const int r = hit_idx[h];
    const float zv = (S == 1) ? 0.5f : linspace01(s, S);            // :607-611
    const float z = zv * (2.f * range) - range;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float d = rd[3 * r + c];
        x[3 * k + c] = surf[3 * r + c] + z * d;
        v[3 * k + c] = d;
    }
}

// volume-render the S surface samples of one hit pixel, renormalise by acc (:616-620), split and
// clip (:629-655)
__global__ void surface_composite_kernel(const float* __restrict__ raw, int C, int S, const int* __restrict__ hit_count, int relight,
                                         ra_config cfg, SurfaceMaps m) {
    const int h = blockIdx.x * TPB + threadIdx.x;
    if (h >= *hit_count) return;
    float out[16];
    const int CC = C - 1;
#pragma unroll
    for (int c = 0; c < 16; ++c) out[c] = 0.f;
    float T = 1.f, accw = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* r = raw + ((size_t)h * S + s) * C;
        const float a = r[CC];
        const float w = a * T;                                  // alpha * cumprod(1 - alpha + 1e-8) (net_utils.py:987-990)
        T *= (1.f - a + 1e-8f);
        accw += w;
        for (int c = 0; c < CC; ++c) out[c] += w * r[c];
    }
    // (+ (1 - acc) * bg_brightness, then / (occ + 1e-8))
    const float inv = 1.f / (accw + 1e-8f);
    for (int c = 0; c < CC; ++c) out[c] = (out[c] + (1.f - accw) * cfg.bg_brightness) * inv;
    float* nrm = relight ? out + 13 : out + 9;
    if (nrm[0] + nrm[1] + nrm[2] == 0.f) { nrm[0] = nrm[1] = nrm[2] = 1.f; }     // :642
    const float nn = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]) + 1e-8f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m.cpts[3 * h + c] = out[c];
        m.bpts[3 * h + c] = out[3 + c];
        m.resd[3 * h + c] = out[6 + c];
        m.norm[3 * h + c] = nrm[c] / nn;
    }
    if (relight) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = fminf(fmaxf(out[9 + c], cfg.albedo_bias), cfg.albedo_bias + cfg.albedo_slope);
            if (m.valbedo) m.valbedo[3 * h + c] = a;
            if (cfg.albedo_multiplier > 0.f) a *= cfg.albedo_multiplier;
            m.albedo[3 * h + c] = a;
        }
        m.rough[h] = fminf(fmaxf(out[12], cfg.roughness_bias), cfg.roughness_bias + cfg.roughness_slope);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) m.rgb[3 * h + c] = out[12 + c];
    }
}

// ------------------------------------------------------------------------------------------ shadow rays
__global__ void light_dirs_kernel(const float* __restrict__ xyz, int L, float* __restrict__ ldir) {
    const int l = blockIdx.x * TPB + threadIdx.x;
    if (l >= L) return;
    const float x = xyz[3 * l], y = xyz[3 * l + 1], z = xyz[3 * l + 2];
    const float n = sqrtf(x * x + y * y + z * z) + 1e-8f;       // normalize(), net_utils.py:1626
    ldir[3 * l] = x / n; ldir[3 * l + 1] = y / n; ldir[3 * l + 2] = z / n;
}

// workgroup = 64 neighbouring hit slots x 32 lights; a wave handles ONE light for the 64 slots at a time (so that the
// traced rays it emits stay neighbours in the coarse level), 8 rounds cover the 32 lights.  lvis / ldot are [slot][light]:
// they are staged in LDS and written as whole 128-byte rows (a direct store would touch 64 cache lines per instruction).
// get_near_far_aabb (net_utils.py:1683-1712, return_raw path): direction components in (-1e-16, 1e-8) become +1e-8 in place
// (:1698; the mirrored line :1699 then matches nothing, so tiny negatives stay), slab test
__device__ __forceinline__ void aabb_near_far(const float* bbox, const float o[3], const float dir[3], float& nr, float& fr) {
    float d[3] = {dir[0], dir[1], dir[2]};
    nr = -3.0e38f; fr = 3.0e38f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (d[c] < 1e-8f && d[c] > -1e-16f) d[c] = 1e-8f;
        const float t0 = (bbox[c] - o[c]) / d[c], t1 = (bbox[3 + c] - o[c]) / d[c];
        nr = fmaxf(nr, fminf(t0, t1));
        fr = fminf(fr, fmaxf(t0, t1));
    }
}
struct Box6 { float v[6]; };
__global__ void debug_aabb_kernel(const float* __restrict__ o, const float* __restrict__ d, int n, Box6 b, float* __restrict__ nr, float* __restrict__ fr) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const float oo[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]}, dd[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    float a, f;
    aabb_near_far(b.v, oo, dd, a, f);
    nr[i] = a; fr[i] = f;
}

constexpr int SG_LIGHTS = 32;
constexpr float SG_SPLIT_EXTENT = 0.15f;      // metres: a group of 64 Morton-neighbour hit pixels wider than this lies in two places
__global__ __launch_bounds__(TPB) void shadow_gen_kernel(ShadowGen g) {
    __shared__ float t_ldot[64][SG_LIGHTS + 1], t_lvis[64][SG_LIGHTS + 1];
    const int nh = *g.hit_count;
    const int lchunks = (g.L + SG_LIGHTS - 1) / SG_LIGHTS;
    const int grp = blockIdx.x / lchunks, lc = blockIdx.x - grp * lchunks;
    if (grp * 64 >= nh) return;                 // uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int h = grp * 64 + lane;
    const bool hv = h < nh;
    int r = 0;
    float nrm[3] = {0.f, 0.f, 0.f}, o[3] = {0.f, 0.f, 0.f}, acc = 0.f;
    float box[6] = {g.bbox[0], g.bbox[1], g.bbox[2], g.bbox[3], g.bbox[4], g.bbox[5]};
    if (hv) {
        r = g.hit_idx[h];
        if (g.n_boxes > 1) {                        // the box of the render chunk this ray belongs to (kernel arguments: scalar loads)
            const int rc = g.perm ? g.perm[r] : r;          // position in the caller's ray order
            int j = 0;
            for (int k = 1; k < g.n_boxes; ++k) j += rc >= g.box_start[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) box[k] = g.boxes[j][k];
        }
        nrm[0] = g.norm[3 * h]; nrm[1] = g.norm[3 * h + 1]; nrm[2] = g.norm[3 * h + 2];
        o[0] = g.surf[3 * r]; o[1] = g.surf[3 * r + 1]; o[2] = g.surf[3 * r + 2];
        acc = g.acc[r];
    }
    constexpr int ROUNDS = SG_LIGHTS / (TPB / 64);
    constexpr int NL = ROUNDS * (TPB / 64);               // lights per workgroup (= SG_LIGHTS)
    __shared__ int cnt[2][NL + 1];                        // traced rays per light among slots 0..31 / 32..63 of the group, then their offsets
    float nrv[ROUNDS], frv[ROUNDS];
    unsigned tmask = 0;
    // A group is 64 hit pixels that are neighbours in the Morton order of the surface points: a patch of a few centimetres in a whole frame.
    // In ONE RANK's share of a sharded frame (8 x 8 pixel tiles dealt round robin) a group that is not aligned with a tile holds pixels of two
    // tiles 25 cm apart, and every wave of the coarse level then sweeps the boxes of both places (profiles/r05_shard_tile_size.txt).  Such a
    // group emits its rays half by half — the first 32 slots for all lights, then the other 32 — so that a wave of 64 consecutive rays stays in
    // one place (4 lights x ~16 surviving pixels instead of 2 x ~32).  Only the order of the ray list changes; results are scattered by slot.
    float ext = 0.f;
    {
        float lo3[3], hi3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo3[k] = hv ? o[k] : 3e38f; hi3[k] = hv ? o[k] : -3e38f; }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) { lo3[k] = fminf(lo3[k], __shfl_xor(lo3[k], sh)); hi3[k] = fmaxf(hi3[k], __shfl_xor(hi3[k], sh)); }
        ext = fmaxf(fmaxf(hi3[0] - lo3[0], hi3[1] - lo3[1]), hi3[2] - lo3[2]);
    }
    // (not for the ground pass: its groups are metres wide anyway and four lights per wave cost more than they save there: + 9 % on its coarse
    // launches, measured)
    const bool halves = g.split_wide_groups && ext > SG_SPLIT_EXTENT;            // uniform over the workgroup: its four waves hold the same 64 slots
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
        const int ll = round * (TPB / 64) + wv;            // light within the chunk (wave-uniform)
        const int l = lc * SG_LIGHTS + ll;
        bool trace = false;
        float nr = 0.f, fr = 0.f;
        if (hv && l < g.L) {
            const float dx = g.ldir[3 * l], dy = g.ldir[3 * l + 1], dz = g.ldir[3 * l + 2];
            const float ldot = dx * nrm[0] + dy * nrm[1] + dz * nrm[2];                                   // :292
            float lv;
            if (g.no_visibility) lv = 1.f;
            else if (g.local_visibility) lv = ldot > 0.f ? 1.f : 0.f;
            else {
                const bool front = (ldot > 0.f) && (acc > 0.f);                                           // :303
                lv = 0.f;
                if (front) {
                    const float d[3] = {dx, dy, dz};
                    aabb_near_far(box, o, d, nr, fr);
                    nr = fmaxf(nr, g.near_offset);                                                        // :311
                    fr = fmaxf(fr, g.near_offset);
                    trace = nr < fr;
                    lv = 1.f;           // outside the box: visible (:341); traced rays are overwritten later
                }
            }
            t_ldot[lane][ll] = ldot;
            t_lvis[lane][ll] = lv;
        }
        nrv[round] = nr; frv[round] = fr;
        if (trace) tmask |= 1u << round;
        const unsigned long long m = __ballot(trace);
        if (lane == 0) { cnt[0][ll] = __popcll(m & 0xffffffffull); cnt[1][ll] = __popcll(m >> 32); }
    }
    __syncthreads();
    // ONE atomic per workgroup for the traced-ray list (same-address atomics serialise: per wave they were the whole kernel
    // time); a wave's rays (one light x 64 neighbouring slots — or, for a group in two places, 32) stay contiguous
    if (threadIdx.x == 0) {
        int tot = 0;
        if (halves) {
            for (int hf = 0; hf < 2; ++hf)
                for (int k = 0; k < NL; ++k) { const int c = cnt[hf][k]; cnt[hf][k] = tot; tot += c; }
        } else {
            for (int k = 0; k < NL; ++k) {
                const int c0 = cnt[0][k], c1 = cnt[1][k];
                cnt[0][k] = tot; cnt[1][k] = tot + c0; tot += c0 + c1;
            }
        }
        cnt[0][NL] = tot ? atomicAdd(g.ray_count, tot) : 0;
    }
    __syncthreads();
    const int gbase = cnt[0][NL];
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
        const int ll = round * (TPB / 64) + wv;
        const bool trace = (tmask >> round) & 1u;
        const unsigned long long m = __ballot(trace);
        if (trace) {
            const int hf = lane >> 5;
            const unsigned mh = hf ? (unsigned)(m >> 32) : (unsigned)m;
            const int s = gbase + cnt[hf][ll] + __popc(mh & ((1u << (lane & 31)) - 1u));
            const int l = lc * SG_LIGHTS + ll;
            g.ray_pix[s] = r;
            g.ray_light[s] = l;
            g.ray_slot[s] = h * g.L + l;
            g.near_[s] = nrv[round];
            g.far_[s] = frv[round];
        }
    }
    // rows of 32 lights: thread t -> (row = t / 32 + 8 k, light = t % 32)
    const int col = threadIdx.x & (SG_LIGHTS - 1), l = lc * SG_LIGHTS + col;
    for (int row = threadIdx.x / SG_LIGHTS; row < 64; row += TPB / SG_LIGHTS) {
        const int hh = grp * 64 + row;
        if (hh < nh && l < g.L) {
            g.ldot[(size_t)hh * g.L + l] = t_ldot[row][col];
            g.lvis[(size_t)hh * g.L + l] = t_lvis[row][col];
        }
    }
}

__global__ void shadow_scatter_kernel(const float* __restrict__ occ, const int* __restrict__ ray_slot, const int* __restrict__ ray_count,
                                      int max_rays, float* __restrict__ lvis) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= min(*ray_count, max_rays)) return;
    lvis[ray_slot[i]] = occ[i];
}

// ------------------------------------------------------------------------------------------ shading
// safe_divide with its in-place clamps (relight_utils.py:618-633). a and b are clamped by reference
// because the reference aliases them with tensors it keeps using.
__device__ __forceinline__ float safe_div(float& a, float& b) {
    const float eps = 1e-8f;
    if (a < eps && a >= 0.f) a = eps;
    if (a > -eps && a <= 0.f) a = -eps;
    if (b < eps && b >= 0.f) b = eps;
    if (b > -eps && b <= 0.f) b = -eps;
    float d = a / b;
    if (d != d) d = 0.f;
    if (isinf(d)) d = 0.f;
    return fminf(fmaxf(d, -1e10f), 1e10f);
}

__device__ __forceinline__ void fnormalize(float v[3]) {       // F.normalize(eps=1e-7)
    const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-7f);
    v[0] /= n; v[1] /= n; v[2] /= n;
}

// equirect bilinear lookup, align_corners=False, border padding (relight_utils.py:106-127)
__device__ __forceinline__ void sample_probe(const float* __restrict__ img, int H, int W, const float d[3], float out[3]) {
    const float theta = acosf(d[2]) - 1e-6f;
    const float phi = atan2f(d[1], d[0]);
    const float qy = (theta / PI_F) * 2.f - 1.f;
    const float qx = -phi / PI_F;
    float ix = ((qx + 1.f) * W - 1.f) * 0.5f;
    float iy = ((qy + 1.f) * H - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    out[0] = out[1] = out[2] = 0.f;
    auto add = [&](int xx, int yy, float w) {
        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
            const float* p = img + ((size_t)yy * W + xx) * 3;
            out[0] += w * p[0]; out[1] += w * p[1]; out[2] += w * p[2];
        }
    };
    add(x0, y0, wx0 * wy0);
    add(x1, y0, wx1 * wy0);
    add(x0, y1, wx0 * wy1);
    add(x1, y1, wx1 * wy1);
}

// The key lights of a frame: the lights that hold at least the fraction `share` of a probe's power (radiance x solid angle, channel mean)
// — and at least four times the mean share 1 / L: a small light set has no key lights just because it is small — under ANY of the frame's
// probes; at most kmax of them in all, those with the largest share (under the probe that favours them most) first.  The light-visibility
// rays towards them are traced in compensated arithmetic (ra_config.key_light_share): a DFSS penumbra value is d * sharp / (2 t), which
// amplifies the 5e-5 distance error of plain f16 operands up to 500 x per light; summed over a probe's 512 lights those errors average
// out — unless a few lights carry the probe's power, whose rays leave a pixel in nearly the same direction and err together.
// smax[l]: the largest share of light l so far (accumulate != 0: over the probes of earlier calls too).  One workgroup; L floats of LDS.
__global__ __launch_bounds__(TPB) void key_lights_kernel(const float* __restrict__ probes, int n, int ph, int pw, const float* __restrict__ ldir,
                                                         const float* __restrict__ area, int L, float share, int kmax, int accumulate,
                                                         float* __restrict__ smax, unsigned char* __restrict__ key) {
    extern __shared__ float kl_w[];
    __shared__ float red[TPB / 64];
    for (int q = 0; q < n; ++q) {
        const float* img = probes + (size_t)q * ph * pw * 3;
        float part = 0.f;
        for (int l = threadIdx.x; l < L; l += TPB) {
            const float d[3] = {ldir[3 * l], ldir[3 * l + 1], ldir[3 * l + 2]};
            float c[3];
            sample_probe(img, ph, pw, d, c);
            float w = (c[0] + c[1] + c[2]) * (1.f / 3.f) * area[l];
            w = w > 0.f ? w : 0.f;                    // (also a NaN radiance)
            kl_w[l] = w;
            part += w;
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) part += __shfl_xor(part, sh);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
        float total = 0.f;
#pragma unroll
        for (int k = 0; k < TPB / 64; ++k) total += red[k];
        for (int l = threadIdx.x; l < L; l += TPB) {
            const float sh_l = total > 0.f ? kl_w[l] / total : 0.f;
            smax[l] = (q == 0 && !accumulate) ? sh_l : fmaxf(smax[l], sh_l);      // each light is one thread's
        }
        __syncthreads();
    }
    for (int l = threadIdx.x; l < L; l += TPB) kl_w[l] = smax[l];
    __syncthreads();
    const float thr = fmaxf(share, 4.f / (float)L);
    for (int l = threadIdx.x; l < L; l += TPB) {
        const float w = kl_w[l];
        bool cand = w >= thr;
        if (cand) {
            int rank = 0;
            for (int j = 0; j < L; ++j) rank += (kl_w[j] > w) || (kl_w[j] == w && j < l);
            cand = rank < kmax;
        }
        key[l] = (unsigned char)cand;
    }
}

__device__ __forceinline__ float srgb(float x) {                // relight_utils.py:179-192
    x = fminf(fmaxf(x, 0.f), 1.f);
    return (x <= 0.0031308f) ? x * 12.92f : 1.055f * powf(x + 1e-7f, 1.f / 2.4f) - (1.055f - 1.f);
}

// Microfacet.__call__ (relight_utils.py:484-577, cancel_cosine = True) split into its per-pixel and per-light parts;
// safe_divide's in-place clamps of its arguments (the aliasing of cos^2 in _get_d / _get_g) are reproduced.
struct MfView { float v[3], n[3], a2, v_dot_n, cos_v, g_den0; };
__device__ __forceinline__ MfView mf_view(const float p2c[3], const float normal[3], float rough) {
    MfView m;
#pragma unroll
    for (int c = 0; c < 3; ++c) { m.v[c] = p2c[c]; m.n[c] = normal[c]; }
    fnormalize(m.v);
    fnormalize(m.n);
    const float alpha = rough * rough;
    m.a2 = alpha * alpha;
    m.v_dot_n = fminf(fmaxf(m.v[0] * m.n[0] + m.v[1] * m.n[1] + m.v[2] * m.n[2], 1e-4f), 1.f);
    // view-only part of G (_get_g :580-595); cos_theta_v is clamped in place by the first safe_divide
    float cos_v = m.n[0] * m.v[0] + m.n[1] * m.v[1] + m.n[2] * m.v[2];
    {
        const float eps = 1e-8f;
        if (cos_v < eps && cos_v >= 0.f) cos_v = eps;
        if (cos_v > -eps && cos_v <= 0.f) cos_v = -eps;
    }
    m.cos_v = cos_v;
    float cvs = fminf(fmaxf(cos_v * cos_v, 0.f), 1.f);
    float one_m = 1.f - cvs;
    float tan_v_sq = safe_div(one_m, cvs);
    tan_v_sq = fminf(fmaxf(tan_v_sq, 0.f), 1e10f);
    m.g_den0 = 1.f + sqrtf(1.f + m.a2 * tan_v_sq);
    return m;
}
// brdf[c] = glossy + albedo/pi * clip(l.n) (or the ablation variants); sbrdf = the albedo-0 value (:740)
__device__ __forceinline__ void mf_light(const MfView& m, const float p2l[3], const float alb[3], const ra_config& cfg, float brdf[3], float& sbrdf) {
    float pl[3] = {p2l[0], p2l[1], p2l[2]};
    fnormalize(pl);
    const float l_dot_n = fminf(fmaxf(pl[0] * m.n[0] + pl[1] * m.n[1] + pl[2] * m.n[2], 1e-4f), 1.f);
    float hv[3] = {pl[0] + m.v[0], pl[1] + m.v[1], pl[2] + m.v[2]};
    fnormalize(hv);
    const float omc5 = 1.f - (pl[0] * hv[0] + pl[1] * hv[1] + pl[2] * hv[2]);
    const float f = cfg.fresnel_f0 + (1.f - cfg.fresnel_f0) * (omc5 * omc5 * omc5 * omc5 * omc5);
    // D (_get_d :598-608)
    const float cos_m = hv[0] * m.n[0] + hv[1] * m.n[1] + hv[2] * m.n[2];
    const float chi_d = cos_m > 0.f ? 1.f : 0.f;
    float cms = cos_m * cos_m;
    float omc = 1.f - cms;
    const float tan_m_sq = safe_div(omc, cms);          // clamps cms in place
    float dden = PI_F * (cms * cms) * ((m.a2 + tan_m_sq) * (m.a2 + tan_m_sq));
    float dnum = m.a2 * chi_d;
    const float dd = safe_div(dnum, dden);
    // G
    float cos_t = hv[0] * m.v[0] + hv[1] * m.v[1] + hv[2] * m.v[2];
    float cvc = m.cos_v;
    const float dv = safe_div(cos_t, cvc);
    float gnum = (dv > 0.f ? 1.f : 0.f) * 2.f;
    float gden = m.g_den0;
    const float gg = safe_div(gnum, gden);
    float mnum = f * gg * dd;
    float mden = 4.f * 1.f * fabsf(m.v_dot_n);
    const float glossy = safe_div(mnum, mden);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lam = alb[c] / PI_F * l_dot_n;
        brdf[c] = cfg.lambert_only ? lam : (cfg.glossy_only ? glossy : glossy + lam);
    }
    sbrdf = cfg.lambert_only ? 0.f : glossy;
}
// test hook: the BRDF on arbitrary (light, point) direction pairs, p2l (L,N,3) like the reference's surf2light
__global__ void debug_brdf_kernel(const float* __restrict__ p2l, const float* __restrict__ p2c, const float* __restrict__ nrm, const float* __restrict__ alb,
                                  const float* __restrict__ rough, int L, int N, ra_config cfg, float* __restrict__ out) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= L * N) return;
    const int p = i % N;
    const float v[3] = {p2c[3 * p], p2c[3 * p + 1], p2c[3 * p + 2]}, n[3] = {nrm[3 * p], nrm[3 * p + 1], nrm[3 * p + 2]};
    const float a[3] = {alb[3 * p], alb[3 * p + 1], alb[3 * p + 2]}, l[3] = {p2l[3 * i], p2l[3 * i + 1], p2l[3 * i + 2]};
    const MfView mv = mf_view(v, n, rough[p]);
    float brdf[3], sb;
    mf_light(mv, l, a, cfg, brdf, sb);
    out[3 * i] = brdf[0]; out[3 * i + 1] = brdf[1]; out[3 * i + 2] = brdf[2];
}

// one wave per (pixel slot); lanes stride over the L lights; probes looped inside so the BRDF,
// visibility and area weights are computed once per light for all probes.
constexpr int MAXP = 8;
__global__ __launch_bounds__(TPB) void shade_kernel(ShadeIn in, ra_config cfg) {
    const int n = in.count ? min(*in.count, in.n) : in.n;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const int h = wave;
    const int r = in.idx ? in.idx[h] : h;
    const float sp[3] = {in.surf[3 * r], in.surf[3 * r + 1], in.surf[3 * r + 2]};
    float v[3] = {in.ray_o[3 * r] - sp[0], in.ray_o[3 * r + 1] - sp[1], in.ray_o[3 * r + 2] - sp[2]};
    {   // surf2cam = normalize(ray_o - surf)  (:716), then F.normalize inside Microfacet
        const float nn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-8f;
        v[0] /= nn; v[1] /= nn; v[2] /= nn;
        fnormalize(v);
    }
    const float nraw[3] = {in.norm[3 * h], in.norm[3 * h + 1], in.norm[3 * h + 2]};
    const float alb[3] = {in.albedo[3 * h], in.albedo[3 * h + 1], in.albedo[3 * h + 2]};
    const MfView mv = mf_view(v, nraw, in.rough[h]);

    float rgb[MAXP][3], shd[MAXP][3], spc[MAXP][3];
#pragma unroll
    for (int q = 0; q < MAXP; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[q][c] = shd[q][c] = spc[q][c] = 0.f;
    float vis_sum = 0.f;                                       // cfg.vis_shade_map: sum over the lights of lvis (1) / ldot (2)

    for (int l = lane; l < in.L; l += 64) {
        float s2l[3] = {in.light_xyz[3 * l] - sp[0], in.light_xyz[3 * l + 1] - sp[1], in.light_xyz[3 * l + 2] - sp[2]};
        {
            const float nn = sqrtf(s2l[0] * s2l[0] + s2l[1] * s2l[1] + s2l[2] * s2l[2]) + 1e-8f;      // :715
            s2l[0] /= nn; s2l[1] /= nn; s2l[2] /= nn;
        }
        float brdf[3], sbrdf;
        mf_light(mv, s2l, alb, cfg, brdf, sbrdf);
        const float area = in.light_area[l];
        const float lv = in.lvis[(size_t)h * in.L + l];
        const float ld = cfg.only_visibility ? 1.f : in.ldot[(size_t)h * in.L + l];       // :720-722
        const float spec_ld = 1.f / (fabsf(1.f) + 1e-8f);    // :743
        vis_sum += cfg.vis_shade_map == 1 ? lv : ld;
        for (int q = 0; q < in.n_probes; ++q) {
            float Lr[3];
            sample_probe(in.probes + (size_t)q * in.ph * in.pw * 3, in.ph, in.pw, s2l, Lr);
            if (cfg.only_visibility) Lr[0] = Lr[1] = Lr[2] = (Lr[0] + Lr[1] + Lr[2]) / 3.f;        // light.mean(dim=-1) (:723)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sh = lv * 1.f * area * Lr[c];     // cancel_cosine: ldot -> 1 (:724-727)
                rgb[q][c] += brdf[c] * sh;
                shd[q][c] += lv * ld * area * Lr[c];
                spc[q][c] += sbrdf * (1.f * spec_ld * area * Lr[c]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vis_sum += __shfl_xor(vis_sum, o);
    for (int q = 0; q < in.n_probes; ++q) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = rgb[q][c], b = shd[q][c], d = spc[q][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); d += __shfl_xor(d, o); }
            if (lane == 0) {
                const size_t k = ((size_t)q * in.n + h) * 3 + c;
                if (in.rgb) in.rgb[k] = cfg.tonemapping ? srgb(a) : a;
                if (in.shade) in.shade[k] = cfg.vis_shade_map ? vis_sum / (float)in.L : b * cfg.shading_albedo / PI_F;     // :756-757
                if (in.spec && in.want_spec) in.spec[k] = d;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ scatter
__global__ void slot_index_kernel(const int* __restrict__ hit_idx, const int* __restrict__ hit_count, int* __restrict__ slot_of_ray) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k < *hit_count) slot_of_ray[hit_idx[k]] = k;
}

// all output maps of a chunk in one launch (replaces one memset + one scatter per map): element k of the concatenated
// (ray, channel) spaces of the jobs; a ray without a hit slot writes zeros (multi_scatter_zeros, :680-688), a hit ray its slot's
// values, times acc where the map is premultiplied (alpha_output_)
__global__ void emit_maps_kernel(EmitMaps m) {
    long long k = (long long)blockIdx.x * TPB + threadIdx.x;
    int j = 0;
    long long start = 0;
    while (j < m.n_jobs && k >= m.end[j]) { start = m.end[j]; ++j; }
    if (j >= m.n_jobs) return;
    const MapJob& J = m.job[j];
    k -= start;
    const int r = (int)(k / J.C), c = (int)(k - (long long)r * J.C);
    const int slot = m.slot_of_ray[r];
    float v = 0.f;
    if (slot >= 0) {
        v = J.src_full ? J.src[(size_t)r * J.C + c] : J.src[(size_t)slot * J.C + c];
        if (J.premul) v *= m.acc[r];
    }
    J.dst[(size_t)(m.perm ? m.perm[r] : r) * J.C + c] = v;      // perm: internal (sorted) ray -> caller's ray index
}

__global__ void scatter_maps_kernel(const int* __restrict__ hit_idx, const int* __restrict__ hit_count, int premultiply,
                                    const float* __restrict__ acc_full, const float* __restrict__ src, int C, float* __restrict__ dst,
                                    int src_full, const int* __restrict__ perm) {
    const long long k = (long long)blockIdx.x * TPB + threadIdx.x;
    const int nh = *hit_count;
    if (k >= (long long)nh * C) return;
    const int h = (int)(k / C), c = (int)(k - (long long)h * C);
    const int r = hit_idx[h];
    float v = src_full ? src[(size_t)r * C + c] : src[k];
    if (premultiply) v *= acc_full[r];
    dst[(size_t)(perm ? perm[r] : r) * C + c] = v;      // perm: internal (sorted) ray -> caller's ray index
}

__global__ void gather_rows_kernel(const int* __restrict__ hit_idx, const int* __restrict__ hit_count, const float* __restrict__ src,
                                   int C, float* __restrict__ dst) {
    const long long k = (long long)blockIdx.x * TPB + threadIdx.x;
    const int nh = *hit_count;
    if (k >= (long long)nh * C) return;
    const int h = (int)(k / C), c = (int)(k - (long long)h * C);
    dst[k] = src[(size_t)hit_idx[h] * C + c];
}

__global__ void accumulate_kernel(const int* __restrict__ count, unsigned long long* __restrict__ dst) {
    atomicAdd(dst, (unsigned long long)*count);
}

__global__ void fill_kernel(float* p, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------ volume path
// sample (ray r, depth s) lives at ((r / 64) * S + s) * 64 + r % 64: a wave = 64 neighbouring rays (the chunk's rays are
// Morton-sorted) at ONE depth, so the coarse level sweeps a centimetre-sized patch instead of half a ray
__device__ __forceinline__ size_t vol_slot(int r, int s, int S) { return ((size_t)(r >> 6) * S + s) * 64 + (r & 63); }

__global__ void volume_samples_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ nr,
                                      const float* __restrict__ fr, int P, int S, float* __restrict__ x, float* __restrict__ v) {
    const long long k = (long long)blockIdx.x * TPB + threadIdx.x;
    const int Pp = (P + 63) & ~63;
    if (k >= (long long)Pp * S) return;
    const int j = (int)(k & 63);
    const long long gs = k >> 6;
    const int s = (int)(gs % S), rg = (int)(gs / S);
    const bool pad = rg * 64 + j >= P;                          // padding slots of the last 64-ray group: never composited
    const int r = min(rg * 64 + j, P - 1);
    const float tv = linspace01(s, S);
    const float z = nr[r] * (1.f - tv) + fr[r] * tv;            // base_renderer.py:17-18
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // a padding sample sits far outside every box: the coarse level drops it, so it costs no full query and is not
        // counted as one (it used to repeat the last ray: up to 63 x S duplicate queries per launch in n_fine_full)
        x[3 * k + c] = pad ? 1.0e6f : ro[3 * r + c] + rd[3 * r + c] * z;
        v[3 * k + c] = rd[3 * r + c];
    }
}

// One workgroup composites 64 neighbouring rays (one row group of the sample layout): 16 depth segments x 64 rays.  A thread
// composites its segment with a local transmittance starting at 1, the segments' transmittances are chained through LDS, and
// thread (g, j) then sums channel g of ray j over the segments in depth order (a fixed order: the image does not depend on how
// the rays are chunked).  One thread per ray walking all S samples (the first version) left 128 waves per 8192-ray chunk on the
// chip, each with S dependent strided reads: 326 us for 67 MB.
constexpr int VC_SEG = 16;
__global__ __launch_bounds__(64 * VC_SEG) void volume_composite_kernel(const float* __restrict__ raw, const float* __restrict__ nr,
                                                                          const float* __restrict__ fr, int P, int S, float bg, ra_render_out out,
                                                                          const int* __restrict__ perm) {
    __shared__ float part[VC_SEG][17][64];      // [segment][15 channels, acc, depth][ray]
    __shared__ float tseg[VC_SEG][64];
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r = min(blockIdx.x * 64 + j, P - 1);
    const int len = (S + VC_SEG - 1) / VC_SEG, s0 = g * len, s1 = min(S, s0 + len);
    const float n_ = nr[r], f_ = fr[r];
    float o[15];
#pragma unroll
    for (int c = 0; c < 15; ++c) o[c] = 0.f;
    float T = 1.f, acc = 0.f, depth = 0.f;
    const float4* base = reinterpret_cast<const float4*>(raw);
    for (int s = s0; s < s1; ++s) {
        const float4* p = base + vol_slot(r, s, S) * 4;
        const float4 q3 = p[3];
        const float a = q3.w;
        const float w = a * T;
        T *= (1.f - a + 1e-8f);
        acc += w;
        const float tv = linspace01(s, S);
        depth += w * (n_ * (1.f - tv) + f_ * tv);
        if (a != 0.f) {
            const float4 q0 = p[0], q1 = p[1], q2 = p[2];
            o[0] += w * q0.x; o[1] += w * q0.y; o[2] += w * q0.z; o[3] += w * q0.w;
            o[4] += w * q1.x; o[5] += w * q1.y; o[6] += w * q1.z; o[7] += w * q1.w;
            o[8] += w * q2.x; o[9] += w * q2.y; o[10] += w * q2.z; o[11] += w * q2.w;
            o[12] += w * q3.x; o[13] += w * q3.y; o[14] += w * q3.z;
        }
    }
    tseg[g][j] = T;
    __syncthreads();
    float pre = 1.f;
    for (int k = 0; k < g; ++k) pre *= tseg[k][j];
#pragma unroll
    for (int c = 0; c < 15; ++c) part[g][c][j] = o[c] * pre;
    part[g][15][j] = acc * pre;
    part[g][16][j] = depth * pre;
    __syncthreads();
    if (blockIdx.x * 64 + j >= P) return;
    const int w_ = perm ? perm[r] : r;                          // internal (sorted) ray -> caller's ray index
    float a_ = 0.f;
#pragma unroll
    for (int k = 0; k < VC_SEG; ++k) a_ += part[k][15][j];
    if (g == 15) {
        float d_ = 0.f;
#pragma unroll
        for (int k = 0; k < VC_SEG; ++k) d_ += part[k][16][j];
        if (out.acc) out.acc[w_] = a_;
        if (out.depth) out.depth[w_] = d_;
        return;
    }
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < VC_SEG; ++k) v += part[k][g][j];
    v += (1.f - a_) * bg;
    float* dst = g < 3 ? out.cpts : (g < 6 ? out.bpts : (g < 9 ? out.resd : (g < 12 ? out.norm : out.rgb)));
    if (dst) dst[3 * w_ + g % 3] = v;
}

inline dim3 grid_for(long long n) { return dim3((unsigned)((n + TPB - 1) / TPB)); }

}  // namespace

void launch_debug_aabb(const float* o, const float* d, int n, const float* bbox6, float* nr, float* fr, hipStream_t s) {
    if (n <= 0) return;
    Box6 b; for (int k = 0; k < 6; ++k) b.v[k] = bbox6[k];
    hipLaunchKernelGGL(debug_aabb_kernel, grid_for(n), dim3(TPB), 0, s, o, d, n, b, nr, fr);
}

void launch_debug_brdf(const float* p2l, const float* p2c, const float* nrm, const float* alb, const float* rough, int L, int N, const ra_config& cfg,
                       float* out, hipStream_t s) {
    if (L * N <= 0) return;
    hipLaunchKernelGGL(debug_brdf_kernel, grid_for((long long)L * N), dim3(TPB), 0, s, p2l, p2c, nrm, alb, rough, L, N, cfg, out);
}

void launch_trace_init(const TraceState& ts, int n, const int* n_dev, const ra_trace_params& p, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(trace_init_kernel, grid_for(n), dim3(TPB), 0, s, ts, n, n_dev, p.offset, p.relax);
}

void launch_trace_update(const TraceState& ts, const float* sdf, int n, const int* n_dev, int iter, const ra_trace_params& p,
                         hipStream_t s) {
    if (n <= 0) return;
    if (p.soft_shadow) hipLaunchKernelGGL(trace_update_kernel<true>, grid_for(n), dim3(TPB), 0, s, ts, sdf, n, n_dev, iter, p);
    else hipLaunchKernelGGL(trace_update_kernel<false>, grid_for(n), dim3(TPB), 0, s, ts, sdf, n, n_dev, iter, p);
}

void launch_surface_finish(const float* ray_o, const float* ray_d, const float* st, const float* occ, int P, float* surf,
                           float* depth, float* acc, int* hit_idx, int* hit_count, hipStream_t s, int* slot_of_ray, bool counter_is_zero) {
    if (!counter_is_zero) hipMemsetAsync(hit_count, 0, sizeof(int), s);
    if (P <= 0) return;
    hipLaunchKernelGGL(surface_finish_kernel, grid_for(P), dim3(TPB), 0, s, ray_o, ray_d, st, occ, P, surf, depth, acc, hit_idx, hit_count, slot_of_ray);
}

void launch_slot_index(const int* hit_idx, const int* hit_count, int P, int* slot_of_ray, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(slot_index_kernel, grid_for(P), dim3(TPB), 0, s, hit_idx, hit_count, slot_of_ray);
}

void launch_emit_maps(const EmitMaps& m, hipStream_t s) {
    if (m.n_jobs <= 0 || m.P <= 0) return;
    hipLaunchKernelGGL(emit_maps_kernel, grid_for(m.end[m.n_jobs - 1]), dim3(TPB), 0, s, m);
}

void launch_surface_samples(const float* surf, const float* ray_d, const int* hit_idx, const int* hit_count, int P, int S,
                            float range, float* x, float* v, int* n_out, hipStream_t s) {
    if (P <= 0) { hipMemsetAsync(n_out, 0, sizeof(int), s); return; }
    hipLaunchKernelGGL(surface_samples_kernel, grid_for((long long)P * S), dim3(TPB), 0, s, surf, ray_d, hit_idx, hit_count, S, range, x, v, n_out);
}

void launch_surface_composite(const float* raw, int C, int S, const int* hit_count, int P, int relight, const ra_config& cfg,
                              const SurfaceMaps& m, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(surface_composite_kernel, grid_for(P), dim3(TPB), 0, s, raw, C, S, hit_count, relight, cfg, m);
}

void launch_key_lights(const float* probes, int n, int ph, int pw, const float* ldir, const float* area, int L, float share, int kmax,
                       int accumulate, float* smax, unsigned char* key, hipStream_t s) {
    hipLaunchKernelGGL(key_lights_kernel, dim3(1), dim3(TPB), (size_t)L * sizeof(float), s, probes, n, ph, pw, ldir, area, L, share, kmax, accumulate, smax, key);
}

void launch_light_dirs(const float* xyz, int L, float* ldir, hipStream_t s) {
    hipLaunchKernelGGL(light_dirs_kernel, grid_for(L), dim3(TPB), 0, s, xyz, L, ldir);
}

void launch_shadow_gen(const ShadowGen& g, int P, hipStream_t s, bool counter_is_zero) {
    if (!counter_is_zero) hipMemsetAsync(g.ray_count, 0, sizeof(int), s);
    if (P <= 0) return;
    const long long groups = ((long long)P + 63) / 64;
    const int lchunks = (g.L + SG_LIGHTS - 1) / SG_LIGHTS;
    hipLaunchKernelGGL(shadow_gen_kernel, dim3((unsigned)(groups * lchunks)), dim3(TPB), 0, s, g);
}

// ------------------------------------------------------------------------------------------ N1: ground plane
// ray / plane hit: moller_trumbore (mesh_utils.py:710-738) on a triangle of the plane reduces to
// t = -((o - orig).N) / (d.N + eps) with N = |a|^2 n (a: the triangle's random in-plane edge, which only rescales eps)
__global__ void ground_hit_kernel(GroundIn g, float* __restrict__ t_out, float* __restrict__ surf, float* __restrict__ depth,
                                  float* __restrict__ norm_slots, int* __restrict__ hit_idx, int* __restrict__ hit_count) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    bool hit = false;
    if (i < g.P) {
        const float o[3] = {g.ray_o[3 * i], g.ray_o[3 * i + 1], g.ray_o[3 * i + 2]};
        const float d[3] = {g.ray_d[3 * i], g.ray_d[3 * i + 1], g.ray_d[3 * i + 2]};
        const float num = (o[0] - g.orig[0]) * g.n[0] + (o[1] - g.orig[1]) * g.n[1] + (o[2] - g.orig[2]) * g.n[2];
        const float den = d[0] * g.n[0] + d[1] * g.n[1] + d[2] * g.n[2] + 1e-8f;
        const float t = -num / den;
        t_out[i] = t;
#pragma unroll
        for (int c = 0; c < 3; ++c) { surf[3 * i + c] = o[c] + t * d[c]; norm_slots[3 * i + c] = g.n[c]; }
        depth[i] = fminf(fmaxf(t, -g.env_r), g.env_r);
        hit = g.acc[i] > 0.f;
    }
    const unsigned long long m = __ballot(hit);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(hit_count, __popcll(m));
    base = __shfl(base, 0);
    if (hit) hit_idx[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

// one wave per traced pixel, lanes stride the lights (render_ground :488-523)
__global__ __launch_bounds__(TPB) void ground_shade_kernel(GroundShade in, ra_config cfg) {
    const int n = min(*in.hit_count, in.g.P);
    const int h = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (h >= n) return;
    const int r = in.hit_idx[h];
    const float t = in.t[r];
    const float sx = in.surf[3 * r] - in.g.orig[0], sy = in.surf[3 * r + 1] - in.g.orig[1], sz = in.surf[3 * r + 2] - in.g.orig[2];
    const float dist = t <= 0.f ? 1e9f : sqrtf(sx * sx + sy * sy + sz * sz);          // looking up: no ground (:497)
    const float w = fminf(fmaxf((dist - in.g.env_r) / in.g.env_r, 0.f), 1.f);
    float alb[3] = {in.g.albedo[0], in.g.albedo[1], in.g.albedo[2]};
    if (in.g.attach_envmap) {
        const float d[3] = {in.g.ray_d[3 * r], in.g.ray_d[3 * r + 1], in.g.ray_d[3 * r + 2]};
        sample_probe(in.probe, in.ph, in.pw, d, alb);
    }
    float sum[3] = {0.f, 0.f, 0.f};
    float vis_sum = 0.f;                                       // cfg.vis_shade_map: sum over the lights of lvis (1) / ldot (2)
    for (int l = lane; l < in.L; l += 64) {
        const float ld[3] = {in.ldir[3 * l], in.ldir[3 * l + 1], in.ldir[3 * l + 2]};
        float ldot = ld[0] * in.g.n[0] + ld[1] * in.g.n[1] + ld[2] * in.g.n[2];           // not clamped (:504)
        const float lv = in.lvis[(size_t)h * in.L + l] * (1.f - w) + w;                  // :505
        float Lr[3];
        sample_probe(in.probe, in.ph, in.pw, ld, Lr);
        if (cfg.only_visibility) { ldot = 1.f; Lr[0] = Lr[1] = Lr[2] = (Lr[0] + Lr[1] + Lr[2]) / 3.f; }      // :516-519
        vis_sum += cfg.vis_shade_map == 1 ? lv : ldot;
        const float k = lv * ldot * in.light_area[l];
#pragma unroll
        for (int c = 0; c < 3; ++c) sum[c] += k * Lr[c];
        if (in.lvis_out) in.lvis_out[(size_t)r * in.L + l] = lv;
        if (in.ldot_out) in.ldot_out[(size_t)r * in.L + l] = ldot;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vis_sum += __shfl_xor(vis_sum, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = sum[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) {
            const float rgb = alb[c] / PI_F * a;
            const float sh = a * cfg.shading_albedo / PI_F;
            if (in.rgb) in.rgb[3 * r + c] = cfg.tonemapping ? srgb(rgb) : rgb;
            if (in.albedo) in.albedo[3 * r + c] = alb[c];
            if (in.shade) in.shade[3 * r + c] = (cfg.vis_shade_map ? vis_sum / (float)in.L : sh) * in.g.shading_multiplier;     // :537-539
            if (in.spec) in.spec[3 * r + c] = sh / 20.f;
        }
    }
}

// novel-light re-shade of the ground layer (novel_light_sphere_tracing.py:70-99): one wave per frame pixel, lanes stride the
// lights, up to MAXP probes per pass share the 4 KB of cached visibility / cosine a pixel reads
__global__ __launch_bounds__(TPB) void ground_reshade_kernel(GroundReshade in, int q0, int nq) {
    const int p = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (p >= in.P) return;
    float sum[MAXP][3];
#pragma unroll
    for (int q = 0; q < MAXP; ++q) sum[q][0] = sum[q][1] = sum[q][2] = 0.f;
    for (int l = lane; l < in.L; l += 64) {
        const float ld[3] = {in.ldir[3 * l], in.ldir[3 * l + 1], in.ldir[3 * l + 2]};     // normalize(xyz - 0) (:76)
        const float k = in.lvis[(size_t)p * in.L + l] * in.ldot[(size_t)p * in.L + l] * in.light_area[l];
        for (int q = 0; q < nq; ++q) {
            float Lr[3];
            sample_probe(in.probes + (size_t)(q0 + q) * in.ph * in.pw * 3, in.ph, in.pw, ld, Lr);
#pragma unroll
            for (int c = 0; c < 3; ++c) sum[q][c] += k * Lr[c];
        }
    }
    const float d[3] = {in.ray_d[3 * p], in.ray_d[3 * p + 1], in.ray_d[3 * p + 2]};
    for (int q = 0; q < nq; ++q) {
        float alb[3];
        if (in.attach_envmap) {
            if (in.images) sample_probe(in.images + (size_t)(q0 + q) * in.ih * in.iw * 3, in.ih, in.iw, d, alb);
            else sample_probe(in.probes + (size_t)(q0 + q) * in.ph * in.pw * 3, in.ph, in.pw, d, alb);
        } else {
            alb[0] = in.albedo_map[3 * p]; alb[1] = in.albedo_map[3 * p + 1]; alb[2] = in.albedo_map[3 * p + 2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = sum[q][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) {
                const size_t k = ((size_t)(q0 + q) * in.P + p) * 3 + c;
                if (in.rgb) in.rgb[k] = srgb(alb[c] / PI_F * a);
                if (in.albedo) in.albedo[k] = alb[c];
                if (in.shade) in.shade[k] = a / PI_F;
                if (in.spec) in.spec[k] = a / PI_F / 20.f;
            }
        }
    }
}

void launch_ground_reshade(const GroundReshade& in, hipStream_t s) {
    if (in.P <= 0) return;
    for (int q0 = 0; q0 < in.n_probes; q0 += MAXP) {
        const int nq = in.n_probes - q0 < MAXP ? in.n_probes - q0 : MAXP;
        hipLaunchKernelGGL(ground_reshade_kernel, grid_for((long long)in.P * 64), dim3(TPB), 0, s, in, q0, nq);
    }
}

__global__ void blend_ground_base_kernel(const float* __restrict__ ground, const float* __restrict__ acc, int F, int C, float* __restrict__ dst) {
    const long long k = (long long)blockIdx.x * TPB + threadIdx.x;
    if (k >= (long long)F * C) return;
    dst[k] = ground ? ground[k] * acc[k / C] : 0.f;
}

__global__ void blend_ground_human_kernel(const float* __restrict__ human, const long long* __restrict__ inds, const float* __restrict__ acc,
                                          int P, int C, float* __restrict__ dst) {
    const long long k = (long long)blockIdx.x * TPB + threadIdx.x;
    if (k >= (long long)P * C) return;
    const int j = (int)(k / C), c = (int)(k - (long long)j * C);
    const long long f = inds[j];
    dst[f * C + c] += human[k] * (1.f - acc[f]);
}

void launch_blend_ground(const float* ground, const float* human, const long long* inds, const float* acc, int F, int P, int C, float* dst,
                         hipStream_t s) {
    if (F <= 0) return;
    hipLaunchKernelGGL(blend_ground_base_kernel, grid_for((long long)F * C), dim3(TPB), 0, s, ground, acc, F, C, dst);
    if (human && P > 0) hipLaunchKernelGGL(blend_ground_human_kernel, grid_for((long long)P * C), dim3(TPB), 0, s, human, inds, acc, P, C, dst);
}

void launch_ground_hit(const GroundIn& g, float* t, float* surf, float* depth, float* norm_slots, int* hit_idx, int* hit_count, hipStream_t s) {
    hipMemsetAsync(hit_count, 0, sizeof(int), s);
    if (g.P <= 0) return;
    hipLaunchKernelGGL(ground_hit_kernel, grid_for(g.P), dim3(TPB), 0, s, g, t, surf, depth, norm_slots, hit_idx, hit_count);
}

void launch_ground_shade(const GroundShade& in, const ra_config& cfg, hipStream_t s) {
    if (in.g.P <= 0) return;
    hipLaunchKernelGGL(ground_shade_kernel, grid_for((long long)in.g.P * 64), dim3(TPB), 0, s, in, cfg);
}

// ------------------------------------------------------------------------------------------ N3: body state
// per vertex: template -> T pose (pose_points_to_tpose_points with the big-pose blend, blend_utils.py:290-300) -> posed
// (tpose_points_to_pose_points :303-313) -> world (pose_points_to_world_points :264-273); bone matrices staged in LDS
__global__ void lbs_verts_kernel(const float* __restrict__ tverts, const float* __restrict__ weights, const float* __restrict__ A,
                                 const float* __restrict__ big_A, const float* __restrict__ R, const float* __restrict__ Th, int n_verts,
                                 int n_bones, float* __restrict__ tpose, float* __restrict__ pverts, float* __restrict__ wverts) {
    extern __shared__ float sA[];                      // [n_bones][12] posed, then [n_bones][12] big pose
    float* sB = sA + n_bones * 12;
    for (int k = threadIdx.x; k < n_bones * 12; k += blockDim.x) {
        const int j = k / 12, e = k - j * 12;
        sA[k] = A[j * 16 + e];
        sB[k] = big_A[j * 16 + e];
    }
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_verts) return;
    float Ma[12], Mb[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Ma[e] = Mb[e] = 0.f;
    for (int j = 0; j < n_bones; ++j) {
        const float w = weights[(size_t)v * n_bones + j];
#pragma unroll
        for (int e = 0; e < 12; ++e) { Ma[e] += w * sA[j * 12 + e]; Mb[e] += w * sB[j * 12 + e]; }
    }
    const float t[3] = {tverts[3 * v] - Mb[3], tverts[3 * v + 1] - Mb[7], tverts[3 * v + 2] - Mb[11]};
    const float Rb[9] = {Mb[0], Mb[1], Mb[2], Mb[4], Mb[5], Mb[6], Mb[8], Mb[9], Mb[10]};
    float Ri[9];
    {   // adjugate / (det + 1e-8)  (torch_inverse_3x3, blend_utils.py:125-165)
        Ri[0] = Rb[4] * Rb[8] - Rb[7] * Rb[5]; Ri[3] = -Rb[3] * Rb[8] + Rb[6] * Rb[5]; Ri[6] = Rb[3] * Rb[7] - Rb[6] * Rb[4];
        Ri[1] = -Rb[1] * Rb[8] + Rb[7] * Rb[2]; Ri[4] = Rb[0] * Rb[8] - Rb[6] * Rb[2]; Ri[7] = -Rb[0] * Rb[7] + Rb[6] * Rb[1];
        Ri[2] = Rb[1] * Rb[5] - Rb[4] * Rb[2]; Ri[5] = -Rb[0] * Rb[5] + Rb[3] * Rb[2]; Ri[8] = Rb[0] * Rb[4] - Rb[3] * Rb[1];
        const float inv = 1.f / (Rb[0] * Ri[0] + Rb[1] * Ri[3] + Rb[2] * Ri[6] + 1e-8f);
#pragma unroll
        for (int i = 0; i < 9; ++i) Ri[i] *= inv;
    }
    float tp[3], pp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tp[r] = Ri[3 * r] * t[0] + Ri[3 * r + 1] * t[1] + Ri[3 * r + 2] * t[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) pp[r] = Ma[4 * r] * tp[0] + Ma[4 * r + 1] * tp[1] + Ma[4 * r + 2] * tp[2] + Ma[4 * r + 3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (tpose) tpose[3 * v + r] = tp[r];
        if (pverts) pverts[3 * v + r] = pp[r];
        if (wverts) wverts[3 * v + r] = pp[0] * R[3 * r] + pp[1] * R[3 * r + 1] + pp[2] * R[3 * r + 2] + Th[r];       // p @ R^T + Th
    }
}

// pytorch3d Meshes.verts_normals: every corner adds the cross product of ITS two edges; the incident corners of a vertex
// are listed (host-built, cached) in the order index_add visits them (corner 1 pass, corner 2 pass, corner 0 pass, faces
// ascending), so the fp32 sum is deterministic
__global__ void vert_normals_kernel(const float* __restrict__ verts, const int* __restrict__ faces, const int* __restrict__ adj_start,
                                    const int* __restrict__ adj, int n_verts, float* __restrict__ normals) {
    const int v = blockIdx.x * TPB + threadIdx.x;
    if (v >= n_verts) return;
    float n[3] = {0.f, 0.f, 0.f};
    for (int k = adj_start[v]; k < adj_start[v + 1]; ++k) {
        const int code = adj[k], f = code >> 2, corner = code & 3;
        const int ia = faces[3 * f + corner], ib = faces[3 * f + (corner + 1) % 3], ic = faces[3 * f + (corner + 2) % 3];
        // corner c: cross(v[c+1] - v[c], v[c+2] - v[c])
        const float e1[3] = {verts[3 * ib] - verts[3 * ia], verts[3 * ib + 1] - verts[3 * ia + 1], verts[3 * ib + 2] - verts[3 * ia + 2]};
        const float e2[3] = {verts[3 * ic] - verts[3 * ia], verts[3 * ic + 1] - verts[3 * ia + 1], verts[3 * ic + 2] - verts[3 * ia + 2]};
        n[0] += e1[1] * e2[2] - e1[2] * e2[1];
        n[1] += e1[2] * e2[0] - e1[0] * e2[2];
        n[2] += e1[0] * e2[1] - e1[1] * e2[0];
    }
    const float len = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-6f);      // F.normalize(eps=1e-6)
    normals[3 * v] = n[0] / len; normals[3 * v + 1] = n[1] / len; normals[3 * v + 2] = n[2] / len;
}

// get_bounds (data_utils.py:616-622): one workgroup, min / max over the points, then the padding
__global__ __launch_bounds__(1024) void bounds_kernel(const float* __restrict__ pts, int n, float padding, float* __restrict__ out) {
    __shared__ float smin[16][3], smax[16][3];
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], pts[3 * i + c]); hi[c] = fmaxf(hi[c], pts[3 * i + c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) { smin[wv][c] = lo[c]; smax[wv][c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = smin[0][threadIdx.x], b = smax[0][threadIdx.x];
        for (int w = 1; w < 16; ++w) { a = fminf(a, smin[w][threadIdx.x]); b = fmaxf(b, smax[w][threadIdx.x]); }
        out[threadIdx.x] = a - padding;
        out[3 + threadIdx.x] = b + padding;
    }
}

// Bone transforms of one frame on the device, in float64 like the reference's numpy twin (get_rigid_transformation_and_joints,
// data_utils.py:1004-1069 / smplx.lbs.batch_rigid_transform): per joint a Rodrigues rotation (angle = |r + 1e-8|, entries rounded to
// float32 as batch_rodrigues returns them), the chain of 4 x 4 products along the kinematic tree (parents in topological order), then
// transforms[:, :3, 3] -= transforms[:, :3, :3] @ joint.  One wave: the chain is 51 dependent products whatever the width; lane e < 16
// owns element e of the running product.  Also the global rotation cv2.Rodrigues(Rh) and Th -> RT (12 floats).
// in: [poses 3J | tjoints 3J | big_A 16J | Rh 3 | Th 3 | parents J (int bits)], the block ra_pose_frame stages.
__global__ __launch_bounds__(64) void bone_transforms_kernel(const float* __restrict__ in, int J, float* __restrict__ A, float* __restrict__ joints,
                                                             float* __restrict__ RT) {
#pragma clang fp contract(off)                        // operation by operation as the host code it replaces: no fused multiply-adds
    extern __shared__ double sh[];                    // T[J][16] then chain[J][16]
    double* Tm = sh;
    double* chain = sh + (size_t)J * 16;
    const float* poses = in;
    const float* tj = in + 3 * J;
    const float* Rh = in + 3 * J + 3 * J + 16 * J;
    const float* Th = Rh + 3;
    const int* parents = reinterpret_cast<const int*>(Th + 3);
    const int lane = threadIdx.x;
    for (int j = lane; j < J; j += 64) {
        const float* r = poses + 3 * j;
        const double x = (double)r[0] + 1e-8, y = (double)r[1] + 1e-8, z = (double)r[2] + 1e-8;
        const double angle = sqrt(x * x + y * y + z * z);
        const double d[3] = {r[0] / angle, r[1] / angle, r[2] / angle};
        const double K[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
        const double sn = sin(angle), cs = cos(angle);
        double* t = Tm + (size_t)j * 16;
        const int par = parents[j];
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) {
                double kk = 0;
                for (int k = 0; k < 3; ++k) kk += K[3 * a + k] * K[3 * k + b];
                t[4 * a + b] = (double)(float)((a == b ? 1.0 : 0.0) + sn * K[3 * a + b] + (1.0 - cs) * kk);
            }
            t[4 * a + 3] = (double)tj[3 * j + a] - (j ? (double)tj[3 * par + a] : 0.0);
        }
        t[12] = t[13] = t[14] = 0.0; t[15] = 1.0;
    }
    __syncthreads();
    if (lane < 16) chain[lane] = Tm[lane];
    __syncthreads();
    for (int j = 1; j < J; ++j) {
        if (lane < 16) {
            const double* pm = chain + (size_t)parents[j] * 16;
            const double* t = Tm + (size_t)j * 16;
            const int r = lane >> 2, k = lane & 3;
            double a = 0;
            for (int m = 0; m < 4; ++m) a += pm[4 * r + m] * t[4 * m + k];
            chain[(size_t)j * 16 + lane] = a;
        }
        __syncthreads();
    }
    for (int j = lane; j < J; j += 64) {
        double* o = chain + (size_t)j * 16;
        for (int r = 0; r < 3; ++r) joints[3 * j + r] = (float)o[4 * r + 3];
        for (int r = 0; r < 4; ++r) {
            double rot = 0;
            for (int k = 0; k < 3; ++k) rot += o[4 * r + k] * (double)tj[3 * j + k];
            o[4 * r + 3] -= rot;          // transforms[..., 3] -= transforms @ [joint, 0]
        }
        for (int e = 0; e < 16; ++e) A[(size_t)j * 16 + e] = (float)o[e];
    }
    if (lane == 0) {                      // cv2.Rodrigues(Rh)
        const double r[3] = {Rh[0], Rh[1], Rh[2]};
        const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        double Rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (th >= 1e-12) {
            const double k[3] = {r[0] / th, r[1] / th, r[2] / th};
            const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
            const double sn = sin(th), cs = cos(th);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double kk = 0;
                    for (int m = 0; m < 3; ++m) kk += K[3 * i + m] * K[3 * m + j];
                    Rd[3 * i + j] = (i == j ? 1.0 : 0.0) + sn * K[3 * i + j] + (1.0 - cs) * kk;
                }
        }
        for (int e = 0; e < 9; ++e) RT[e] = (float)Rd[e];
        for (int e = 0; e < 3; ++e) RT[9 + e] = Th[e];
    }
}

void launch_bone_transforms(const float* staged, int J, float* A, float* joints, float* RT, hipStream_t s) {
    hipLaunchKernelGGL(bone_transforms_kernel, dim3(1), dim3(64), (size_t)J * 32 * sizeof(double), s, staged, J, A, joints, RT);
}

void launch_lbs_verts(const float* tverts, const float* weights, const float* A, const float* big_A, const float* R, const float* Th,
                      int n_verts, int n_bones, float* tpose, float* pverts, float* wverts, hipStream_t s) {
    if (n_verts <= 0) return;
    hipLaunchKernelGGL(lbs_verts_kernel, dim3((n_verts + 255) / 256), dim3(256), (size_t)n_bones * 24 * sizeof(float), s, tverts, weights, A,
                       big_A, R, Th, n_verts, n_bones, tpose, pverts, wverts);
}

void launch_vert_normals(const float* verts, const int* faces, const int* adj_start, const int* adj, int n_verts, float* normals, hipStream_t s) {
    if (n_verts <= 0) return;
    hipLaunchKernelGGL(vert_normals_kernel, grid_for(n_verts), dim3(TPB), 0, s, verts, faces, adj_start, adj, n_verts, normals);
}

void launch_bounds(const float* pts, int n, float padding, float* bounds6, hipStream_t s) {
    hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(1024), 0, s, pts, n, padding, bounds6);
}

// ------------------------------------------------------------------------------------------ N4: envmap utilities
// shift_image (relight_utils.py:69-85): grid x = ((j + 0.5 + shift) mod W) / W * 2 - 1, grid_sample(align_corners=False, border)
__global__ void shift_envmap_kernel(const float* __restrict__ img, int H, int W, int C, float shift, float* __restrict__ out) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= H * W) return;
    const int y = k / W, x = k - y * W;
    float gx = fmodf((float)x + 0.5f + shift, (float)W);
    if (gx < 0.f) gx += (float)W;                        // python's % is non-negative
    gx = gx / (float)W * 2.f - 1.f;
    float ix = ((gx + 1.f) * W - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    const float fx = floorf(ix);
    const int x0 = (int)fx, x1 = x0 + 1;
    const float w1 = ix - fx, w0 = 1.f - w1;
    for (int c = 0; c < C; ++c) {
        float v = w0 * img[((size_t)y * W + x0) * C + c];
        if (x1 < W) v += w1 * img[((size_t)y * W + x1) * C + c];
        out[(size_t)k * C + c] = v;
    }
}

// add_light_probe (relight_utils.py:38-54): pixel (y, x) of the inset looks along gen_light_xyz(uH, uW)[y][x] (relight_utils.py:423-465)
// rotated into the world by the probe axes of gen_light_dir
__global__ void light_probe_kernel(ProbeInset p, const float* __restrict__ probe, float* __restrict__ rgb) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= p.uH * p.uW) return;
    const int y = k / p.uW, x = k - y * p.uW;
    const float lat_half = PI_F / p.uH / 2.f, lng_half = 2.f * PI_F / p.uW / 2.f;
    // torch.linspace(a, b, n)[i]: a + i * step for the first half, b - (n - 1 - i) * step for the second
    auto lin = [](float a, float b, int n, int i) { const float st = (b - a) / (float)(n - 1); return i < n / 2 ? a + i * st : b - (n - 1 - i) * st; };
    const float lat = p.uH > 1 ? lin(PI_F / 2.f - lat_half, -PI_F / 2.f + lat_half, p.uH, y) : PI_F / 2.f - lat_half;
    const float lng = p.uW > 1 ? lin(PI_F - lng_half, -PI_F + lng_half, p.uW, x) : PI_F - lng_half;
    float v[3] = {cosf(lat) * cosf(lng), cosf(lat) * sinf(lng), sinf(lat)};
    const float nn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-8f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = (v[0] * p.axes[3 * r] + v[1] * p.axes[3 * r + 1] + v[2] * p.axes[3 * r + 2]) / nn;   // v @ axes^T
    float c[3];
    sample_probe(probe, p.ph, p.pw, d, c);
    float* o = rgb + ((size_t)y * p.W + x) * 3;
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
}

// the in-place box growth of one render chunk (sphere_tracing_renderer.py:1020-1022): wbounds[0] -= m, wbounds[1] += m, torch's fp32 arithmetic
__global__ void grow_bounds_kernel(float* __restrict__ wb, float m) {
    const int k = threadIdx.x;
    if (k < 6) wb[k] = k < 3 ? wb[k] - m : wb[k] + m;
}
void launch_grow_bounds(float* wbounds6, float margin, hipStream_t s) { hipLaunchKernelGGL(grow_bounds_kernel, dim3(1), dim3(64), 0, s, wbounds6, margin); }

void launch_shift_envmap(const float* img, int H, int W, int C, float shift, float* out, hipStream_t s) {
    if (H * W <= 0) return;
    hipLaunchKernelGGL(shift_envmap_kernel, grid_for(H * W), dim3(TPB), 0, s, img, H, W, C, shift, out);
}

void launch_light_probe(const ProbeInset& p, const float* probe, float* rgb, hipStream_t s) {
    if (p.uH * p.uW <= 0) return;
    hipLaunchKernelGGL(light_probe_kernel, grid_for(p.uH * p.uW), dim3(TPB), 0, s, p, probe, rgb);
}

// ------------------------------------------------------------------------------------------ N2: ray generation
// one pixel's ray and box interval (data_utils.py:827-845, 860-875): direction in fp64, rounded once; the rest fp32
__device__ __forceinline__ bool pixel_ray(const RayCam& c, int pix, float o[3], float d[3], float& nr, float& fr) {
    const double x = (double)(pix % c.W), y = (double)(pix / c.W);
    double pc[3], pw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pc[k] = x * c.Kinv[3 * k] + y * c.Kinv[3 * k + 1] + c.Kinv[3 * k + 2] - c.T[k];    // xy1 @ inv(K).T - T
#pragma unroll
    for (int k = 0; k < 3; ++k) pw[k] = pc[0] * c.R[k] + pc[1] * c.R[3 + k] + pc[2] * c.R[6 + k] - c.o[k];          // (.) @ R - ray_o
    const double inv = 1.0 / sqrt(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) { d[k] = (float)(pw[k] * inv); o[k] = (float)c.o[k]; }
    const float nd = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float t_near = -3.4e38f, t_far = 3.4e38f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = d[k] / nd;
        if (v < 1e-5f && v > -1e-10f) v = 1e-5f;
        if (v > -1e-5f && v < 1e-10f) v = -1e-5f;
        const float lo = c.bdev ? c.bdev[k] : c.bmin[k], hi = c.bdev ? c.bdev[3 + k] : c.bmax[k];     // the box may still be on its way (N3's bounds kernel)
        const float t0 = (lo - o[k]) / v, t1 = (hi - o[k]) / v;
        t_near = fmaxf(t_near, fminf(t0, t1));
        t_far = fminf(t_far, fmaxf(t0, t1));
    }
    nr = t_near / nd / nd;          // get_full_near_far and get_near_far both divide by |d| (= 1)
    fr = t_far / nd / nd;
    return t_near < t_far;
}

__global__ void ray_mask_kernel(RayCam c, unsigned char* __restrict__ mask) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= c.H * c.W) return;
    float o[3], d[3], nr, fr;
    mask[i] = pixel_ray(c, i, o, d, nr, fr) ? 1 : 0;
}

__global__ void ray_emit_kernel(RayCam c, const int* __restrict__ pix_idx, const int* __restrict__ count, float* __restrict__ ro,
                                float* __restrict__ rd, float* __restrict__ near_, float* __restrict__ far_) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= *count) return;
    float o[3], d[3], nr, fr;
    pixel_ray(c, pix_idx[i], o, d, nr, fr);
#pragma unroll
    for (int k = 0; k < 3; ++k) { ro[3 * i + k] = o[k]; rd[3 * i + k] = d[k]; }
    near_[i] = nr;
    far_[i] = fr;
}

size_t gen_rays_temp_bytes(int n) {
    size_t bytes = 0;
    hipcub::DeviceSelect::Flagged(nullptr, bytes, hipcub::CountingInputIterator<int>(0), (const unsigned char*)nullptr, (int*)nullptr, (int*)nullptr, n);
    return bytes;
}

int launch_gen_rays(const RayCam& cam, unsigned char* mask, int* pix_idx, int* count_dev, void* temp, size_t temp_bytes,
                    float* ray_o, float* ray_d, float* near, float* far, hipStream_t s) {
    const int n = cam.H * cam.W;
    if (n <= 0) { hipMemsetAsync(count_dev, 0, sizeof(int), s); return 0; }
    hipLaunchKernelGGL(ray_mask_kernel, grid_for(n), dim3(TPB), 0, s, cam, mask);
    // stable selection: the in-box pixels in row-major order, as the reference's boolean mask indexing gives them
    if (hipcub::DeviceSelect::Flagged(temp, temp_bytes, hipcub::CountingInputIterator<int>(0), mask, pix_idx, count_dev, n, s) != hipSuccess) return 1;
    hipLaunchKernelGGL(ray_emit_kernel, grid_for(n), dim3(TPB), 0, s, cam, pix_idx, count_dev, ray_o, ray_d, near, far);
    return 0;
}

size_t sort_hits_temp_bytes(int P) {
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, P);
    return bytes;
}

int launch_sort_hits(const float* surf, const float* acc, int P, const float* bbox_min, unsigned* keys_in, unsigned* keys_out,
                     int* vals_in, int* hit_idx_out, void* temp, size_t temp_bytes, hipStream_t s) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(hit_keys_kernel, grid_for(P), dim3(TPB), 0, s, surf, acc, P, bbox_min[0], bbox_min[1], bbox_min[2], keys_in, vals_in);
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, hit_idx_out, P, 0, 32, s) == hipSuccess ? 0 : 1;
}

void launch_shadow_scatter(const float* occ, const int* ray_slot, const int* ray_count, int max_rays, float* lvis, hipStream_t s) {
    if (max_rays <= 0) return;
    hipLaunchKernelGGL(shadow_scatter_kernel, grid_for(max_rays), dim3(TPB), 0, s, occ, ray_slot, ray_count, max_rays, lvis);
}

void launch_shade(const ShadeIn& in, const ra_config& cfg, hipStream_t s) {
    if (in.n <= 0) return;
    hipLaunchKernelGGL(shade_kernel, grid_for((long long)in.n * 64), dim3(TPB), 0, s, in, cfg);
}

void launch_scatter_maps(const int* hit_idx, const int* hit_count, int P, int premultiply, const float* acc_full, const float* src,
                         int C, float* dst, int src_full, const int* perm, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(scatter_maps_kernel, grid_for((long long)P * C), dim3(TPB), 0, s, hit_idx, hit_count, premultiply, acc_full, src, C, dst, src_full, perm);
}

int launch_sort_rays(const float* ro, const float* rd, const float* nr, const float* fr, int P, const float* bbox_min, unsigned* keys_in,
                     unsigned* keys_out, int* vals_in, int* perm, void* temp, size_t temp_bytes, float* so, float* sd, float* sn, float* sf,
                     hipStream_t s, float near_min, float far_max) {
    if (P <= 0) return 0;
    float* origin_dev = nullptr;
    if (!bbox_min) {            // no box from the caller: the grid starts at the lower corner of the entry points (device side)
        origin_dev = reinterpret_cast<float*>(keys_out);        // 12 bytes of the sort's output buffer, consumed before the sort runs
        hipLaunchKernelGGL(entry_min_kernel, dim3(1), dim3(1024), 0, s, ro, rd, nr, P, origin_dev);
    }
    hipLaunchKernelGGL(ray_keys_kernel, grid_for(P), dim3(TPB), 0, s, ro, rd, nr, P, bbox_min ? bbox_min[0] : 0.f, bbox_min ? bbox_min[1] : 0.f,
                       bbox_min ? bbox_min[2] : 0.f, origin_dev, keys_in, vals_in);
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, perm, P, 0, 32, s) != hipSuccess) return 1;
    hipLaunchKernelGGL(gather_rays_kernel, grid_for(P), dim3(TPB), 0, s, perm, P, ro, rd, nr, fr, so, sd, sn, sf, near_min, far_max);
    return 0;
}

void launch_gather_shard_rays(const long long* idx, int n, const float* ro, const float* rd, const float* nr, const float* fr, float* so, float* sd,
                              float* sn, float* sf, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_shard_rays_kernel, grid_for(n), dim3(TPB), 0, s, idx, n, ro, rd, nr, fr, so, sd, sn, sf);
}

void launch_scatter_rows(const float* src, const long long* src_idx, const long long* dst_idx, long long n, int C, float* dst, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(scatter_rows_kernel, grid_for(n * C), dim3(TPB), 0, s, src, src_idx, dst_idx, n, C, dst);
}

void launch_accumulate(const int* count, unsigned long long* dst, hipStream_t s) {
    hipLaunchKernelGGL(accumulate_kernel, dim3(1), dim3(1), 0, s, count, dst);
}

void launch_gather_rows(const int* hit_idx, const int* hit_count, int P, const float* src, int C, float* dst, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, grid_for((long long)P * C), dim3(TPB), 0, s, hit_idx, hit_count, src, C, dst);
}

__global__ void iota_kernel(int* __restrict__ idx, int n, int* __restrict__ count) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i == 0 && count) *count = n;
    if (i < n) idx[i] = i;
}
void launch_iota(int* idx, int n, int* count, hipStream_t s) {
    hipLaunchKernelGGL(iota_kernel, grid_for(n > 0 ? n : 1), dim3(TPB), 0, s, idx, n, count);
}

// 4x4 helpers of world_to_bigpose_transform (base_network.py:338-359); affine_inverse = [[R^T, -R^T t], [last row kept]]
__device__ __forceinline__ void affine_inverse4(const float* A, float* o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[4 * r + c] = A[4 * c + r];
        o[4 * r + 3] = -(A[r] * A[3] + A[4 + r] * A[7] + A[8 + r] * A[11]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) o[12 + c] = A[12 + c];
}
__device__ __forceinline__ void matmul4(const float* A, const float* B, float* o) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) o[4 * r + c] = A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c] + A[4 * r + 2] * B[8 + c] + A[4 * r + 3] * B[12 + c];
}
__global__ void bigpose_compose_kernel(const float* __restrict__ mats, const float* __restrict__ d2, int n, float inv2r2, const float* __restrict__ R,
                                       const float* __restrict__ Th, int invert, float* __restrict__ out) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    // bottom-right element of a blended 4x4: sum_k w_k / (sum_k w_k + eps)  (base_network.py:288-290; bone matrices end in 1)
    float ws = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ws += expf(-d2[3 * i + k] * inv2r2);
    const float sden = ws + 1.1920928955078125e-07f;
    float sb = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) sb += expf(-d2[3 * i + k] * inv2r2) / sden;
    float A[16], B[16], P2W[16], t0[16], t1[16], t2[16];
#pragma unroll
    for (int e = 0; e < 12; ++e) { A[e] = mats[(size_t)i * 24 + e]; B[e] = mats[(size_t)i * 24 + 12 + e]; }
    A[12] = A[13] = A[14] = 0.f; A[15] = sb;
    B[12] = B[13] = B[14] = 0.f; B[15] = sb;
#pragma unroll
    for (int r = 0; r < 3; ++r) { P2W[4 * r] = R[3 * r]; P2W[4 * r + 1] = R[3 * r + 1]; P2W[4 * r + 2] = R[3 * r + 2]; P2W[4 * r + 3] = Th[r]; }
    P2W[12] = P2W[13] = P2W[14] = 0.f; P2W[15] = 1.f;
    affine_inverse4(P2W, t0);         // w2p
    affine_inverse4(A, t1);           // p2t
    matmul4(B, t1, t2);               // (t2b @ p2t) @ w2p: torch evaluates the chain left to right
    matmul4(t2, t0, t1);
    if (invert) { affine_inverse4(t1, t0); for (int e = 0; e < 16; ++e) out[(size_t)i * 16 + e] = t0[e]; }
    else for (int e = 0; e < 16; ++e) out[(size_t)i * 16 + e] = t1[e];
}
void launch_bigpose_compose(const float* mats, const float* d2, int n, float blend_radius, const float* R, const float* Th, int invert, float* out,
                            hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(bigpose_compose_kernel, grid_for(n), dim3(TPB), 0, s, mats, d2, n, 1.f / (2.f * blend_radius * blend_radius), R, Th, invert, out);
}

void launch_fill(float* p, size_t n, float v, hipStream_t s) {
    if (n == 0) return;
    if (v == 0.f) { hipMemsetAsync(p, 0, n * sizeof(float), s); return; }
    hipLaunchKernelGGL(fill_kernel, grid_for((long long)n), dim3(TPB), 0, s, p, n, v);
}

void launch_volume_samples(const float* ray_o, const float* ray_d, const float* near_, const float* far_, int P, int S, float* x,
                           float* v, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(volume_samples_kernel, grid_for((long long)((P + 63) & ~63) * S), dim3(TPB), 0, s, ray_o, ray_d, near_, far_, P, S, x, v);
}

void launch_volume_composite(const float* raw, int C, const float* near_, const float* far_, int P, int S, float bg,
                             const ra_render_out& out, const int* perm, hipStream_t s) {
    if (P <= 0) return;
    if (C != 16) { fprintf(stderr, "relightableavatar: volume compositing expects 16 raw channels, got %d\n", C); abort(); }
    hipLaunchKernelGGL(volume_composite_kernel, dim3((P + 63) / 64), dim3(64 * VC_SEG), 0, s, raw, near_, far_, P, S, bg, out, perm);
}
