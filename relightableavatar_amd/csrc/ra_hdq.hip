// K1/K2: hierarchical distance query, coarse level + skinning warp (gfx950).
//
//   reference: geodesic_knn            lib/utils/sample_utils.py:103-162  (+ pytorch3d knn_points, K=3)
//              world_to_bigpose        lib/networks/deform/base_network.py:238-336
//              blend / inverse / warps lib/utils/blend_utils.py:125-165,212-218,252-313
//              coarse distance rule    lib/networks/deform/base_network.py:374-375
//
// One thread per query point.  Posed vertices stream through LDS in tiles of 2048 float4;
// every lane of a wave reads the same vertex (LDS broadcast), keeps its three nearest in
// registers, then resolves signs / the geodesic neighbour rule from L2-resident per-vertex
// tables.  Points closer than dist_th to the body are compacted (wave ballot + one atomic per
// wave) into the fine list together with their big-pose position, which is what the fused MLP
// kernel consumes — no host round trip (the reference synchronises on mask.sum().item(),
// net_utils.py:387).
// Skinning: A_bw = sum_k w_k * (sum_j weights[nn_k][j] A_j); the inner sum is a per-vertex table
// built once per frame (vert_blend_kernel), so a point gathers 3x24 floats instead of 3x52.
#include "ra_kernels.hpp"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int KNN_THREADS = 256;

#ifdef RA_COARSE_TS
// test builds (tools/coarse_timestamps.py): the first 64 workgroups of each of the last 64 coarse launches stamp the cycle counter of
// every wave at the phase boundaries of the search.  [launch & 63][workgroup][wave][8]: 0 start, 1 query point ready, 2 seed found,
// 3 seed leaf scanned, 4 sweep done, 5 merged (split variants), 6 end of kernel, 7 = leaves scanned << 32 | groups of 4 candidates that entered the insert path << 12 | super boxes opened
__device__ long long ra_coarse_ts[64 * 64 * 16 * 8];
extern "C" int ra_coarse_read_timestamps(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ra_coarse_ts), sizeof(ra_coarse_ts)); }
#define RA_CSTAMP(k) do { if (tsp) { __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0) tsp[k] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define RA_CSTAMP(k) do { } while (0)
#endif
constexpr int VT = 2048;    // vertices per LDS tile (32 KB)

__device__ __forceinline__ void inv3d(const float R[9], float M[9]) {   // blend_utils.py:125-165
    M[0] = R[4] * R[8] - R[7] * R[5];
    M[3] = -R[3] * R[8] + R[6] * R[5];
    M[6] = R[3] * R[7] - R[6] * R[4];
    M[1] = -R[1] * R[8] + R[7] * R[2];
    M[4] = R[0] * R[8] - R[6] * R[2];
    M[7] = -R[0] * R[7] + R[6] * R[1];
    M[2] = R[1] * R[5] - R[4] * R[2];
    M[5] = -R[0] * R[5] + R[3] * R[2];
    M[8] = R[0] * R[4] - R[3] * R[1];
    const float D = R[0] * M[0] + R[1] * M[3] + R[2] * M[6];
    const float inv = 1.f / (D + 1e-8f);
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] *= inv;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ void vert_blend_kernel(const float* __restrict__ weights, const float* __restrict__ A, const float* __restrict__ big_A,
                                  int n_verts, int n_bones, float* __restrict__ vertA) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_verts) return;
    float acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.f;
    for (int j = 0; j < n_bones; ++j) {
        const float w = weights[(size_t)v * n_bones + j];
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            acc[e] += w * A[j * 16 + e];
            acc[12 + e] += w * big_A[j * 16 + e];
        }
    }
#pragma unroll
    for (int e = 0; e < 24; ++e) vertA[(size_t)v * 24 + e] = acc[e];
}

__global__ void pack_verts_kernel(const float* __restrict__ pv, int n, float4* __restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) out[v] = make_float4(pv[3 * v], pv[3 * v + 1], pv[3 * v + 2], 0.f);
}

// one wave per output row: coalesced reads of the row's ncond weights, lane-strided partial sums, one shuffle reduction
// (round 5: a thread per row walked its 156 weights alone, 40 us per launch, three launches per frame)
__global__ __launch_bounds__(256) void fold_bias_kernel(const float* __restrict__ W, int ld, int col0, int ncond, const float* __restrict__ cond,
                                                        const float* __restrict__ bias, float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= 256) return;
    const float* row = W + (size_t)r * ld + col0;
    float a = 0.f;
    for (int c = lane; c < ncond; c += 64) a = fmaf(row[c], cond[c], a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) out[r] = bias[r] + a;
}

// ---------------------------------------------------------------------------------------------
// Vertex boxes for the exact 3-NN, rebuilt per frame on the stream (bvh_rank_kernel + bvh_boxes_kernel below):
//   bbox -> 30-bit Morton codes -> the vertices' rank in (code, index) order -> sorted float4 (xyz, index) ->
//   leaf boxes (32 consecutive points) -> super boxes (8 consecutive leaves).
// A flat, wide, 3-level structure on purpose: queries are swept through it with wave-uniform
// control flow (hdq_coarse_kernel), so there are no dependent-load chains to wait on.
// ---------------------------------------------------------------------------------------------
constexpr int BVH_THREADS = 1024;
constexpr int BVH_LEAF = 32;
constexpr int BVH_FAN = 8;
constexpr int BVH_MAXN = 16384;
constexpr int BVH_MAXL = BVH_MAXN / BVH_LEAF;          // 512 leaves
constexpr int BVH_MAXS = BVH_MAXL / BVH_FAN;           // 64 super boxes

__device__ __forceinline__ unsigned expand10(unsigned v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// Per frame, two launches.
// 1. bvh_rank_kernel: the ORDER of the posed vertices — ascending (30-bit Morton code, vertex index) — by RANKING instead of sorting: every
//    workgroup computes the box of all vertices and all n keys itself (7 loads per thread: cheaper than a launch in between), then ranks its
//    own 64 vertices, rank = number of smaller keys, the n comparisons of a vertex dealt to the 16 waves of the workgroup (each key is read
//    once per wave, from LDS with a wave-uniform address).  O(n^2) work, 47 M comparisons for SMPL, but on all CUs: the single-workgroup
//    bitonic sorts this replaced (round 2: 91 barrier steps in LDS, 100 us of a 131 us build; round 5 first try: registers + shuffles, 78 us —
//    7 500 wave instructions on ONE CU) were the largest fixed cost of a frame.  Keys are unique (the index is in the low bits), so the
//    order is THE sorted order, whatever computes it (test_box_structure_is_morton_sorted).
// 2. bvh_boxes_kernel: leaf records, leaf boxes, super boxes in that order (one workgroup, 32 lanes per leaf).
constexpr int RANK_WAVES = BVH_THREADS / 64;      // 16

__global__ __launch_bounds__(BVH_THREADS) void bvh_rank_kernel(const float4* __restrict__ pv, int n, int* __restrict__ order) {
    __shared__ unsigned long long keys[BVH_MAXN];     // code << 32 | index (static: 128 KB of the CU's 160 KB)
    __shared__ float red[6][BVH_THREADS / 64];
    __shared__ float bb[6];
    __shared__ int cnt[RANK_WAVES][64];
    const int tid = threadIdx.x;
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = tid; i < n; i += BVH_THREADS) {
        const float4 v = pv[i];
        mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z);
        mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
    if ((tid & 63) == 0) for (int c = 0; c < 3; ++c) { red[c][tid >> 6] = mn[c]; red[3 + c][tid >> 6] = mx[c]; }
    __syncthreads();
    if (tid < 6) {
        float a = red[tid][0];
        for (int w = 1; w < BVH_THREADS / 64; ++w) a = tid < 3 ? fminf(a, red[tid][w]) : fmaxf(a, red[tid][w]);
        bb[tid] = a;
    }
    __syncthreads();
    const int n4 = (n + 3) & ~3;                       // the ranking loop reads four keys at a time: pad with keys larger than any real one
    for (int i = tid; i < n4; i += BVH_THREADS) {
        unsigned long long k = ~0ull;
        if (i < n) {
            const float4 v = pv[i];
            const float p[3] = {v.x, v.y, v.z};
            unsigned q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e = fmaxf(bb[3 + c] - bb[c], 1e-12f);
                q[c] = (unsigned)fminf(fmaxf((p[c] - bb[c]) / e * 1023.f, 0.f), 1023.f);
            }
            const unsigned code = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
            k = ((unsigned long long)code << 32) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    const int lane = tid & 63, g = tid >> 6;
    const int e = blockIdx.x * 64 + lane;
    const unsigned long long mine = e < n ? keys[e] : 0ull;
    // wave g compares against keys [j0, j1): a quarter-aligned sixteenth of the array
    const int chunk = ((n4 / 4 + RANK_WAVES - 1) / RANK_WAVES) * 4;
    const int j0 = g * chunk, j1 = min(n4, j0 + chunk);
    int c = 0;
    for (int j = j0; j < j1; j += 4) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&keys[j]);
        const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(&keys[j + 2]);
        c += (a.x < mine) + (a.y < mine) + (b.x < mine) + (b.y < mine);
    }
    cnt[g][lane] = c;
    __syncthreads();
    if (g == 0 && e < n) {
        int r = 0;
#pragma unroll
        for (int w = 0; w < RANK_WAVES; ++w) r += cnt[w][lane];
        order[r] = e;
    }
}

__global__ __launch_bounds__(BVH_THREADS) void bvh_boxes_kernel(const float4* __restrict__ pv, const int* __restrict__ order, int n,
                                                                float* __restrict__ leaves, float4* __restrict__ sbox, int nl, int ns) {
    __shared__ float lb[BVH_MAXL][6];
    const int tid = threadIdx.x, e = tid & 31;
    // the points of a leaf as x[32] | y[32] | z[32] | id[32] (the leaf scan reads groups of candidates with wave-uniform addresses),
    // padded with +inf points: (p - inf)^2 = inf never beats a finite bound; the leaf's box by a half-wave reduction
    for (int l = tid >> 5; l < nl; l += BVH_THREADS / 32) {
        const int i = l * BVH_LEAF + e;
        float4 v = make_float4(__int_as_float(0x7f800000), __int_as_float(0x7f800000), __int_as_float(0x7f800000), __int_as_float(0x7fffffff));
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        if (i < n) {
            const int id = order[i];
            v = pv[id];
            v.w = __int_as_float(id);
            lo[0] = hi[0] = v.x; lo[1] = hi[1] = v.y; lo[2] = hi[2] = v.z;
        }
        float* soa = leaves + (size_t)l * (4 * BVH_LEAF) + e;
        soa[0] = v.x; soa[BVH_LEAF] = v.y; soa[2 * BVH_LEAF] = v.z; soa[3 * BVH_LEAF] = v.w;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
        if (e == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { lb[l][c] = lo[c]; lb[l][3 + c] = hi[c]; }
        }
    }
    __syncthreads();
    for (int sidx = tid; sidx < ns; sidx += BVH_THREADS) {
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        for (int k = 0; k < BVH_FAN; ++k) {
            const int l = sidx * BVH_FAN + k;
            if (l < nl)
#pragma unroll
                for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], lb[l][c]); hi[c] = fmaxf(hi[c], lb[l][3 + c]); }
        }
        sbox[2 * sidx] = make_float4(lo[0], lo[1], lo[2], 0.f);
        sbox[2 * sidx + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        // the super box's 8 leaf boxes as 4 PAIR records of 12 floats (lo.x[2] lo.y[2] lo.z[2] hi.x[2] hi.y[2] hi.z[2]): the sweep tests two
        // leaf boxes per packed instruction.  Missing leaves of the last super box: an inverted box, infinitely far from everything.
        float* rec = reinterpret_cast<float*>(sbox + 2 * (size_t)ns) + (size_t)sidx * (6 * BVH_FAN);
        for (int k = 0; k < BVH_FAN; ++k) {
            const int l = sidx * BVH_FAN + k;
#pragma unroll
            for (int c = 0; c < 6; ++c) rec[12 * (k >> 1) + 2 * c + (k & 1)] = l < nl ? lb[l][c] : (c < 3 ? 3e38f : -3e38f);
        }
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// lane K of every 16-lane row, as a DPP source modifier (gfx90a+ row_newbcast); folds into the consuming VALU op
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, false));
}

__device__ __forceinline__ float box_dist2(const float p[3], float lx, float ly, float lz, float hx, float hy, float hz) {
    const float dx = fmaxf(fmaxf(lx - p[0], p[0] - hx), 0.f);
    const float dy = fmaxf(fmaxf(ly - p[1], p[1] - hy), 0.f);
    const float dz = fmaxf(fmaxf(lz - p[2], p[2] - hz), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

// squared distance with ONE rounding order for every scan (what the compiler's contraction of dx*dx + dy*dy + dz*dz has produced since
// round 1): exact-neighbour ties and the last bit of the coarse distance do not depend on which scan found the vertex
__device__ __forceinline__ float dist2(float dx, float dy, float dz) { return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, dy * dy)); }

// top-3 insertion; ties resolved towards the lower vertex index (what an ascending scan gives).
// Pure value selects under one wave-uniform guard: keeps the six state words in VGPRs (a 3-way
// branchy version made LLVM spill them to scratch behind a computed store address).
// dedup (wave-uniform): the lists may already hold this vertex (hinted start: the previous iteration's neighbours come round again
// when their leaf is scanned) — a vertex is never entered twice
__device__ __forceinline__ void knn_insert(float d, int id, float& d0, float& d1, float& d2, int& i0, int& i1, int& i2, bool dedup = false) {
    if (__builtin_expect(__ballot(d <= d2) == 0ull, 1)) return;      // common case: one compare + one scalar branch per candidate
    bool fresh = true;
    if (dedup) fresh = (id != i0) & (id != i1) & (id != i2);
    const bool c2 = fresh && (d < d2 || (d == d2 && id < i2));
    const bool c1 = fresh && (d < d1 || (d == d1 && id < i1));
    const bool c0 = fresh && (d < d0 || (d == d0 && id < i0));
    const float nd2 = c1 ? d1 : (c2 ? d : d2);
    const int ni2 = c1 ? i1 : (c2 ? id : i2);
    const float nd1 = c0 ? d0 : (c1 ? d : d1);
    const int ni1 = c0 ? i0 : (c1 ? id : i1);
    d0 = c0 ? d : d0;
    i0 = c0 ? id : i0;
    d1 = nd1; i1 = ni1; d2 = nd2; i2 = ni2;
}

__device__ __forceinline__ void ray_point(const RaySet& rs, int i, float x[3]) {
    if (rs.mode == 0) {
        x[0] = rs.x[3 * i]; x[1] = rs.x[3 * i + 1]; x[2] = rs.x[3 * i + 2];
    } else if (rs.mode == 1) {
        const float t = rs.t[i];
        x[0] = rs.o[3 * i] + t * rs.d[3 * i];
        x[1] = rs.o[3 * i + 1] + t * rs.d[3 * i + 1];
        x[2] = rs.o[3 * i + 2] + t * rs.d[3 * i + 2];
    } else {
        const float t = rs.t[i];
        const int p = rs.pix[i], l = rs.light[i];
        x[0] = rs.o[3 * p] + t * rs.ldir[3 * l];
        x[1] = rs.o[3 * p + 1] + t * rs.ldir[3 * l + 1];
        x[2] = rs.o[3 * p + 2] + t * rs.ldir[3 * l + 2];
    }
}

// SPLIT > 1 (small launches, < 1 wave per SIMD: the serial sweep of one wave IS the launch time): the SPLIT waves of a
// workgroup serve the SAME 64 queries and share out the opened super boxes; their three-bests are merged through LDS.
// Each wave prunes against its own (weaker) bound, which is still conservative, so the merged result is exact.
// HINT: the pass starts from the neighbours of the previous tracing iteration (RaySet.nn_hint with hint_valid); compile time, so that the
// unhinted passes (a trace's first iteration, the volume path) keep the leaner insert path (measured: +13 % on the volume path's launch
// when the flag was a run-time one)
template <bool BVH, int SPLIT, bool HINT = false>
__global__ __launch_bounds__(SPLIT == 1 ? KNN_THREADS : 64 * SPLIT) void hdq_coarse_kernel(FrameState fr, RaySet rs, int n_launch, float th, float inv2r2,
                                                                  HdqOut out, int dbg) {
    static_assert(SPLIT == 1 || (BVH && SPLIT >= 2 && SPLIT <= 16), "SPLIT > 1: the workgroup is SPLIT waves on the same 64 queries");
    constexpr int NT = SPLIT == 1 ? KNN_THREADS : 64 * SPLIT;      // threads per workgroup
    // BVH: super + leaf boxes (2 float4 each); brute force: a vertex tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[BVH ? 16 : VT * 16];       // the O(N) scan's vertex tile
#ifdef RA_COARSE_TS
    long long* tsp = blockIdx.x < 64 ? ra_coarse_ts + ((((dbg >> 8) & 63) * 64 + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 : nullptr;
    dbg &= 255;
    if (tsp && (threadIdx.x & 63) == 0) { tsp[1] = tsp[2] = tsp[3] = tsp[4] = tsp[5] = tsp[6] = tsp[7] = 0; }
    RA_CSTAMP(0);
#endif
    const int n = rs.n_dev ? min(*rs.n_dev, n_launch) : n_launch;
    const int base = blockIdx.x * (SPLIT == 1 ? KNN_THREADS : 64);
    if (base >= n) return;               // whole block idle (uniform)
    if (blockIdx.x == 0 && threadIdx.x == 0 && out.counters) atomicAdd(&out.counters->n_coarse, (unsigned long long)n);
    const int i = base + (SPLIT == 1 ? threadIdx.x : (threadIdx.x & 63));
    const bool live_q = i < n && !(rs.skip && rs.skip[i < n ? i : 0]);
    bool live = live_q;
    if (__syncthreads_count(live_q) == 0) return;       // nothing to query in this workgroup (uniform)
    float x[3] = {0.f, 0.f, 0.f};
    if (live) ray_point(rs, i, x);
    // world -> pose: (x - Th) R   (blend_utils.py:252-261)
    const float xt[3] = {x[0] - fr.Th[0], x[1] - fr.Th[1], x[2] - fr.Th[2]};
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = xt[0] * fr.R[c] + xt[1] * fr.R[3 + c] + xt[2] * fr.R[6 + c];

    RA_CSTAMP(1);
    float d0 = 3.0e38f, d1 = 3.0e38f, d2 = 3.0e38f;
    int i0 = 0, i1 = 0, i2 = 0;    // valid indices also for idle lanes (they run the table look-ups below)
    if (dbg & 1) { d0 = 1e-4f; d1 = 2e-4f; d2 = 3e-4f; i0 = 0; i1 = 1; i2 = 2; }
    else if (BVH) {
        // exact 3-NN by WAVE-UNIFORM sweeps over the flat box structure (the 64 queries of a wave
        // are neighbouring rays): a box is opened when ANY lane may still find a closer vertex in it
        // (conservative box distance vs the lane's current third-best), all loads have wave-uniform
        // addresses and no load depends on another one.  Seed: the leaf nearest to the wave's first
        // live lane is scanned first so that every lane starts with a finite bound.
        const int nl = fr.bvh_leaves, ns = fr.bvh_supers;
        if (!live) { d0 = d1 = d2 = -1.f; }            // idle lanes: every test fails, nothing is inserted
        const unsigned long long lm = __ballot(live);
        const int first = lm ? __ffsll((long long)lm) - 1 : 0;
        const int lane = threadIdx.x & 63;
        // Leaf scan on packed fp32: a leaf is stored as x[32] | y[32] | z[32] (bvh_boxes_kernel), a PAIR of candidates is three 8-byte
        // reads at wave-uniform addresses (scalar loads: the operands arrive in SGPR pairs) and 3 v_pk_add + v_pk_mul + 2 v_pk_fma for
        // 2 x 64 distances, one min + compare + not-taken branch per pair: ~4.5 issue slots per candidate (the earlier scan — leaf in
        // two float4 registers per lane, coordinates as DPP row broadcasts of the subtraction — took 11: a DPP op costs two slots).
        // d = fma(dz, dz, fma(dx, dx, dy * dy)) in both halves: the rounding the O(N) validation scan has (dist2 below).
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 px2 = {p[0], p[0]}, py2 = {p[1], p[1]}, pz2 = {p[2], p[2]};
        constexpr bool hinted = HINT;          // start from the previous iteration's neighbours
#ifdef RA_COARSE_TS
        int n_rare = 0;
#endif
        // Eight candidates per coordinate arrive as one s_load_dwordx8 (constant address space: the structure was written by an earlier
        // kernel); the next eight are requested after the first group of the current eight has been tested, so that the compiler's
        // lgkmcnt(0) before the first use does not wait for the request just issued.  Four candidates per branch: two independent
        // packed chains (no dependent-issue bubbles).
        typedef float f8 __attribute__((ext_vector_type(8)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef int i4 __attribute__((ext_vector_type(4)));
        // wave-uniform reads of the structure: scalar loads through the constant address space
        auto soa8 = [&](int off) __attribute__((always_inline)) { return *(const f8 __attribute__((address_space(4)))*)(fr.bvh_soa + off); };
        auto soa_id4 = [&](int off) __attribute__((always_inline)) { return *(const i4 __attribute__((address_space(4)))*)(fr.bvh_soa + off); };
        auto box4 = [&](int off) __attribute__((always_inline)) {                       // offset in floats from the first super box
            return *(const f4 __attribute__((address_space(4)))*)(reinterpret_cast<const float*>(fr.bvh_sbox) + off);
        };
        auto box1 = [&](int off) __attribute__((always_inline)) { return reinterpret_cast<const float*>(fr.bvh_sbox)[off]; };      // per-lane read (seed search)
        auto scan_leaf = [&](int l) __attribute__((always_inline)) {
            const int L = l * (4 * BVH_LEAF);
            f8 cx = soa8(L), cy = soa8(L + BVH_LEAF), cz = soa8(L + 2 * BVH_LEAF), nx = cx, ny = cy, nz = cz;
            static_for<0, 4>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                static_for<0, 2>([&](auto g_) {
                    constexpr int g = decltype(g_)::value;
                    const f2 ax = {cx[4 * g], cx[4 * g + 1]}, ay = {cy[4 * g], cy[4 * g + 1]}, az = {cz[4 * g], cz[4 * g + 1]};
                    const f2 bx = {cx[4 * g + 2], cx[4 * g + 3]}, by = {cy[4 * g + 2], cy[4 * g + 3]}, bz = {cz[4 * g + 2], cz[4 * g + 3]};
                    const f2 dxa = px2 - ax, dya = py2 - ay, dza = pz2 - az, dxb = px2 - bx, dyb = py2 - by, dzb = pz2 - bz;
                    f2 da = dya * dya, db = dyb * dyb;
                    da = __builtin_elementwise_fma(dxa, dxa, da);
                    db = __builtin_elementwise_fma(dxb, dxb, db);
                    da = __builtin_elementwise_fma(dza, dza, da);
                    db = __builtin_elementwise_fma(dzb, dzb, db);
                    if (__builtin_expect(__ballot(fminf(fminf(da.x, da.y), fminf(db.x, db.y)) <= d2) != 0ull, 0)) {
#ifdef RA_COARSE_TS
                        ++n_rare;
#endif
                        // the vertex ids are only needed here; near the surface this path is NOT rare (the 64 queries of a wave find their
                        // neighbours all over the leaf under them): one scalar load, not a vector-memory round trip per group
                        const i4 id = soa_id4(L + 3 * BVH_LEAF + 8 * q + 4 * g);
                        knn_insert(da.x, id.x, d0, d1, d2, i0, i1, i2, hinted);
                        knn_insert(da.y, id.y, d0, d1, d2, i0, i1, i2, hinted);
                        knn_insert(db.x, id.z, d0, d1, d2, i0, i1, i2, hinted);
                        knn_insert(db.y, id.w, d0, d1, d2, i0, i1, i2, hinted);
                    }
                    if constexpr (g == 0 && q < 3) { nx = soa8(L + 8 * (q + 1)); ny = soa8(L + BVH_LEAF + 8 * (q + 1)); nz = soa8(L + 2 * BVH_LEAF + 8 * (q + 1)); }
                });
                cx = nx; cy = ny; cz = nz;
            });
        };
        int n_scan = 0, n_open = 0;
        if (lm != 0ull) {
            // --- seed: the super box, then the leaf in it, nearest to the wave's first live query.  Lane j looks at box j (one box test and
            // a wave minimum instead of a loop over the boxes in every lane); ties -> lowest index.
            int seed = -1;
            if (hinted) {
                // the ray's neighbours of one iteration ago, re-measured at the new point (the same arithmetic the scans use, so a vertex
                // that comes round again has the identical distance) and ordered (distance, index)
                if (live) {
                    float e[3]; int h[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        h[k] = rs.hint_src ? rs.hint_src[3 * (size_t)rs.hint_src_index[i] + k] : rs.nn_hint[3 * (size_t)i + k];
                        const float4 v = fr.pverts4[h[k]];
                        e[k] = dist2(p[0] - v.x, p[1] - v.y, p[2] - v.z);
                    }
                    auto before = [&](int a, int b) { return e[a] < e[b] || (e[a] == e[b] && h[a] < h[b]); };
                    auto swap = [&](int a, int b) { const float te = e[a]; e[a] = e[b]; e[b] = te; const int th_ = h[a]; h[a] = h[b]; h[b] = th_; };
                    if (before(1, 0)) swap(0, 1);
                    if (before(2, 1)) swap(1, 2);
                    if (before(1, 0)) swap(0, 1);
                    d0 = e[0]; d1 = e[1]; d2 = e[2]; i0 = h[0]; i1 = h[1]; i2 = h[2];
                }
            } else {
                auto bc = [&](float v) __attribute__((always_inline)) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), first)); };
                const float q[3] = {bc(p[0]), bc(p[1]), bc(p[2])};
                auto wave_argmin = [&](float d) __attribute__((always_inline)) {
                    float m = d;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
                    return __ffsll((long long)__ballot(d == m)) - 1;
                };
                float d = 3.4e38f;
                if (lane < ns) { const int o = 8 * lane; d = box_dist2(q, box1(o), box1(o + 1), box1(o + 2), box1(o + 4), box1(o + 5), box1(o + 6)); }
                const int bi = wave_argmin(d);
                d = 3.4e38f;
                // leaf `lane` of that super box, from its pair record (a missing leaf is an inverted box: infinitely far)
                if (lane < BVH_FAN) { const int o = 8 * ns + bi * (6 * BVH_FAN) + 12 * (lane >> 1) + (lane & 1); d = box_dist2(q, box1(o), box1(o + 2), box1(o + 4), box1(o + 6), box1(o + 8), box1(o + 10)); }
                seed = bi * BVH_FAN + wave_argmin(d);
            }
            RA_CSTAMP(2);
            if (!hinted) {
                n_scan = 1;
                scan_leaf(seed);
            }
            RA_CSTAMP(3);
            // --- the sweep: a super box is opened when any lane may still find a closer vertex in it; its 8 leaf boxes are tested in pairs
            // (packed subtract / multiply-add on SGPR operands: 18 VALU per pair, 17 per box before), each leaf against the bounds as they
            // stand after the scans before it.  Boxes come through the scalar cache (no LDS staging, no barrier in the sweep).
            // SPLIT > 1: wave w of the workgroup sweeps super boxes w, w + SPLIT, ... (a static deal: every wave prunes with its own bounds).
#pragma unroll 1
            for (int j = SPLIT > 1 ? (int)(threadIdx.x >> 6) : 0; j < ns; j += SPLIT) {
                const f4 slo = box4(8 * j), shi = box4(8 * j + 4);
                if (__ballot(box_dist2(p, slo.x, slo.y, slo.z, shi.x, shi.y, shi.z) * 0.99999f <= d2) == 0ull) continue;
                ++n_open;
#pragma unroll 1
                for (int pr = 0; pr < BVH_FAN / 2; ++pr) {
                    const int ro = 8 * ns + j * (6 * BVH_FAN) + 12 * pr;
                    const f4 ra = box4(ro), rb = box4(ro + 4), rc = box4(ro + 8);          // lo.x[2] lo.y[2] | lo.z[2] hi.x[2] | hi.y[2] hi.z[2]
                    const f2 ax = f2{ra.x, ra.y} - px2, ay = f2{ra.z, ra.w} - py2, az = f2{rb.x, rb.y} - pz2;
                    const f2 bx = px2 - f2{rb.z, rb.w}, by = py2 - f2{rc.x, rc.y}, bz = pz2 - f2{rc.z, rc.w};
                    const f2 ex = {__builtin_fmaxf(__builtin_fmaxf(ax.x, bx.x), 0.f), __builtin_fmaxf(__builtin_fmaxf(ax.y, bx.y), 0.f)};
                    const f2 ey = {__builtin_fmaxf(__builtin_fmaxf(ay.x, by.x), 0.f), __builtin_fmaxf(__builtin_fmaxf(ay.y, by.y), 0.f)};
                    const f2 ez = {__builtin_fmaxf(__builtin_fmaxf(az.x, bz.x), 0.f), __builtin_fmaxf(__builtin_fmaxf(az.y, bz.y), 0.f)};
                    f2 dd = ex * ex;
                    dd = __builtin_elementwise_fma(ey, ey, dd);
                    dd = __builtin_elementwise_fma(ez, ez, dd);
                    dd *= 0.99999f;
                    const int l = j * BVH_FAN + 2 * pr;
                    if (l != seed && __ballot(dd.x <= d2) != 0ull) { scan_leaf(l); ++n_scan; }
                    if (l + 1 != seed && __ballot(dd.y <= d2) != 0ull) { scan_leaf(l + 1); ++n_scan; }
                }
            }
        }
        RA_CSTAMP(4);
#ifdef RA_COARSE_TS
        if (tsp && (threadIdx.x & 63) == 0) tsp[7] = ((long long)n_scan << 32) | ((long long)n_rare << 12) | n_open;
#endif
        if (SPLIT > 1) {
            __shared__ float md[SPLIT > 1 ? SPLIT - 1 : 1][3][64];
            __shared__ int mi[SPLIT > 1 ? SPLIT - 1 : 1][3][64];
            const int wv = threadIdx.x >> 6;
            if (wv > 0) {
                md[wv - 1][0][lane] = d0; md[wv - 1][1][lane] = d1; md[wv - 1][2][lane] = d2;
                mi[wv - 1][0][lane] = i0; mi[wv - 1][1][lane] = i1; mi[wv - 1][2][lane] = i2;
            }
            __syncthreads();
            if (wv == 0) {
#pragma unroll
                for (int w = 0; w < SPLIT - 1; ++w)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        // every wave scanned the seed leaf (for its bound): a vertex may come back more than once
                        const int id = mi[w][k][lane];
                        const bool dup = id == i0 || id == i1 || id == i2;
                        knn_insert(dup ? 3.4e38f : md[w][k][lane], id, d0, d1, d2, i0, i1, i2);
                    }
            } else {
                live = false;           // waves 1.. only helped with the search
            }
        }
        RA_CSTAMP(5);
        if ((dbg & 4) && out.counters && lane == 0) {
            atomicAdd(&out.counters->n_shadow_rays, (unsigned long long)n_scan);      // profiling aid: leaves scanned
            atomicAdd(&out.counters->n_hit_pixels, (unsigned long long)n_open);       //                supers opened
        }
    } else {
        // brute force: vertices stream through LDS, all lanes read the same vertex (broadcast)
        float4* sv = reinterpret_cast<float4*>(smem);
        for (int v0 = 0; v0 < fr.n_verts; v0 += VT) {
            const int nv = min(VT, fr.n_verts - v0);
            __syncthreads();
            for (int j = threadIdx.x; j < nv; j += KNN_THREADS) sv[j] = fr.pverts4[v0 + j];
            __syncthreads();
#pragma unroll 4
            for (int j = 0; j < nv; ++j) {
                const float4 v = sv[j];
                const float dx = p[0] - v.x, dy = p[1] - v.y, dz = p[2] - v.z;
                const float d = dist2(dx, dy, dz);
                knn_insert(d, v0 + j, d0, d1, d2, i0, i1, i2);
            }
        }
    }
    if (rs.nn_hint && live) {                // the next iteration's starting point (live: wave 0 of a split workgroup holds the merged result)
        int* hw = rs.nn_hint + 3 * (size_t)i;
        hw[0] = i0; hw[1] = i1; hw[2] = i2;
    }
    if (dbg & 2) { if (live) out.sdf[i] = d0 + d1 + d2 + (float)(i0 + i1 + i2); return; }
    // signed coarse distances (sample_utils.py:124-128) and the geodesic neighbour rule (:148-160)
    float dk[3] = {d0, d1, d2};
    int ik[3] = {i0, i1, i2};
    float sk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 v = fr.pverts4[ik[k]];
        const float* nn = fr.pnorm + 3 * (size_t)ik[k];
        const float dot = (p[0] - v.x) * nn[0] + (p[1] - v.y) * nn[1] + (p[2] - v.z) * nn[2];
        sk[k] = sqrtf(dk[k]) * sgn(dot);
    }
    const float th2 = th * th;
    if (!(dbg & 32)) {
        const float* t0 = fr.tverts + 3 * (size_t)ik[0];
#pragma unroll
        for (int k = 1; k < 3; ++k) {
            const float* tk = fr.tverts + 3 * (size_t)ik[k];
            const float ex = tk[0] - t0[0], ey = tk[1] - t0[1], ez = tk[2] - t0[2];
            if (!(ex * ex + ey * ey + ez * ez < th2)) { dk[k] = dk[0]; ik[k] = ik[0]; sk[k] = sk[0]; }
        }
    }
    float smpl = (sk[0] + sk[1] + sk[2]) / 3.f;
    if (dbg & 32) {
        // cfg.use_geodesic_filter = False: knn_with_filter (sample_utils.py:164-194) — one distance per point, sqrt(mean_k d_k^2) with the
        // sign of max_k sign((x - v_k) . n_k); the neighbours stay as found
        const float sg = fmaxf(fmaxf(sgn(sk[0]), sgn(sk[1])), sgn(sk[2]));       // sk[k] = sqrt(d_k) sign(dot_k): the same sign (0 with d_k = 0)
        smpl = sqrtf((dk[0] + dk[1] + dk[2]) / 3.f) * sg;
        sk[0] = sk[1] = sk[2] = smpl;
    }
    smpl = (smpl < -th) ? smpl : fabsf(smpl);       // base_network.py:375
    const bool fine = live && (d0 < th2);
    if (live) {
        out.sdf[i] = smpl;
        if (out.raw_zero && !fine) {                // Network.forward: zero raw channels outside dist_th (the full query writes the fine rows)
            float* rz = out.raw_zero + (size_t)i * out.raw_C;
            for (int c = 0; c < out.raw_C; ++c) rz[c] = 0.f;
        }
        if (out.dbg_sdf_batch) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { out.dbg_sdf_batch[3 * i + k] = sk[k]; out.dbg_nn_batch[3 * i + k] = ik[k]; out.dbg_d2[3 * i + k] = dk[k]; }
        }
    }
    // compaction: one atomic per workgroup (same-address atomics serialise in L2: per wave they cost as much as the
    // whole post-processing), wave offsets through LDS
    // (with HdqOut::key, the fine points of rays towards key lights form a second list the same way)
    __shared__ int wcount[NT / 64 + 1], wcount2[NT / 64 + 1];
    const bool fine2 = fine && out.key != nullptr && out.key[rs.light[i]] != 0;
    const unsigned long long m = __ballot(fine && !fine2), m2 = out.key ? __ballot(fine2) : 0ull;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { wcount[wv] = __popcll(m); wcount2[wv] = __popcll(m2); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0, tot2 = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { const int cnt = wcount[w]; wcount[w] = tot; tot += cnt; const int c2 = wcount2[w]; wcount2[w] = tot2; tot2 += c2; }
        wcount[NT / 64] = (dbg & 8) ? base : (tot ? atomicAdd(out.fine_count, tot) : 0);
        wcount2[NT / 64] = tot2 ? atomicAdd(out.fine_count2, tot2) : 0;
    }
    __syncthreads();
    RA_CSTAMP(6);
    if (!fine || (dbg & 16)) return;
    const int slot = fine2 ? wcount2[NT / 64] + wcount2[wv] + __popcll(m2 & ((1ull << lane) - 1ull))
                           : wcount[NT / 64] + wcount[wv] + __popcll(m & ((1ull << lane) - 1ull));
    // gaussian-weighted blend of the per-vertex transforms (base_network.py:287-296)
    float w[3], ws = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { w[k] = expf(-dk[k] * inv2r2); ws += w[k]; }
    ws += 1.1920928955078125e-07f;     // torch.finfo(float32).eps
    float M[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) M[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float wk = w[k] / ws;
        const float4* src = reinterpret_cast<const float4*>(fr.vertA + (size_t)ik[k] * 24);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float4 a = src[q];
            M[4 * q] += wk * a.x; M[4 * q + 1] += wk * a.y; M[4 * q + 2] += wk * a.z; M[4 * q + 3] += wk * a.w;
        }
    }
    const float AR[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
    float Ai[9];
    inv3d(AR, Ai);
    const float q[3] = {p[0] - M[3], p[1] - M[7], p[2] - M[11]};
    float tp[3], bp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tp[r] = Ai[3 * r] * q[0] + Ai[3 * r + 1] * q[1] + Ai[3 * r + 2] * q[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) bp[r] = M[12 + 4 * r] * tp[0] + M[12 + 4 * r + 1] * tp[1] + M[12 + 4 * r + 2] * tp[2] + M[12 + 4 * r + 3];
    if (fine2) {
        out.fine_idx2[slot] = i;
        out.bpts2[3 * slot] = bp[0]; out.bpts2[3 * slot + 1] = bp[1]; out.bpts2[3 * slot + 2] = bp[2];
        return;
    }
    out.fine_idx[slot] = i;
    out.bpts[3 * slot] = bp[0]; out.bpts[3 * slot + 1] = bp[1]; out.bpts[3 * slot + 2] = bp[2];
    if (out.mats) {
        float4* dst = reinterpret_cast<float4*>(out.mats + (size_t)slot * 24);
#pragma unroll
        for (int qd = 0; qd < 6; ++qd) dst[qd] = make_float4(M[4 * qd], M[4 * qd + 1], M[4 * qd + 2], M[4 * qd + 3]);
    }
    if (out.dbg_bpts) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { out.dbg_bpts[3 * i + c] = bp[c]; out.dbg_tpts[3 * i + c] = tp[c]; }
#pragma unroll
        for (int e = 0; e < 24; ++e) out.dbg_mats[(size_t)i * 24 + e] = M[e];
    }
}

}  // namespace

void launch_vert_blend(const float* weights, const float* A, const float* big_A, int n_verts, int n_bones, float* vertA,
                       hipStream_t s) {
    hipLaunchKernelGGL(vert_blend_kernel, dim3((n_verts + 255) / 256), dim3(256), 0, s, weights, A, big_A, n_verts, n_bones, vertA);
}

void launch_pack_verts(const float* pverts, int n_verts, float4* out, hipStream_t s) {
    hipLaunchKernelGGL(pack_verts_kernel, dim3((n_verts + 255) / 256), dim3(256), 0, s, pverts, n_verts, out);
}

void launch_fold_bias(const float* W, int ld, int col0, int ncond, const float* cond, const float* bias, float* out,
                      hipStream_t s) {
    hipLaunchKernelGGL(fold_bias_kernel, dim3(64), dim3(256), 0, s, W, ld, col0, ncond, cond, bias, out);
}

int bvh_leaf_count(int n_verts) {
    if (n_verts > BVH_MAXN) return 0;     // the sort must fit LDS; larger meshes fall back to the O(N) scan
    return (n_verts + BVH_LEAF - 1) / BVH_LEAF;
}
int bvh_super_count(int n_leaves) { return (n_leaves + BVH_FAN - 1) / BVH_FAN; }

void launch_bvh_build(const float4* pverts4, int n_verts, int* order, float* leaves, float4* sbox, int n_leaves, int n_supers, hipStream_t s) {
    hipLaunchKernelGGL(bvh_rank_kernel, dim3((n_verts + 63) / 64), dim3(BVH_THREADS), 0, s, pverts4, n_verts, order);
    hipLaunchKernelGGL(bvh_boxes_kernel, dim3(1), dim3(BVH_THREADS), 0, s, pverts4, order, n_verts, leaves, sbox, n_leaves, n_supers);
}

void launch_hdq_coarse(const FrameState& fr, const RaySet& rs, int n, float th, float blend_radius, const HdqOut& out,
                       hipStream_t s, bool geodesic) {
    // out.fine_count must be zero on entry (ra_api.cpp hands every pass a fresh counter of the chunk's pre-zeroed set)
    if (n <= 0) return;
    const float inv2r2 = 1.f / (2.f * blend_radius * blend_radius);
    const dim3 grid((n + KNN_THREADS - 1) / KNN_THREADS);
    // launches below ~1.5 waves per SIMD are latency-bound: spread each group of 64 queries over the 4 waves of a workgroup
    int dbg = 0, probe = 0, split_max = 196608;
#ifdef RA_TESTING            // profiling aids (test builds only, tools/build_variant.sh)
    // RA_COARSE_DBG: 1 skip the search, 2 skip everything after it.  RA_COARSE_PROBE: an extra launch in ablation mode `probe` on the
    // SAME inputs before every real launch (which then overwrites its outputs): the ablated time is read from a kernel trace
    static const int e_dbg = getenv("RA_COARSE_DBG") ? atoi(getenv("RA_COARSE_DBG")) : 0;
    static const int e_probe = getenv("RA_COARSE_PROBE") ? atoi(getenv("RA_COARSE_PROBE")) : 0;
    static const int e_split = getenv("RA_COARSE_SPLIT_MAX") ? atoi(getenv("RA_COARSE_SPLIT_MAX")) : 196608;
    dbg = e_dbg; probe = e_probe; split_max = e_split;
#endif
    if (!geodesic) dbg |= 32;        // knn_with_filter instead of geodesic_knn (ra_config.use_geodesic_filter)
#ifdef RA_COARSE_TS
    static int launch_no = 0;
    dbg |= (launch_no++ & 63) << 8;
#endif
    const bool hint = (rs.nn_hint != nullptr && rs.hint_valid != 0) || rs.hint_src != nullptr;
    for (int pass = probe ? 0 : 1; pass < 2; ++pass) {
        const int d = pass == 0 ? probe : dbg;
        if (fr.bvh_leaves > 0 && n <= split_max)
        {
            // more waves per 64 queries the smaller the launch (about 4-8 k waves in flight on the 1024 SIMDs)
            const int groups = (n + 63) / 64;
            if (hint) {
                if (n <= split_max / 8) hipLaunchKernelGGL((hdq_coarse_kernel<true, 16, true>), dim3(groups), dim3(1024), 0, s, fr, rs, n, th, inv2r2, out, d);
                else if (n <= split_max / 2) hipLaunchKernelGGL((hdq_coarse_kernel<true, 8, true>), dim3(groups), dim3(512), 0, s, fr, rs, n, th, inv2r2, out, d);
                else hipLaunchKernelGGL((hdq_coarse_kernel<true, 4, true>), dim3(groups), dim3(256), 0, s, fr, rs, n, th, inv2r2, out, d);
            } else {
                if (n <= split_max / 8) hipLaunchKernelGGL((hdq_coarse_kernel<true, 16>), dim3(groups), dim3(1024), 0, s, fr, rs, n, th, inv2r2, out, d);
                else if (n <= split_max / 2) hipLaunchKernelGGL((hdq_coarse_kernel<true, 8>), dim3(groups), dim3(512), 0, s, fr, rs, n, th, inv2r2, out, d);
                else hipLaunchKernelGGL((hdq_coarse_kernel<true, 4>), dim3(groups), dim3(256), 0, s, fr, rs, n, th, inv2r2, out, d);
            }
        }
        else if (fr.bvh_leaves > 0 && hint) hipLaunchKernelGGL((hdq_coarse_kernel<true, 1, true>), grid, dim3(KNN_THREADS), 0, s, fr, rs, n, th, inv2r2, out, d);
        else if (fr.bvh_leaves > 0) hipLaunchKernelGGL((hdq_coarse_kernel<true, 1>), grid, dim3(KNN_THREADS), 0, s, fr, rs, n, th, inv2r2, out, d);
        else hipLaunchKernelGGL((hdq_coarse_kernel<false, 1>), grid, dim3(KNN_THREADS), 0, s, fr, rs, n, th, inv2r2, out, d);
        if (pass == 0) hipMemsetAsync(out.fine_count, 0, sizeof(int), s);
    }
}
