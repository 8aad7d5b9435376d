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

namespace {

constexpr int KNN_THREADS = 256;
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

__global__ void fold_bias_kernel(const float* __restrict__ W, int ld, int col0, int ncond, const float* __restrict__ cond,
                                 const float* __restrict__ bias, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 256) return;
    float a = bias[r];
    for (int c = 0; c < ncond; ++c) a += W[(size_t)r * ld + col0 + c] * cond[c];
    out[r] = a;
}

// ---------------------------------------------------------------------------------------------
// BVH over the posed vertices, rebuilt per frame on the stream by ONE workgroup:
//   bbox -> 30-bit Morton codes -> bitonic sort in LDS -> sorted float4 (xyz, index) ->
//   leaf boxes (8 points) -> upper levels of an implicit complete binary tree (heap order).
// ---------------------------------------------------------------------------------------------
constexpr int BVH_THREADS = 1024;
constexpr int BVH_LEAF = 8;
constexpr int BVH_MAXN = 16384;

__device__ __forceinline__ unsigned expand10(unsigned v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(BVH_THREADS) void bvh_build_kernel(const float4* __restrict__ pv, int n, float4* __restrict__ pts,
                                                                float* __restrict__ boxes /* 2*nl x 6 scratch */,
                                                                float4* __restrict__ pairs, int nl) {
    __shared__ unsigned long long keys[BVH_MAXN];     // code << 32 | index (static: 128 KB of the CU's 160 KB)
    __shared__ float red[6][BVH_THREADS / 64];
    __shared__ float bb[6];
    const int tid = threadIdx.x;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    // bbox
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
    // Morton keys
    for (int i = tid; i < np2; i += BVH_THREADS) {
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
    // bitonic sort (ascending)
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += BVH_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // sorted points + leaf boxes
    for (int i = tid; i < n; i += BVH_THREADS) {
        const unsigned id = (unsigned)(keys[i] & 0xffffffffull);
        float4 v = pv[id];
        v.w = __int_as_float((int)id);
        pts[i] = v;
    }
    __syncthreads();
    for (int l = tid; l < nl; l += BVH_THREADS) {
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        for (int e = 0; e < BVH_LEAF; ++e) {
            const int i = l * BVH_LEAF + e;
            if (i < n) {
                const unsigned id = (unsigned)(keys[i] & 0xffffffffull);
                const float4 v = pv[id];
                lo[0] = fminf(lo[0], v.x); lo[1] = fminf(lo[1], v.y); lo[2] = fminf(lo[2], v.z);
                hi[0] = fmaxf(hi[0], v.x); hi[1] = fmaxf(hi[1], v.y); hi[2] = fmaxf(hi[2], v.z);
            }
        }
        float* b = boxes + (size_t)(nl + l) * 6;
        b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
    }
    __syncthreads();
    __threadfence_block();
    for (int lvl = nl >> 1; lvl >= 1; lvl >>= 1) {      // nodes [lvl, 2*lvl)
        for (int k = tid; k < lvl; k += BVH_THREADS) {
            const int node = lvl + k;
            const float* a = boxes + (size_t)(2 * node) * 6;
            const float* c = a + 6;
            float* b = boxes + (size_t)node * 6;
#pragma unroll
            for (int d = 0; d < 3; ++d) { b[d] = fminf(a[d], c[d]); b[3 + d] = fmaxf(a[3 + d], c[3 + d]); }
            pairs[3 * node + 0] = make_float4(a[0], a[1], a[2], a[3]);
            pairs[3 * node + 1] = make_float4(a[4], a[5], c[0], c[1]);
            pairs[3 * node + 2] = make_float4(c[2], c[3], c[4], c[5]);
        }
        __syncthreads();
        __threadfence_block();
    }
}

__device__ __forceinline__ float box_dist2(const float p[3], float lx, float ly, float lz, float hx, float hy, float hz) {
    const float dx = fmaxf(fmaxf(lx - p[0], p[0] - hx), 0.f);
    const float dy = fmaxf(fmaxf(ly - p[1], p[1] - hy), 0.f);
    const float dz = fmaxf(fmaxf(lz - p[2], p[2] - hz), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

// top-3 insertion; ties resolved towards the lower vertex index (what an ascending scan gives)
__device__ __forceinline__ void knn_insert(float d, int id, float& d0, float& d1, float& d2, int& i0, int& i1, int& i2) {
    if (d < d2 || (d == d2 && id < i2)) {
        if (d < d0 || (d == d0 && id < i0)) { d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = id; }
        else if (d < d1 || (d == d1 && id < i1)) { d2 = d1; i2 = i1; d1 = d; i1 = id; }
        else { d2 = d; i2 = id; }
    }
}

constexpr int BVH_STACK = 14;     // depth of a 2048-leaf tree + 2

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

template <bool BVH>
__global__ __launch_bounds__(KNN_THREADS) void hdq_coarse_kernel(FrameState fr, RaySet rs, int n_launch, float th, float inv2r2,
                                                                  HdqOut out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[BVH ? 16 : VT * 16];   // brute force: a vertex tile
    const int n = rs.n_dev ? min(*rs.n_dev, n_launch) : n_launch;
    const int base = blockIdx.x * KNN_THREADS;
    if (base >= n) return;               // whole block idle (uniform)
    if (blockIdx.x == 0 && threadIdx.x == 0 && out.counters) atomicAdd(&out.counters->n_coarse, (unsigned long long)n);
    const int i = base + threadIdx.x;
    const bool live = i < n;
    float x[3] = {0.f, 0.f, 0.f};
    if (live) ray_point(rs, i, x);
    // world -> pose: (x - Th) R   (blend_utils.py:252-261)
    const float xt[3] = {x[0] - fr.Th[0], x[1] - fr.Th[1], x[2] - fr.Th[2]};
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = xt[0] * fr.R[c] + xt[1] * fr.R[3 + c] + xt[2] * fr.R[6 + c];

    float d0 = 3.0e38f, d1 = 3.0e38f, d2 = 3.0e38f;
    int i0 = 0, i1 = 0, i2 = 0;    // valid indices also for idle lanes (they run the table look-ups below)
    if (BVH) {
        // exact 3-NN by a WAVE-COOPERATIVE nearest-first traversal: the 64 queries of a wave are
        // neighbouring rays, so they share one stack (held across the lanes of a VGPR: entry i
        // lives in lane i, read with v_readlane, written with a lane-select) and every node / leaf is loaded once
        // per wave with a wave-uniform address; a subtree is entered when ANY lane still needs it.
        const int lane = threadIdx.x & 63;
        const int nl = fr.bvh_leaves;
        if (!live) { d0 = d1 = d2 = -1.f; }            // idle lanes: every test fails, nothing is inserted
        int vstk = 1;                                    // lane 0 holds the root
        int sp = 1;
        while (sp > 0) {
            sp = __builtin_amdgcn_readfirstlane(sp) - 1;
            const int node = __builtin_amdgcn_readlane(vstk, sp);
            if (node >= nl) {
                const float* bx = fr.bvh_boxes + (size_t)node * 6;
                const float db = box_dist2(p, bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]);
                if (__ballot(db * 0.99999f <= d2) == 0ull) continue;
                const int b = (node - nl) * BVH_LEAF;
                const int ne = min(BVH_LEAF, fr.n_verts - b);
                for (int k = 0; k < ne; ++k) {
                    const float4 v = fr.bvh_pts[b + k];
                    const float dx = p[0] - v.x, dy = p[1] - v.y, dz = p[2] - v.z;
                    const float d = dx * dx + dy * dy + dz * dz;
                    knn_insert(d, __float_as_int(v.w), d0, d1, d2, i0, i1, i2);
                }
                continue;
            }
            const float4 a = fr.bvh_pairs[3 * node], b4 = fr.bvh_pairs[3 * node + 1], c4 = fr.bvh_pairs[3 * node + 2];
            const float dl = box_dist2(p, a.x, a.y, a.z, a.w, b4.x, b4.y);
            const float dr = box_dist2(p, b4.z, b4.w, c4.x, c4.y, c4.z, c4.w);
            const bool wl = dl * 0.99999f <= d2, wr = dr * 0.99999f <= d2;
            const unsigned long long ml = __ballot(wl), mr = __ballot(wr);
            if (ml != 0ull && mr != 0ull) {
                const int votes_l = __popcll(__ballot(wl && (dl <= dr || !wr)));
                const int votes_r = __popcll(__ballot(wr && (dr < dl || !wl)));
                const int nearc = votes_l >= votes_r ? 2 * node : 2 * node + 1;
                vstk = (lane == sp) ? (nearc ^ 1) : vstk;
                vstk = (lane == sp + 1) ? nearc : vstk;
                sp += 2;
            } else if (ml != 0ull) {
                vstk = (lane == sp) ? 2 * node : vstk;
                sp += 1;
            } else if (mr != 0ull) {
                vstk = (lane == sp) ? 2 * node + 1 : vstk;
                sp += 1;
            }
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
                const float d = dx * dx + dy * dy + dz * dz;
                if (d < d2) knn_insert(d, v0 + j, d0, d1, d2, i0, i1, i2);
            }
        }
    }
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
    {
        const float* t0 = fr.tverts + 3 * (size_t)ik[0];
#pragma unroll
        for (int k = 1; k < 3; ++k) {
            const float* tk = fr.tverts + 3 * (size_t)ik[k];
            const float ex = tk[0] - t0[0], ey = tk[1] - t0[1], ez = tk[2] - t0[2];
            if (!(ex * ex + ey * ey + ez * ez < th2)) { dk[k] = dk[0]; ik[k] = ik[0]; sk[k] = sk[0]; }
        }
    }
    float smpl = (sk[0] + sk[1] + sk[2]) / 3.f;
    smpl = (smpl < -th) ? smpl : fabsf(smpl);       // base_network.py:375
    const bool fine = live && (d0 < th2);
    if (live) {
        out.sdf[i] = smpl;
        if (out.dbg_sdf_batch) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { out.dbg_sdf_batch[3 * i + k] = sk[k]; out.dbg_nn_batch[3 * i + k] = ik[k]; out.dbg_d2[3 * i + k] = dk[k]; }
        }
    }
    // compaction: one atomic per wave
    const unsigned long long m = __ballot(fine);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    int wbase = 0;
    if (lane == 0) wbase = atomicAdd(out.fine_count, __popcll(m));
    wbase = __shfl(wbase, 0);
    if (!fine) return;
    const int slot = wbase + __popcll(m & ((1ull << lane) - 1ull));
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
    hipLaunchKernelGGL(fold_bias_kernel, dim3(1), dim3(256), 0, s, W, ld, col0, ncond, cond, bias, out);
}

int bvh_leaf_count(int n_verts) {
    if (n_verts > BVH_MAXN) return 0;     // node ids must fit 12 bits + the sort must fit LDS
    int nl = 1;
    while (nl * BVH_LEAF < n_verts) nl <<= 1;
    return nl;
}

void launch_bvh_build(const float4* pverts4, int n_verts, float4* bvh_pts, float4* bvh_pairs, int n_leaves, hipStream_t s) {
    int np2 = 1;
    while (np2 < n_verts) np2 <<= 1;
    // boxes scratch lives behind the pairs array (3 float4 per node for 2*nl nodes, boxes need 2*nl*6 floats)
    float* boxes = reinterpret_cast<float*>(bvh_pairs + (size_t)3 * n_leaves);
    hipLaunchKernelGGL(bvh_build_kernel, dim3(1), dim3(BVH_THREADS), 0, s, pverts4, n_verts, bvh_pts, boxes, bvh_pairs, n_leaves);
}

void launch_hdq_coarse(const FrameState& fr, const RaySet& rs, int n, float th, float blend_radius, const HdqOut& out,
                       hipStream_t s) {
    hipMemsetAsync(out.fine_count, 0, sizeof(int), s);
    if (n <= 0) return;
    const float inv2r2 = 1.f / (2.f * blend_radius * blend_radius);
    const dim3 grid((n + KNN_THREADS - 1) / KNN_THREADS);
    if (fr.bvh_leaves > 0) hipLaunchKernelGGL(hdq_coarse_kernel<true>, grid, dim3(KNN_THREADS), 0, s, fr, rs, n, th, inv2r2, out);
    else hipLaunchKernelGGL(hdq_coarse_kernel<false>, grid, dim3(KNN_THREADS), 0, s, fr, rs, n, th, inv2r2, out);
}
