// Fused MLP kernels for gfx950 (MI355X): residual-deformation + signed-distance networks evaluated
// back to back on a tile of 128 columns without ever writing activations to HBM.
//
//   reference: ResidualDeformation.forward  lib/networks/deform/base_network.py:34-42
//              MLP.forward                  lib/utils/net_utils.py:1263-1273
//              SignedDistanceNetwork        lib/networks/deform/base_network.py:78-97
//              SphereSignedDistanceField    lib/utils/net_utils.py:1337-1352
//              HDQ blend                    lib/networks/deform/base_network.py:374-382
//              forward_geometry (normals)   lib/networks/deform/base_network.py:456-494
//              material heads               lib/networks/relight/relight_network.py:45-47,97-104
//              RenderNetwork                lib/networks/deform/base_network.py:152-171
//
// Design (DESIGN.md "K3/K4"):
//  * one workgroup = 4 waves = one tile of TM=128 columns; activations live in LDS as bf16
//    [128][264] (row stride 528 B = 16*33 -> conflict-free ds_read_b128 B fragments);
//    73.7 KB LDS per workgroup -> two workgroups per CU, so one workgroup's VALU epilogue
//    (bias, activation, bf16 pack) overlaps the other's MFMA stream on the same SIMDs.
//  * D = W . X^T with v_mfma_f32_32x32x16_bf16: A = weights (rows n), B = activations (cols m).
//    Wave w owns output rows [64w, 64w+64) for all 128 columns: 2x4 accumulator tiles = 128 VGPRs.
//    Weight A-fragments are pre-packed in fragment order (ra_pack.cpp) and come straight from
//    L2 with one coalesced 1 KiB load per fragment; they never touch LDS.
//  * the D layout gives every lane 4 consecutive output rows of one column -> the epilogue
//    packs 4 bf16 and writes them with one ds_write_b64 as the next layer's B operand.
//  * skip connections re-compute the positional encoding into the tile instead of keeping it
//    (keeps LDS at 2 workgroups/CU); cond (156 constants per frame) is folded into biases.
//  * the "full" kernel carries 3 forward-mode tangent columns next to each primal column
//    (tile = 32 points x {value, d/dbx, d/dby, d/dbz}) to produce d sdf / d bpts exactly as
//    autograd does, then runs the material heads / colour net on the same tile.
#include "ra_common.hpp"

namespace {

constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_SOFTPLUS = 2;
constexpr float INV_2PI = 0.15915494309189535f;
// forward-mode tangent columns are carried scaled by TS for fp16 headroom (everything acting on
// them is linear); heads divide it back out.
constexpr float TS = 1.f / 16.f;

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// element type of weights/activations: bf16 or f16 (same MFMA rate, f16 has 3 more mantissa bits)
template <typename E> struct Tr;
template <> struct Tr<bf16> {
    typedef bf16x8 x8; typedef bf16x4 x4;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Tr<f16> {
    typedef f16x8 x8; typedef f16x4 x4;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// y = act(z), dy = act'(z)
template <int ACT, bool NEED_DY>
__device__ __forceinline__ float act_fn(float z, float& dy) {
    if (ACT == ACT_RELU) {
        if (NEED_DY) dy = z > 0.f ? 1.f : 0.f;
        return __builtin_amdgcn_fmed3f(z, 0.f, 3.0e38f);         // max(z, 0) in one instruction (fmaxf adds a canonicalising self-max)
    } else if (ACT == ACT_SOFTPLUS) {
        // nn.Softplus(beta=100): log(1+exp(100 z))/100 = max(z,0) + ln2/100 * log2(1 + exp2(-|100 z log2e|)),
        // overflow-free, two native base-2 transcendentals; torch's threshold=20 branch returns z where the
        // correction term is < 2.1e-11, i.e. identical in fp32.
        const float t = z * 144.26950408889634f;                  // 100 * log2(e)
        const float e = __builtin_amdgcn_exp2f(-fabsf(t));
        if (NEED_DY) dy = (t >= 0.f ? 1.f : e) * __builtin_amdgcn_rcpf(1.f + e);       // sigmoid(100 z); v_rcp_f32 (1 ulp) instead of an IEEE division
        return fmaf(__builtin_amdgcn_logf(1.f + e), 0.0069314718055994531f, __builtin_amdgcn_fmed3f(z, 0.f, 3.0e38f));
    } else {
        if (NEED_DY) dy = 1.f;
        return z;
    }
}

// A fragments (weights) of the first two k-steps of a wide layer, fetched ahead of time: issued
// before the previous layer's epilogue so their L2 latency hides behind the VALU work.
template <typename E> struct APre { typename Tr<E>::x8 a[2][2]; };

template <typename E, int KS>
__device__ __forceinline__ APre<E> gemm_prefetch(const typename Tr<E>::x8* __restrict__ wl, int wave, int lane) {
    APre<E> p;
    const typename Tr<E>::x8* a0p = wl + (size_t)((2 * wave + 0) * KS) * 64 + lane;
    const typename Tr<E>::x8* a1p = wl + (size_t)((2 * wave + 1) * KS) * 64 + lane;
    p.a[0][0] = a0p[0];
    p.a[0][1] = a1p[0];
    p.a[1][0] = a0p[KS > 1 ? 64 : 0];
    p.a[1][1] = a1p[KS > 1 ? 64 : 0];
    return p;
}

// acc[nt][mt] += W[64*wave + 32*nt .. +32, kcols] . X[32*mt .. +32, kcol0 .. kcol0 + 16*KS]^T
// Software pipelined: the weight fragments of k-step ks+2 are requested (global -> VGPR, L2
// resident, one coalesced 1 KiB load each) while k-step ks is multiplied, so two k-steps of loads
// are always in flight (counted vmcnt, never a drain inside the loop).
template <typename E, int MT>
__device__ __forceinline__ void kstep(f32x16 (&acc)[2][MT], typename Tr<E>::x8 a0, typename Tr<E>::x8 a1, const E* bp) {
    typedef typename Tr<E>::x8 x8;
    x8 b[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[mt] = *reinterpret_cast<const x8*>(bp + mt * 32 * XS);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[0][mt] = Tr<E>::mfma(a0, b[mt], acc[0][mt]);
        acc[1][mt] = Tr<E>::mfma(a1, b[mt], acc[1][mt]);
    }
}

template <typename E, int KS, int MT>
__device__ __forceinline__ void gemm_wide(f32x16 (&acc)[2][MT], const APre<E>& pre, const typename Tr<E>::x8* __restrict__ wl,
                                          const E* xs, int kcol0, int wave, int lane) {
    typedef typename Tr<E>::x8 x8;
    static_assert(KS % 2 == 0, "k-steps are processed in pairs");
    const x8* a0p = wl + (size_t)((2 * wave + 0) * KS) * 64 + lane;
    const x8* a1p = wl + (size_t)((2 * wave + 1) * KS) * 64 + lane;
    const E* bp = xs + (lane & 31) * XS + kcol0 + (lane >> 5) * 8;
    x8 c0 = pre.a[0][0], c1 = pre.a[0][1], n0 = pre.a[1][0], n1 = pre.a[1][1];
    // rolled loop, two k-steps per trip: the four fragment loads issued at the top of a trip are
    // consumed in the NEXT trip, i.e. they stay in flight across 16 MFMAs (+ the co-resident wave's)
#pragma unroll 1
    for (int ks = 0; ks < KS - 2; ks += 2) {
        const x8 f0 = a0p[(ks + 2) * 64];
        const x8 f1 = a1p[(ks + 2) * 64];
        const x8 g0 = a0p[(ks + 3) * 64];
        const x8 g1 = a1p[(ks + 3) * 64];
        __builtin_amdgcn_sched_barrier(0);      // keep the four loads at the top of the trip
        kstep<E, MT>(acc, c0, c1, bp + ks * 16);
        kstep<E, MT>(acc, n0, n1, bp + (ks + 1) * 16);
        c0 = f0; c1 = f1; n0 = g0; n1 = g1;
    }
    kstep<E, MT>(acc, c0, c1, bp + (KS - 2) * 16);
    kstep<E, MT>(acc, n0, n1, bp + (KS - 1) * 16);
}

template <int MT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][MT]) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
}

template <typename E>
__device__ __forceinline__ void store4(E* dst, float a, float b, float c, float d) {
    typename Tr<E>::x4 v;
    v[0] = (E)a; v[1] = (E)b; v[2] = (E)c; v[3] = (E)d;
    *reinterpret_cast<typename Tr<E>::x4*>(dst) = v;
}

// accumulators start at the bias of their row (primal columns) / at zero (tangent columns)
template <int MT, bool GRAD>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[2][MT], const float* __restrict__ bias, int wave, int lane) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 64 * wave + 32 * nt + 8 * q + 4 * (lane >> 5));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][mt][4 * q + j] = (GRAD && mt > 0) ? 0.f : bv[j];
        }
}

// write act(acc) as E into xs[:, 0..255] (the bias is already inside acc); GRAD: column group 0 is
// the primal, groups 1..3 hold tangents and get act'(z_primal) * t.
template <typename E, int ACT, bool GRAD, int MT>
__device__ __forceinline__ void epilogue_wide(const f32x16 (&acc)[2][MT], E* xs, int wave, int lane) {
    const int mrow = lane & 31;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n0 = 64 * wave + 32 * nt + 8 * q + 4 * (lane >> 5);
            float y[4], dy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = act_fn<ACT, GRAD>(acc[nt][0][4 * q + j], dy[j]);
            store4<E>(xs + mrow * XS + n0, y[0], y[1], y[2], y[3]);
#pragma unroll
            for (int mt = 1; mt < MT; ++mt) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (GRAD) {
                        o[j] = acc[nt][mt][4 * q + j] * dy[j];
                    } else {
                        float d_;
                        o[j] = act_fn<ACT, false>(acc[nt][mt][4 * q + j], d_);
                    }
                }
                store4<E>(xs + (mt * 32 + mrow) * XS + n0, o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// <=32 output rows; wave w handles columns [32w, 32w+32). Lanes 0..31 end up with rows 0..3 of
// column 32w+lane in acc[0..3].
template <typename E, int KS>
__device__ __forceinline__ f32x16 gemm_head(const typename Tr<E>::x8* __restrict__ wl, const E* xs, int kcol0, int colgrp, int lane) {
    typedef typename Tr<E>::x8 x8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const x8* ap = wl + lane;
    const E* bp = xs + (colgrp * 32 + (lane & 31)) * XS + kcol0 + (lane >> 5) * 8;
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
        const x8 a = ap[ks * 64];
        const x8 b = *reinterpret_cast<const x8*>(bp + ks * 16);
        acc = Tr<E>::mfma(a, b, acc);
    }
    return acc;
}

// positional encoding of one column into row[0 .. 3+6L) (embedder.py:26-37 channel order:
// x, then per frequency the 3 sines then the 3 cosines).  Thread half h writes half of the
// frequencies.  jcol == nullptr: primal values; otherwise the (scaled) tangent
//   d channel / d b_j = channel'(x_c) * J[c][j],  jcol = J[:, j] * TS.
// LO: 9 extra columns [3+6L, 3+6L+9) carry the rounding residual v - E(v) of the identity and
// frequency-0 channels (their weights are duplicated by the packer), so the first layer sees
// those inputs at ~2x the element precision.  Columns up to npad are zero-filled.
template <typename E, int L, bool LO>
__device__ __forceinline__ void pe_write_col(E* row, const float x[3], const float* jcol, int h, int npad) {
    float rev[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rev[c] = x[c] * INV_2PI;
    constexpr int NB = 3 + 6 * L;
    if (h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = jcol ? jcol[c] : x[c];
            const E hi = (E)v;
            row[c] = hi;
            if (LO) row[NB + c] = (E)(jcol ? 0.f : v - (float)hi);
        }
    }
    const int f0 = h * (L / 2), f1 = f0 + (L / 2);
#pragma unroll
    for (int f = f0; f < f1; ++f) {
        const float sc = (float)(1 << f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = rev[c] * sc;                  // revolutions; exact power-of-two scaling
            const float s = __builtin_amdgcn_sinf(a);
            const float co = __builtin_amdgcn_cosf(a);
            float vs, vc;
            if (jcol) { vs = sc * co * jcol[c]; vc = -sc * s * jcol[c]; }
            else { vs = s; vc = co; }
            const E hs = (E)vs, hc = (E)vc;
            row[3 + 6 * f + c] = hs;
            row[3 + 6 * f + 3 + c] = hc;
            if (LO && f == 0) {
                row[NB + 3 + c] = (E)(jcol ? 0.f : vs - (float)hs);
                row[NB + 6 + c] = (E)(jcol ? 0.f : vc - (float)hc);
            }
        }
    }
    if (h == 1)
        for (int c = NB + (LO ? 9 : 0); c < npad; ++c) row[c] = (E)0.f;
}

template <typename E>
struct Smem {
    E xs[TM * XS];          // activation tile
    float pts[TM * 4];      // big-pose point of each column (xyz, pad)
    float cpts[TM * 4];     // canonical point (bpts + resd)
    float misc[TM * 4];     // head outputs
    float jac[32 * 12];     // full kernel: d cpts / d bpts per point (row-major 3x3)
    int count;
};

// =============================================================================================
//  K3: HDQ fine query — resd + sdf (sdf only) on 128 points per tile, blend epilogue
// =============================================================================================
template <typename E, bool DEBUG>
__global__ __launch_bounds__(MLP_THREADS, 2) void mlp_sdf_kernel(GeoNet net, const void* __restrict__ wa_,
                                                                const float* __restrict__ ba, FrameState fr, MlpIO io) {
    typedef typename Tr<E>::x8 x8;
    const x8* __restrict__ wa = reinterpret_cast<const x8*>(wa_);
    __shared__ __attribute__((aligned(16))) Smem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sm.count = *io.count;
    __syncthreads();
    const int count = sm.count;
    if (blockIdx.x == 0 && tid == 0 && io.counters) atomicAdd(&io.counters->n_fine_sdf, (unsigned long long)count);
    f32x16 acc[2][4];

    const x8* __restrict__ wa_base = wa;
    for (int tile = blockIdx.x; tile * TM < count; tile += gridDim.x) {
        const int slot0 = tile * TM;
        // launder the weight base once per tile: the fragment loads are invariant across tiles and
        // LICM would otherwise hoist (and spill) them out of the persistent loop
        unsigned wz = 0;
        asm volatile("" : "+s"(wz));          // opaque zero: keeps the global address space of the pointer
        const x8* wa = wa_base + wz;
        // ---- load points
        if (tid < TM) {
            const int s = slot0 + tid;
            float x = 0.f, y = 0.f, z = 0.f;
            if (s < count) { x = io.bpts[3 * s]; y = io.bpts[3 * s + 1]; z = io.bpts[3 * s + 2]; }
            sm.pts[4 * tid] = x; sm.pts[4 * tid + 1] = y; sm.pts[4 * tid + 2] = z; sm.pts[4 * tid + 3] = 0.f;
        }
        __syncthreads();
        const int pm = tid >> 1, ph = tid & 1;
        // ---- residual deformation net
        APre<E> pre = gemm_prefetch<E, 4>(wa + net.r[0].w, wave, lane);
        { const float* p = sm.pts + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
          pe_write_col<E, 10, false>(sm.xs + pm * XS, x, nullptr, ph, 64); }
        __syncthreads();
        init_acc<4, false>(acc, fr.bias_r0, wave, lane);
        gemm_wide<E, 4, 4>(acc, pre, wa + net.r[0].w, sm.xs, 0, wave, lane);
        __syncthreads();
        pre = gemm_prefetch<E, 16>(wa + net.r[1].w, wave, lane);
        epilogue_wide<E, ACT_RELU, false, 4>(acc, sm.xs, wave, lane);
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l < 8; ++l) {
            init_acc<4, false>(acc, l == 4 ? fr.bias_r4 : ba + net.r[l].bias, wave, lane);
            gemm_wide<E, 16, 4>(acc, pre, wa + net.r[l].w, sm.xs, 0, wave, lane);
            __syncthreads();
            if (l == 4) {   // skip: cat([x, input]) — re-encode the input into cols 0..63 and accumulate
                pre = gemm_prefetch<E, 4>(wa + net.r4b.w, wave, lane);
                const float* p = sm.pts + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
                pe_write_col<E, 10, false>(sm.xs + pm * XS, x, nullptr, ph, 64);
                __syncthreads();
                gemm_wide<E, 4, 4>(acc, pre, wa + net.r4b.w, sm.xs, 0, wave, lane);
                __syncthreads();
            }
            if (l < 7) pre = gemm_prefetch<E, 16>(wa + net.r[l + 1].w, wave, lane);
            epilogue_wide<E, ACT_RELU, false, 4>(acc, sm.xs, wave, lane);
            __syncthreads();
        }
        {   // head: resd = tanh(z) * resd_limit; cpts = bpts + resd
            const f32x16 h = gemm_head<E, 16>(wa + net.rhead.w, sm.xs, 0, wave, lane);
            if (lane < 32) {
                const int m = wave * 32 + lane;
                const float* b = ba + net.rhead.bias;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float resd = tanhf(h[c] + b[c]) * io.resd_limit;
                    sm.cpts[4 * m + c] = sm.pts[4 * m + c] + resd;
                    if (DEBUG && io.dbg_resd && slot0 + m < count) io.dbg_resd[3 * (slot0 + m) + c] = resd;
                }
            }
        }
        __syncthreads();
        // ---- signed distance net
        pre = gemm_prefetch<E, 4>(wa + net.s[0].w, wave, lane);
        { const float* p = sm.cpts + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
          pe_write_col<E, 8, true>(sm.xs + pm * XS, x, nullptr, ph, 64); }
        __syncthreads();
        init_acc<4, false>(acc, ba + net.s[0].bias, wave, lane);
        gemm_wide<E, 4, 4>(acc, pre, wa + net.s[0].w, sm.xs, 0, wave, lane);
        __syncthreads();
        pre = gemm_prefetch<E, 16>(wa + net.s[1].w, wave, lane);
        epilogue_wide<E, ACT_SOFTPLUS, false, 4>(acc, sm.xs, wave, lane);
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l < 8; ++l) {
            init_acc<4, false>(acc, ba + net.s[l].bias, wave, lane);
            gemm_wide<E, 16, 4>(acc, pre, wa + net.s[l].w, sm.xs, 0, wave, lane);
            __syncthreads();
            if (l == 4) {   // skip: cat([x(205), input(51)]) / sqrt(2): the input part is a second K=64 pass
                pre = gemm_prefetch<E, 4>(wa + net.s4b.w, wave, lane);
                const float* p = sm.cpts + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
                pe_write_col<E, 8, true>(sm.xs + pm * XS, x, nullptr, ph, 64);
                __syncthreads();
                gemm_wide<E, 4, 4>(acc, pre, wa + net.s4b.w, sm.xs, 0, wave, lane);
                __syncthreads();
            }
            if (l < 7) pre = gemm_prefetch<E, 16>(wa + net.s[l + 1].w, wave, lane);
            epilogue_wide<E, ACT_SOFTPLUS, false, 4>(acc, sm.xs, wave, lane);
            __syncthreads();
        }
        if (DEBUG && io.dbg_feat) {     // feature rows of lin8 (test hook; the product path uses the full kernel)
            zero_acc<4>(acc);
            gemm_wide<E, 16, 4>(acc, gemm_prefetch<E, 16>(wa + net.sfeat.w, wave, lane), wa + net.sfeat.w, sm.xs, 0, wave, lane);
            const float* b = ba + net.sfeat.bias;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = 64 * wave + 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int s = slot0 + mt * 32 + (lane & 31);
                        if (s < count) io.dbg_feat[(size_t)s * 256 + n] = acc[nt][mt][r] + b[n];
                    }
        }
        {   // head: sdf, then the HDQ blend (base_network.py:374-382)
            const f32x16 h = gemm_head<E, 16>(wa + net.shead.w, sm.xs, 0, wave, lane);
            if (lane < 32) {
                const int s = slot0 + wave * 32 + lane;
                if (s < count) {
                    float d = h[0] + ba[net.shead.bias];
                    if (DEBUG && io.dbg_sdf) io.dbg_sdf[s] = d;
                    if (io.sdf) {
                        const int p = io.idx[s];
                        if (io.smooth) {
                            const float smpl = io.sdf[p];
                            const float r = fminf(fmaxf(fabsf(d) / io.dist_th, 0.f), 1.f);
                            d = smpl * r + d * (1.f - r);
                        }
                        io.sdf[p] = d;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// =============================================================================================
//  K4: full geometry forward with forward-mode tangents + material heads / colour net
//      tile = 32 points x {value, d/dbx, d/dby, d/dbz}
// =============================================================================================
__device__ __forceinline__ void inv3(const float R[9], float M[9]) {   // blend_utils.py:125-165
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

__device__ __forceinline__ void normalize3(float v[3]) {   // net_utils.py:1626-1628
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-8f;
    v[0] /= n; v[1] /= n; v[2] /= n;
}

__device__ __forceinline__ float sdf_to_occ_dev(float sdf, float beta) {   // net_utils.py:852-893
    const float x = -sdf;
    float sigma;
    if (x <= 0.f) sigma = 1.f / beta * (0.5f * expf(x / beta));
    else sigma = 1.f / beta * (1.f - 0.5f * expf(-x / beta));
    return 1.f - expf(-fmaxf(sigma, 0.f) * 0.005f);
}

template <typename E>
__global__ __launch_bounds__(MLP_THREADS, 2) void mlp_full_kernel(GeoNet net, MatNet mat, ColNet col,
                                                                 const void* __restrict__ wa_, const float* __restrict__ ba,
                                                                 FrameState fr, FullIO io) {
    typedef typename Tr<E>::x8 x8;
    const x8* __restrict__ wa = reinterpret_cast<const x8*>(wa_);
    __shared__ __attribute__((aligned(16))) Smem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sm.count = *io.count;
    __syncthreads();
    const int count = sm.count;
    if (blockIdx.x == 0 && tid == 0 && io.counters) atomicAdd(&io.counters->n_fine_full, (unsigned long long)count);
    f32x16 acc[2][4];
    // PE mapping: point pm, column group pg (0 primal, 1..3 tangents), frequency half ph
    const int pm = tid & 31, pg = (tid >> 5) & 3, ph = tid >> 7;

    const x8* __restrict__ wa_base = wa;
    for (int tile = blockIdx.x; tile * 32 < count; tile += gridDim.x) {
        const int slot0 = tile * 32;
        unsigned wz = 0;
        asm volatile("" : "+s"(wz));          // opaque zero: keeps the global address space of the pointer
        const x8* wa = wa_base + wz;
        if (tid < 32) {
            const int s = slot0 + tid;
            float x = 0.f, y = 0.f, z = 0.f;
            if (s < count) { x = io.bpts[3 * s]; y = io.bpts[3 * s + 1]; z = io.bpts[3 * s + 2]; }
            sm.pts[4 * tid] = x; sm.pts[4 * tid + 1] = y; sm.pts[4 * tid + 2] = z; sm.pts[4 * tid + 3] = 0.f;
        }
        __syncthreads();
        auto write_pe10 = [&]() {
            const float* p = sm.pts + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
            if (pg == 0) pe_write_col<E, 10, false>(sm.xs + pm * XS, x, nullptr, ph, 64);
            else { const float jc[3] = {pg == 1 ? TS : 0.f, pg == 2 ? TS : 0.f, pg == 3 ? TS : 0.f};
                   pe_write_col<E, 10, false>(sm.xs + (pg * 32 + pm) * XS, x, jc, ph, 64); }
        };
        auto write_pe8 = [&]() {
            const float* p = sm.cpts + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
            if (pg == 0) pe_write_col<E, 8, true>(sm.xs + pm * XS, x, nullptr, ph, 64);
            else { const float* J = sm.jac + 12 * pm; const int j = pg - 1;
                   const float jc[3] = {J[0 + j] * TS, J[3 + j] * TS, J[6 + j] * TS};
                   pe_write_col<E, 8, true>(sm.xs + (pg * 32 + pm) * XS, x, jc, ph, 64); }
        };
        // ---- residual deformation net with tangents
        write_pe10();
        __syncthreads();
        init_acc<4, true>(acc, fr.bias_r0, wave, lane);
        gemm_wide<E, 4, 4>(acc, gemm_prefetch<E, 4>(wa + net.r[0].w, wave, lane), wa + net.r[0].w, sm.xs, 0, wave, lane);
        __syncthreads();
        epilogue_wide<E, ACT_RELU, true, 4>(acc, sm.xs, wave, lane);
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l < 8; ++l) {
            init_acc<4, true>(acc, l == 4 ? fr.bias_r4 : ba + net.r[l].bias, wave, lane);
            gemm_wide<E, 16, 4>(acc, gemm_prefetch<E, 16>(wa + net.r[l].w, wave, lane), wa + net.r[l].w, sm.xs, 0, wave, lane);
            __syncthreads();
            if (l == 4) {
                write_pe10();
                __syncthreads();
                gemm_wide<E, 4, 4>(acc, gemm_prefetch<E, 4>(wa + net.r4b.w, wave, lane), wa + net.r4b.w, sm.xs, 0, wave, lane);
                __syncthreads();
            }
            epilogue_wide<E, ACT_RELU, true, 4>(acc, sm.xs, wave, lane);
            __syncthreads();
        }
        {   // head: wave g holds z (g = 0) or TS * dz/db_{g-1}
            const f32x16 h = gemm_head<E, 16>(wa + net.rhead.w, sm.xs, 0, wave, lane);
            if (lane < 32) { sm.misc[4 * (wave * 32 + lane) + 0] = h[0]; sm.misc[4 * (wave * 32 + lane) + 1] = h[1];
                             sm.misc[4 * (wave * 32 + lane) + 2] = h[2]; }
        }
        __syncthreads();
        if (tid < 32) {
            const float* b = ba + net.rhead.bias;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float th = tanhf(sm.misc[4 * tid + c] + b[c]);
                const float resd = th * io.resd_limit;
                sm.cpts[4 * tid + c] = sm.pts[4 * tid + c] + resd;
                sm.pts[4 * tid + 3] = 0.f;
                const float dth = (1.f - th * th) * io.resd_limit * (1.f / TS);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    sm.jac[12 * tid + 3 * c + j] = (c == j ? 1.f : 0.f) + dth * sm.misc[4 * ((j + 1) * 32 + tid) + c];
                sm.jac[12 * tid + 9 + c] = resd;
            }
        }
        __syncthreads();
        // ---- signed distance net with tangents
        write_pe8();
        __syncthreads();
        init_acc<4, true>(acc, ba + net.s[0].bias, wave, lane);
        gemm_wide<E, 4, 4>(acc, gemm_prefetch<E, 4>(wa + net.s[0].w, wave, lane), wa + net.s[0].w, sm.xs, 0, wave, lane);
        __syncthreads();
        epilogue_wide<E, ACT_SOFTPLUS, true, 4>(acc, sm.xs, wave, lane);
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l < 8; ++l) {
            init_acc<4, true>(acc, ba + net.s[l].bias, wave, lane);
            gemm_wide<E, 16, 4>(acc, gemm_prefetch<E, 16>(wa + net.s[l].w, wave, lane), wa + net.s[l].w, sm.xs, 0, wave, lane);
            __syncthreads();
            if (l == 4) {
                write_pe8();
                __syncthreads();
                gemm_wide<E, 4, 4>(acc, gemm_prefetch<E, 4>(wa + net.s4b.w, wave, lane), wa + net.s4b.w, sm.xs, 0, wave, lane);
                __syncthreads();
            }
            epilogue_wide<E, ACT_SOFTPLUS, true, 4>(acc, sm.xs, wave, lane);
            __syncthreads();
        }
        {   // sdf head on all four column groups: value and the three (scaled) partials
            const f32x16 h = gemm_head<E, 16>(wa + net.shead.w, sm.xs, 0, wave, lane);
            if (lane < 32) sm.misc[4 * (wave * 32 + lane) + 3] = h[0];
        }
        // features (primal columns only) -> xs rows 0..31 as the next net's input
        f32x16 facc[2][1];
        init_acc<1, false>(facc, ba + net.sfeat.bias, wave, lane);
        gemm_wide<E, 16, 1>(facc, gemm_prefetch<E, 16>(wa + net.sfeat.w, wave, lane), wa + net.sfeat.w, sm.xs, 0, wave, lane);
        __syncthreads();
        epilogue_wide<E, ACT_NONE, false, 1>(facc, sm.xs, wave, lane);
        if (io.dbg_feat) {
            const float* b = ba + net.sfeat.bias;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 64 * wave + 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int s = slot0 + (lane & 31);
                    if (s < count) io.dbg_feat[(size_t)s * 256 + n] = facc[nt][0][r];   // bias already inside facc
                }
        }
        __syncthreads();
        // ---- per-point geometry outputs (threads 0..31)
        float nrm[3] = {0.f, 0.f, 0.f}, bv[3] = {0.f, 0.f, 0.f};
        float sdfv = 0.f, occ = 0.f;
        if (tid < 32) {
            sdfv = sm.misc[4 * tid + 3] + ba[net.shead.bias];
            float g[3] = {sm.misc[4 * (32 + tid) + 3] * (1.f / TS), sm.misc[4 * (64 + tid) + 3] * (1.f / TS),
                          sm.misc[4 * (96 + tid) + 3] * (1.f / TS)};
            const int s = slot0 + tid;
            if (io.dbg_grad && s < count) { io.dbg_grad[3 * s] = g[0]; io.dbg_grad[3 * s + 1] = g[1]; io.dbg_grad[3 * s + 2] = g[2]; }
            if (io.dbg_sdf && s < count) io.dbg_sdf[s] = sdfv;
            occ = sdf_to_occ_dev(sdfv, io.beta);
            normalize3(g);
            float A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, B[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (io.mats && s < count) {
                const float* M = io.mats + (size_t)s * 24;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) { A[3 * i + j] = M[4 * i + j]; B[3 * i + j] = M[12 + 4 * i + j]; }
            }
            float Ai[9], Bi[9];
            inv3(A, Ai);
            inv3(B, Bi);
            // normal: big-pose -> T (big_R^T), T -> pose (R_inv^T), pose -> world (R), base_network.py:471-475
            float nt_[3], np_[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) nt_[i] = B[0 + i] * g[0] + B[3 + i] * g[1] + B[6 + i] * g[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) np_[i] = Ai[0 + i] * nt_[0] + Ai[3 + i] * nt_[1] + Ai[6 + i] * nt_[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) nrm[i] = np_[0] * fr.R[3 * i] + np_[1] * fr.R[3 * i + 1] + np_[2] * fr.R[3 * i + 2];
            normalize3(nrm);
            if (!io.relight && io.view && s < count) {   // view dirs to big-pose space, base_network.py:324-334 (not re-normalised)
                const int p = io.idx[s];
                const float v[3] = {io.view[3 * p], io.view[3 * p + 1], io.view[3 * p + 2]};
                float pv[3], tv[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) pv[i] = v[0] * fr.R[i] + v[1] * fr.R[3 + i] + v[2] * fr.R[6 + i];
#pragma unroll
                for (int i = 0; i < 3; ++i) tv[i] = A[0 + i] * pv[0] + A[3 + i] * pv[1] + A[6 + i] * pv[2];
#pragma unroll
                for (int i = 0; i < 3; ++i) bv[i] = Bi[0 + i] * tv[0] + Bi[3 + i] * tv[1] + Bi[6 + i] * tv[2];
            }
        }
        // ---- heads on the 32 primal columns
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
        if (io.relight) {
            init_acc<1, false>(facc, ba + mat.m0.bias, wave, lane);
            gemm_wide<E, 16, 1>(facc, gemm_prefetch<E, 16>(wa + mat.m0.w, wave, lane), wa + mat.m0.w, sm.xs, 0, wave, lane);
            __syncthreads();
            epilogue_wide<E, ACT_SOFTPLUS, false, 1>(facc, sm.xs, wave, lane);
            __syncthreads();
            init_acc<1, false>(facc, ba + mat.m1.bias, wave, lane);
            gemm_wide<E, 16, 1>(facc, gemm_prefetch<E, 16>(wa + mat.m1.w, wave, lane), wa + mat.m1.w, sm.xs, 0, wave, lane);
            __syncthreads();
            epilogue_wide<E, ACT_SOFTPLUS, false, 1>(facc, sm.xs, wave, lane);
            __syncthreads();
            if (wave == 0) {
                const f32x16 h = gemm_head<E, 16>(wa + mat.mhead.w, sm.xs, 0, 0, lane);
                if (lane < 32) {
                    const float* b = ba + mat.mhead.bias;
#pragma unroll
                    for (int c = 0; c < 3; ++c) o4[c] = io.albedo_slope / (1.f + expf(-(h[c] + b[c]))) + io.albedo_bias;
                    o4[3] = io.rough_slope / (1.f + expf(-(h[3] + b[3]))) + io.rough_bias;
                }
            }
        } else {
            init_acc<1, false>(facc, ba + col.c0a.bias, wave, lane);
            gemm_wide<E, 16, 1>(facc, gemm_prefetch<E, 16>(wa + col.c0a.w, wave, lane), wa + col.c0a.w, sm.xs, 0, wave, lane);
            __syncthreads();
            if (tid < 32) {     // [PE4(bvds) 27 | world normal 3 | pad 2] -> cols 0..31
                E* row = sm.xs + tid * XS;
                pe_write_col<E, 4, false>(row, bv, nullptr, 0, 27);
                pe_write_col<E, 4, false>(row, bv, nullptr, 1, 27);
                row[27] = (E)nrm[0]; row[28] = (E)nrm[1]; row[29] = (E)nrm[2];
                row[30] = (E)0.f; row[31] = (E)0.f;
            }
            __syncthreads();
            gemm_wide<E, 2, 1>(facc, gemm_prefetch<E, 2>(wa + col.c0b.w, wave, lane), wa + col.c0b.w, sm.xs, 0, wave, lane);
            __syncthreads();
            epilogue_wide<E, ACT_RELU, false, 1>(facc, sm.xs, wave, lane);
            __syncthreads();
#pragma unroll 1
            for (int l = 0; l < 3; ++l) {
                const WideLayer cl = l == 0 ? col.c1 : (l == 1 ? col.c2 : col.c3);
                init_acc<1, false>(facc, l == 2 ? fr.bias_c3 : ba + cl.bias, wave, lane);
                gemm_wide<E, 16, 1>(facc, gemm_prefetch<E, 16>(wa + cl.w, wave, lane), wa + cl.w, sm.xs, 0, wave, lane);
                __syncthreads();
                epilogue_wide<E, ACT_RELU, false, 1>(facc, sm.xs, wave, lane);
                __syncthreads();
            }
            if (wave == 0) {
                const f32x16 h = gemm_head<E, 16>(wa + col.chead.w, sm.xs, 0, 0, lane);
                if (lane < 32) {
                    const float* b = ba + col.chead.bias;
#pragma unroll
                    for (int c = 0; c < 3; ++c) o4[c] = 1.f / (1.f + expf(-(h[c] + b[c])));
                }
            }
        }
        // ---- scatter raw (threads 0..31 = wave 0 lanes 0..31 hold everything)
        if (tid < 32) {
            const int s = slot0 + tid;
            if (s < count) {
                float* o = io.raw + (size_t)io.idx[s] * io.C;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    o[c] = sm.cpts[4 * tid + c];
                    o[3 + c] = sm.pts[4 * tid + c];
                    o[6 + c] = sm.jac[12 * tid + 9 + c];
                }
                if (io.relight) {
                    o[9] = o4[0]; o[10] = o4[1]; o[11] = o4[2]; o[12] = o4[3];
                    o[13] = nrm[0]; o[14] = nrm[1]; o[15] = nrm[2]; o[16] = occ;
                } else {
                    o[9] = nrm[0]; o[10] = nrm[1]; o[11] = nrm[2];
                    o[12] = o4[0]; o[13] = o4[1]; o[14] = o4[2]; o[15] = occ;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

void launch_mlp_sdf(const GeoNet& net, const void* warena, const float* barena, const FrameState& fr, const MlpIO& io,
                    int max_slots, bool f16w, hipStream_t stream) {
    if (max_slots <= 0) return;
    const int tiles = (max_slots + TM - 1) / TM;
    const int grid = tiles < 512 ? tiles : 512;     // 2 workgroups per CU x 256 CUs, persistent over tiles
    const bool dbg = io.dbg_resd || io.dbg_sdf || io.dbg_feat;
    if (f16w) {
        if (dbg) hipLaunchKernelGGL((mlp_sdf_kernel<f16, true>), dim3(grid), dim3(MLP_THREADS), 0, stream, net, warena, barena, fr, io);
        else hipLaunchKernelGGL((mlp_sdf_kernel<f16, false>), dim3(grid), dim3(MLP_THREADS), 0, stream, net, warena, barena, fr, io);
    } else {
        if (dbg) hipLaunchKernelGGL((mlp_sdf_kernel<bf16, true>), dim3(grid), dim3(MLP_THREADS), 0, stream, net, warena, barena, fr, io);
        else hipLaunchKernelGGL((mlp_sdf_kernel<bf16, false>), dim3(grid), dim3(MLP_THREADS), 0, stream, net, warena, barena, fr, io);
    }
}

void launch_mlp_full(const GeoNet& net, const MatNet& mat, const ColNet& col, const void* warena, const float* barena,
                     const FrameState& fr, const FullIO& io, int max_slots, bool f16w, hipStream_t stream) {
    if (max_slots <= 0) return;
    const int tiles = (max_slots + 31) / 32;
    const int grid = tiles < 512 ? tiles : 512;
    if (f16w) hipLaunchKernelGGL((mlp_full_kernel<f16>), dim3(grid), dim3(MLP_THREADS), 0, stream, net, mat, col, warena, barena, fr, io);
    else hipLaunchKernelGGL((mlp_full_kernel<bf16>), dim3(grid), dim3(MLP_THREADS), 0, stream, net, mat, col, warena, barena, fr, io);
}
