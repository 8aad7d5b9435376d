// K3C: the HDQ fine distance query (resd + sdf MLPs) in COMPENSATED arithmetic — the precise tier of the surface trace.
//
// Why it exists (tools/precision_tiers.py): with plain f16 MFMA operands the distance is 5e-5 rms from the
// reference's fp32 value.  The reference's 16-iteration sphere trace ends in a limit cycle on part of the rays, and a perturbation
// of that size flips the cycle's phase on ~1 % of the pixels (4 mm surface jumps, up to 0.1 rgb) — SURVEY.md:409's max |err| <= 1e-2
// cannot be met.  fp32 itself IS stable there (a differently associated fp32 oracle agrees to 89-110 dB), and compensating ONLY the
// surface trace's distance queries (2 % of a relit frame's fine queries) brings whole frames to 67 dB / max 6e-3 while the 15 M
// shadow queries stay on plain f16 (K3).  Arithmetic: ra_stream.hpp row_block_16 — operands as f16 hi + lo pairs, three MFMAs per
// k-step, fp32 accumulate: 1.3e-7 rms from a float64 evaluation, the same as fp32 arithmetic itself (1.2e-7).
//
//   * tile shape v_mfma_f32_16x16x32_f16: 16 points per wave, 16 row blocks of 16 rows per layer, k-steps of 32 (ra_stream.hpp row_block_16);
//   * weights: the split stream (ra_pack.cpp add16): every A fragment twice, [hi | lo]: 3872 fragments (4 MB) per tile;
//   * activations: fp32 in the accumulators, bias / activation in fp32, then hi = f16(a), lo = f16(a - hi) — two B fragment sets;
//   * encoding: the argument of every sin / cos is reduced in two-constant arithmetic (x 2^k is exact; k = rint(a c_hi),
//     t = fma(a, c_hi, -k) + a c_lo) before v_sin / v_cos — the plain kernel's a * (1 / 2 pi) loses 1.5e-5 rad at 2^9 x, invisible
//     behind an f16 rounding, not here;
//   * 8 waves per workgroup (two per SIMD, 128 points per tile) on launches above 16 Ki points, 4 (one per SIMD, 64 points) down to 8 Ki;
//     below that K3CC (ra_k3cc.hpp): four waves per 16-point tile.  The smaller the launch, the more SIMDs share it.
//   reference: lib/networks/deform/base_network.py:34-42,78-87,374-382; lib/utils/net_utils.py:1263-1273,1337-1352;
//   lib/networks/embedder.py:26-37; hit test and sign-change interpolation that consume the result: sphere_tracing_renderer.py:176-197
#include "ra_stream.hpp"

namespace {

constexpr int STC_FRAGS = 3872;                  // per net: 64 + 3 x 256 + 320 + 3 x 256 + 16 (a 16-row head), x 2 nets
constexpr int STC_STAGES = STC_FRAGS / 16;       // 242
constexpr float INV_2PI_HI = 0.15915494f;        // fp32(1 / 2 pi)
constexpr float INV_2PI_LO = 3.0934362e-09f;     // 1 / 2 pi - INV_2PI_HI (cancels the rounding of the first constant)

// sin / cos of 2^k x with an exactly scaled argument and a two-constant reduction to [-0.5, 0.5] revolutions
__device__ __forceinline__ void sincos_rev(float a, float& sv, float& cv) {
    const float k = __builtin_rintf(a * INV_2PI_HI);
    float t = __builtin_fmaf(a, INV_2PI_HI, -k);
    t = __builtin_fmaf(a, INV_2PI_LO, t);
    sv = __builtin_amdgcn_sinf(t);
    cv = __builtin_amdgcn_cosf(t);
}

// encoding B fragments (hi and lo, two k-steps of 32) of one point for lane group g: slot (g, j) of k-step p is slot q = 8 (2 p + (g >> 1)) + j
// of lane half h = g & 1 of the 32x32 layout, whose slot -> channel map is pe_chan_resd / pe_chan_sdf (ra_pack.cpp).
// SDFNET: the slots the plain kernel uses for its own hi + lo columns (q >= 24 with h = 1, q >= 27) carry zeros here.
template <int L, bool SDFNET>
__device__ __forceinline__ void pe_frags_16(u32x4 (&BpH)[2], u32x4 (&BpL)[2], const float (&x)[3], int g) {
    const int h = g & 1, up = g >> 1;
    float v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q < 3 * L) {
            float sv, cv;
            sincos_rev(x[q % 3] * (float)(1 << (q / 3)), sv, cv);
            v[q] = h ? cv : sv;
        } else if (!SDFNET) {
            v[q] = (q == 3 * L) ? (h ? x[1] : x[0]) : ((q == 3 * L + 1) ? (h ? 0.f : x[2]) : 0.f);
        } else {
            const int r = q - 3 * L;
            v[q] = (r < 3 && !h) ? x[r] : 0.f;
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int q0 = 8 * (2 * p) + 2 * w, q1 = 8 * (2 * p + 1) + 2 * w;
            const float a = up ? v[q1] : v[q0], b = up ? v[q1 + 1] : v[q0 + 1];
            f16x2 hv;
            hv[0] = (f16)a; hv[1] = (f16)b;
            BpH[p][w] = __builtin_bit_cast(unsigned, hv);
            BpL[p][w] = pack2<f16>(a - (float)hv[0], b - (float)hv[1]);
        }
}

// one network: L0 (encoding) .. L7, then the 16-row head; returns the head accumulator (bias included): rows 4 g + i of lane group g
template <bool LAST, int ACT, int PEL, bool SDFNET, typename PipeT>
__device__ __forceinline__ f32x4 run_net_16(PipeT& P, const float (&x)[3], const float* bias, int g) {
    u32x4 B0h[8], B0l[8], B1h[8], B1l[8], Bph[2], Bpl[2];
    Acc16 accA, accB;
    pe_frags_16<PEL, SDFNET>(Bph, Bpl, x, g);
    layer_16<2, ACT, ACT, false>(P, accA, accB, B0h, B0l /* unused */, Bph, Bpl, B0h, B0l, bias, g);
    layer_16<8, ACT, ACT, true>(P, accA, accB, B0h, B0l, Bph, Bpl, B1h, B1l, bias + 256, g);
    layer_16<8, ACT, ACT, true>(P, accA, accB, B1h, B1l, Bph, Bpl, B0h, B0l, bias + 512, g);
    layer_16<8, ACT, ACT, true>(P, accA, accB, B0h, B0l, Bph, Bpl, B1h, B1l, bias + 768, g);
    layer_16<10, ACT, ACT, true>(P, accA, accB, B1h, B1l, Bph, Bpl, B0h, B0l, bias + 1024, g);
    layer_16<8, ACT, ACT, true>(P, accA, accB, B0h, B0l, Bph, Bpl, B1h, B1l, bias + 1280, g);
    layer_16<8, ACT, ACT, true>(P, accA, accB, B1h, B1l, Bph, Bpl, B0h, B0l, bias + 1536, g);
    layer_16<8, ACT, ACT, true>(P, accA, accB, B0h, B0l, Bph, Bpl, B1h, B1l, bias + 1792, g);
    row_block_16<0, 8, ACT, true, true, LAST, 7, 1, PipeT>(P, accA, accB, B1h, B1l, Bph, Bpl, B1h, B1l, bias + 2048, g);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = accA.val(e);
    return out;
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void mlp_sdf_comp_kernel(GeoNet net, const void* __restrict__ stream, const float* __restrict__ ba, FrameState fr, MlpIO io) {
    typedef f16 E;
    __shared__ __attribute__((aligned(16))) StSmem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    constexpr int ST_TM = 16 * NW;     // points per workgroup tile
    // bias table as in K3: resd rows (L0 / L4 carry the per-frame pose condition), resd head, scaled sdf rows, sdf head
    for (int i = tid; i < BIAS_ROWS * 256; i += 64 * NW) {
        const int row = i >> 8, r = i & 255;
        float v = 0.f;
        if (row < 8) v = row == 0 ? fr.bias_r0[r] : (row == 4 ? fr.bias_r4[r] : ba[net.r[row].bias + r]);
        else if (row == 8) v = r < 32 ? ba[net.rhead.bias + r] : 0.f;
        else if (row < 17) v = ba[net.s[row - 9].bias + r] * SP_SCALE;
        else v = r < 32 ? ba[net.shead.bias + r] * SP_SCALE : 0.f;
        sm.bias[i] = v;
    }
    if (tid == 0) sm.count = *io.count;
    __syncthreads();
    const int count = sm.count;
    if (blockIdx.x == 0 && tid == 0 && io.counters) {
        atomicAdd(&io.counters->n_fine_sdf, (unsigned long long)count);
        atomicAdd(&io.counters->n_fine_sdf_comp, (unsigned long long)count);
    }
    const int ntiles = (count + ST_TM - 1) / ST_TM;
    if ((int)blockIdx.x >= ntiles) return;

    Pipe<E, NW, STC_STAGES, 8> P;
    P.g = reinterpret_cast<const char*>(stream);
    P.voff = wave * (16 / NW) * 1024 + lane * 16;
    P.ring = reinterpret_cast<const char*>(sm.ring) + lane * 16;
    P.ring_addr = (unsigned)(size_t)sm.ring + wave * (16 / NW) * 1024;
    P.slot = ST_RING - 1;            // the first sync_stage() advances to slot 0 / stream stage 0
    P.sstage = STC_STAGES - 1;
    P.rd = P.ring;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < ST_AHEAD; ++st) P.issue(st, st);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s = tile * ST_TM + wave * 16 + c;           // the four lane groups of a column hold the same point
        float x[3] = {0.f, 0.f, 0.f};
        int pidx = 0;
        float smpl = 0.f;
        if (s < count) {
            x[0] = io.bpts[3 * s]; x[1] = io.bpts[3 * s + 1]; x[2] = io.bpts[3 * s + 2];
            pidx = io.idx[s];
            if (io.smooth) smpl = io.sdf[pidx];
        }
        static_for<0, 8>([&](auto f_) { P.template fetch<decltype(f_)::value>(); });      // the tile's first PF fragments
        // ---- residual deformation net (ReLU); head: resd = tanh(z) * resd_limit, cpts = bpts + resd
        const f32x4 hr = run_net_16<false, ACT_RELU, 10, false>(P, x, sm.bias, g);
        float cp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float r = tanhf(hr[k]) * io.resd_limit;           // valid in lane group 0 (rows 0..2)
            cp[k] = x[k] + __shfl(r, c);
        }
        // ---- signed distance net (softplus, scaled domain); head row 0 = sdf
        const f32x4 hs = run_net_16<true, ACT_SOFTPLUS, 8, true>(P, cp, sm.bias + 9 * 256, g);
        if (g == 0 && s < count) {
            float d = hs[0] * SP_INV;                                 // head accumulates beta*log2(e) * sdf
            if (io.smooth) {                                          // HDQ blend (base_network.py:374-382)
                const float r = fminf(fmaxf(fabsf(d) / io.dist_th, 0.f), 1.f);
                d = smpl * r + d * (1.f - r);
            }
            io.sdf[pidx] = d;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <int NW>
void launch_c_nw(const GeoNet& net, const void* sarena_c, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream, int grid_slots = 0) {
    const int grid = mlp_grid(max_slots, grid_slots, 16 * NW);
    hipLaunchKernelGGL((mlp_sdf_comp_kernel<NW>), dim3(grid), dim3(64 * NW), 0, stream, net, sarena_c, barena, fr, io);
}

}  // namespace
