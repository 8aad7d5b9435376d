// K3, second generation: software-pipelined HDQ fine query (resd + sdf MLPs) for gfx950.
//
// Same arithmetic as mlp_sdf_kernel (ra_mlp.hip) — bit-identical results — but organised so that
// the VALU epilogue hides inside the MFMA stream of the SAME wave instead of relying on a second
// workgroup:
//   * one workgroup of 8 waves per CU, tile = 128 points; wave w owns output rows 32w..32w+31;
//   * activations double-buffered in LDS (2 x [128][264] E = 135 KB): a layer reads X[cur] and
//     writes X[cur^1], so there is no write-after-read hazard and one barrier per step suffices;
//   * a layer is processed in two column halves (64 points each). While the MFMAs of one half run,
//     the wave executes, k-step by k-step, the bias/activation/pack/ds_write epilogue of the half
//     it finished in the previous step (32 MFMAs interleaved with 32 epilogue elements per lane);
//   * the wave keeps ALL weight fragments of its 32 rows for the current layer in registers
//     (16 k-steps x 4 VGPRs = 64 VGPRs), loaded once per layer and reloaded in place, k-step by
//     k-step, with the next layer's fragments right after their last use (one full step = 1024+
//     MFMA cycles of cover for the L2 latency) — L2 traffic per point equals the first generation.
//   reference: lib/networks/deform/base_network.py:34-42,78-87,374-382; lib/utils/net_utils.py:1263-1273,1337-1352
#include "ra_common.hpp"

namespace {

constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_SOFTPLUS = 2;
constexpr float INV_2PI = 0.15915494309189535f;
constexpr int PIPE_THREADS = 512;

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

template <typename E> struct Tr;
template <> struct Tr<bf16> {
    typedef bf16x8 x8; typedef bf16x4 x4;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Tr<f16> {
    typedef f16x8 x8; typedef f16x4 x4;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

#ifndef RA_ABL
#define RA_ABL 0
#endif

template <int ACT_>
__device__ __forceinline__ float act(float z) {
    constexpr int ACT = (RA_ABL == 1 && ACT_ == ACT_SOFTPLUS) ? ACT_RELU : (RA_ABL == 2 ? ACT_NONE : ACT_);
    if (ACT == ACT_RELU) return fmaxf(z, 0.f);
    if (ACT == ACT_SOFTPLUS) {     // identical expression to ra_mlp.hip act_fn
        const float t = z * 144.26950408889634f;
        const float e = __builtin_amdgcn_exp2f(-fabsf(t));
        return fmaf(__builtin_amdgcn_logf(1.f + e), 0.0069314718055994531f, fmaxf(z, 0.f));
    }
    return z;
}

template <typename E>
__device__ __forceinline__ void store4(E* dst, float a, float b, float c, float d) {
    typename Tr<E>::x4 v;
    v[0] = (E)a; v[1] = (E)b; v[2] = (E)c; v[3] = (E)d;
    *reinterpret_cast<typename Tr<E>::x4*>(dst) = v;
}

// positional encoding of one column, frequencies dealt to Q=4 threads (same values/channel order as
// pe_write_col in ra_mlp.hip; LO adds the 9 residual columns of the identity and frequency-0 channels)
template <typename E, int L, bool LO>
__device__ __forceinline__ void pe_write4(E* row, const float x[3], int q) {
    constexpr int NB = 3 + 6 * L;
    float rev[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rev[c] = x[c] * INV_2PI;
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const E hi = (E)x[c];
            row[c] = hi;
            if (LO) row[NB + c] = (E)(x[c] - (float)hi);
        }
    }
#pragma unroll
    for (int j = 0; j < (L + 3) / 4; ++j) {
        const int f = q + j * 4;
        if (f < L) {
            const float sc = (float)(1 << f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = rev[c] * sc;
                const float sv = __builtin_amdgcn_sinf(a), cv = __builtin_amdgcn_cosf(a);
                const E hs = (E)sv, hc = (E)cv;
                row[3 + 6 * f + c] = hs;
                row[3 + 6 * f + 3 + c] = hc;
                if (LO && f == 0) {
                    row[NB + 3 + c] = (E)(sv - (float)hs);
                    row[NB + 6 + c] = (E)(cv - (float)hc);
                }
            }
        }
    }
    if (q == 3)
        for (int c = NB + (LO ? 9 : 0); c < 64; ++c) row[c] = (E)0.f;
}

template <typename E>
struct PSmem {
    E x[2][TM * XS];        // double-buffered activation tile
    float pts[TM * 4];
    float cpts[TM * 4];
    int count;
};

template <typename E> using X8 = typename Tr<E>::x8;

// accumulators of one column half start at the bias of their row
__device__ __forceinline__ void init_half(f32x16 (&acc)[2], const float* __restrict__ bias_w, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_w + 8 * q + 4 * (lane >> 5));
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[0][4 * q + j] = bv[j]; acc[1][4 * q + j] = bv[j]; }
    }
}

// 4 consecutive rows (8q + 4*(lane>>5) + 0..3 of the wave's 32) of one column -> activation -> E -> LDS
template <typename E, int ACT>
__device__ __forceinline__ void epi_group(const f32x16& acc, int q, E* op) {
    store4<E>(op + 8 * q, act<ACT>(acc[4 * q + 0]), act<ACT>(acc[4 * q + 1]), act<ACT>(acc[4 * q + 2]), act<ACT>(acc[4 * q + 3]));
}

template <typename E, int ACT>
__device__ __forceinline__ void epi_half(const f32x16 (&acc)[2], E* xout_half, int wave, int lane) {
    E* op = xout_half + (lane & 31) * XS + 32 * wave + 4 * (lane >> 5);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) epi_group<E, ACT>(acc[mt], q, op + mt * 32 * XS);
}

// all fragments of the wave's 32 rows of a K=256 layer -> registers
template <typename E>
__device__ __forceinline__ void load_A(X8<E> (&A)[16], const X8<E>* __restrict__ wl, int wave, int lane) {
    const X8<E>* ap = wl + (size_t)(wave * 16) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) A[ks] = ap[ks * 64];
}

// One pipeline step: MFMAs of column half `xin_half` (K = 256, weights in A) into accP, interleaved with
// the epilogue of accE (the half finished in the previous step) into `xout_half`, and — after its last
// use — the in-place reload of A[ks] with the next layer's fragment.
template <typename E, int ACT, bool EPI, bool RELOAD>
__device__ __forceinline__ void step(f32x16 (&accP)[2], const f32x16 (&accE)[2], X8<E> (&A)[16], const X8<E>* __restrict__ wnext,
                                     const E* xin_half, E* xout_half, const float* __restrict__ bias_w, int wave, int lane) {
    typedef X8<E> x8;
    init_half(accP, bias_w, lane);
    const E* bp = xin_half + (lane & 31) * XS + (lane >> 5) * 8;
    E* op = xout_half + (lane & 31) * XS + 32 * wave + 4 * (lane >> 5);
    const x8* np = wnext + (size_t)(wave * 16) * 64 + lane;
    x8 b0 = *reinterpret_cast<const x8*>(bp), b1 = *reinterpret_cast<const x8*>(bp + 32 * XS);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        x8 nb0 = b0, nb1 = b1;
        if (ks + 1 < 16) {
            nb0 = *reinterpret_cast<const x8*>(bp + (ks + 1) * 16);
            nb1 = *reinterpret_cast<const x8*>(bp + 32 * XS + (ks + 1) * 16);
        }
        if (RA_ABL != 5) {
        accP[0] = Tr<E>::mfma(A[ks], b0, accP[0]);
        accP[1] = Tr<E>::mfma(A[ks], b1, accP[1]);
        } else { accP[0][ks] += (float)b0[0] * (float)A[ks][0]; accP[1][ks] += (float)b1[0]; }
        if (RELOAD && RA_ABL != 4) A[ks] = np[ks * 64];
        if (EPI && RA_ABL != 3 && (ks & 1) == 0) {
            const int g = ks >> 1;
            epi_group<E, ACT>(accE[g >> 2], g & 3, op + (g >> 2) * 32 * XS);
        }
        b0 = nb0; b1 = nb1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// non-pipelined wide layer over all 128 columns: acc4 = {accA[0], accA[1], accB[0], accB[1]} (bias pre-loaded by caller)
template <typename E, int KS>
__device__ __forceinline__ void gemm_all(f32x16 (&accA)[2], f32x16 (&accB)[2], const X8<E>* __restrict__ wl, const E* xin, int wave, int lane) {
    typedef X8<E> x8;
    const x8* ap = wl + (size_t)(wave * KS) * 64 + lane;
    const E* bp = xin + (lane & 31) * XS + (lane >> 5) * 8;
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
        const x8 a = ap[ks * 64];
        const x8 b0 = *reinterpret_cast<const x8*>(bp + ks * 16);
        const x8 b1 = *reinterpret_cast<const x8*>(bp + 32 * XS + ks * 16);
        const x8 b2 = *reinterpret_cast<const x8*>(bp + 64 * XS + ks * 16);
        const x8 b3 = *reinterpret_cast<const x8*>(bp + 96 * XS + ks * 16);
        accA[0] = Tr<E>::mfma(a, b0, accA[0]);
        accA[1] = Tr<E>::mfma(a, b1, accA[1]);
        accB[0] = Tr<E>::mfma(a, b2, accB[0]);
        accB[1] = Tr<E>::mfma(a, b3, accB[1]);
    }
}

// <= 32 output rows, K = 256; wave g (< 4) handles column group g. Lanes 0..31: rows 0..3 in acc[0..3]
template <typename E>
__device__ __forceinline__ f32x16 gemm_head(const X8<E>* __restrict__ wl, const E* xin, int colgrp, int lane) {
    typedef X8<E> x8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const x8* ap = wl + lane;
    const E* bp = xin + (colgrp * 32 + (lane & 31)) * XS + (lane >> 5) * 8;
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
        const x8 a = ap[ks * 64];
        const x8 b = *reinterpret_cast<const x8*>(bp + ks * 16);
        acc = Tr<E>::mfma(a, b, acc);
    }
    return acc;
}

// three consecutive regular layers (256 -> 256) in the two-half pipeline. Xin holds the complete input on entry
// (synchronised); on exit Xin holds the complete output of the third layer (synchronised).
template <typename E, int ACT>
__device__ __forceinline__ void pipe_group3(const WideLayer& L0, const WideLayer& L1, const WideLayer& L2, const X8<E>* __restrict__ wa,
                                            const float* __restrict__ ba, E*& Xin, E*& Xout, f32x16 (&accA)[2], f32x16 (&accB)[2],
                                            X8<E> (&A)[16], int wave, int lane) {
    const float* b0 = ba + L0.bias + 32 * wave;
    const float* b1 = ba + L1.bias + 32 * wave;
    const float* b2 = ba + L2.bias + 32 * wave;
    load_A<E>(A, wa + L0.w, wave, lane);
    // layer 0
    step<E, ACT, false, false>(accA, accB, A, wa + L1.w, Xin, Xout, b0, wave, lane);
    step<E, ACT, true, true>(accB, accA, A, wa + L1.w, Xin + 64 * XS, Xout, b0, wave, lane);       // epi (0,C0) -> Xout[C0]; A <- layer 1
    __syncthreads();
    { E* t = Xin; Xin = Xout; Xout = t; }
    // layer 1
    step<E, ACT, true, false>(accA, accB, A, wa + L2.w, Xin, Xin + 64 * XS, b1, wave, lane);      // epi (0,C1) -> Xin[C1]
    __syncthreads();
    step<E, ACT, true, true>(accB, accA, A, wa + L2.w, Xin + 64 * XS, Xout, b1, wave, lane);       // epi (1,C0) -> Xout[C0]; A <- layer 2
    __syncthreads();
    { E* t = Xin; Xin = Xout; Xout = t; }
    // layer 2
    step<E, ACT, true, false>(accA, accB, A, wa + L2.w, Xin, Xin + 64 * XS, b2, wave, lane);      // epi (1,C1) -> Xin[C1]
    __syncthreads();
    step<E, ACT, true, false>(accB, accA, A, wa + L2.w, Xin + 64 * XS, Xout, b2, wave, lane);      // epi (2,C0) -> Xout[C0]
    epi_half<E, ACT>(accB, Xout + 64 * XS, wave, lane);                                            // drain: epi (2,C1) -> Xout[C1]
    __syncthreads();
    { E* t = Xin; Xin = Xout; Xout = t; }
}

// one network: PE -> L0 (K=64) -> L1..L3 (pipelined) -> L4 (K=256 + K=64 skip) -> L5..L7 (pipelined); output in Xin
template <typename E, int ACT, int PEL, bool LO>
__device__ __forceinline__ void run_net(const WideLayer* Lr /* 8 layers */, const WideLayer& Lskip, const float* bias0, const float* bias4,
                                        const X8<E>* __restrict__ wa, const float* __restrict__ ba, const float* pin, E*& Xin, E*& Xout,
                                        f32x16 (&accA)[2], f32x16 (&accB)[2], X8<E> (&A)[16], int tid, int wave, int lane) {
    const int pm = tid >> 2, pq = tid & 3;
    { const float* p = pin + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
      pe_write4<E, PEL, LO>(Xin + pm * XS, x, pq); }
    __syncthreads();
    // L0: K = 64
    init_half(accA, bias0 + 32 * wave, lane);
    init_half(accB, bias0 + 32 * wave, lane);
    gemm_all<E, 4>(accA, accB, wa + Lr[0].w, Xin, wave, lane);
    epi_half<E, ACT>(accA, Xout, wave, lane);
    epi_half<E, ACT>(accB, Xout + 64 * XS, wave, lane);
    __syncthreads();
    { E* t = Xin; Xin = Xout; Xout = t; }
    pipe_group3<E, ACT>(Lr[1], Lr[2], Lr[3], wa, ba, Xin, Xout, accA, accB, A, wave, lane);
    // L4: skip layer = K=256 over Xin + K=64 over the re-encoded input (staged in Xout cols 0..63)
    { const float* p = pin + 4 * pm; const float x[3] = {p[0], p[1], p[2]};
      pe_write4<E, PEL, LO>(Xout + pm * XS, x, pq); }
    init_half(accA, bias4 + 32 * wave, lane);
    init_half(accB, bias4 + 32 * wave, lane);
    gemm_all<E, 16>(accA, accB, wa + Lr[4].w, Xin, wave, lane);
    __syncthreads();                                   // PE visible
    gemm_all<E, 4>(accA, accB, wa + Lskip.w, Xout, wave, lane);
    __syncthreads();                                   // all reads of Xout[:, 0..63] done
    epi_half<E, ACT>(accA, Xout, wave, lane);
    epi_half<E, ACT>(accB, Xout + 64 * XS, wave, lane);
    __syncthreads();
    { E* t = Xin; Xin = Xout; Xout = t; }
    pipe_group3<E, ACT>(Lr[5], Lr[6], Lr[7], wa, ba, Xin, Xout, accA, accB, A, wave, lane);
}

template <typename E>
__global__ __launch_bounds__(PIPE_THREADS, 2) void mlp_sdf_pipe_kernel(GeoNet net, const void* __restrict__ wa_, const float* __restrict__ ba,
                                                                      FrameState fr, MlpIO io) {
    typedef X8<E> x8;
    __shared__ __attribute__((aligned(16))) PSmem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sm.count = *io.count;
    __syncthreads();
    const int count = sm.count;
    if (blockIdx.x == 0 && tid == 0 && io.counters) atomicAdd(&io.counters->n_fine_sdf, (unsigned long long)count);
    f32x16 accA[2], accB[2];
    x8 A[16];
    const x8* __restrict__ wa_base = reinterpret_cast<const x8*>(wa_);

    for (int tile = blockIdx.x; tile * TM < count; tile += gridDim.x) {
        const int slot0 = tile * TM;
        unsigned wz = 0;
        asm volatile("" : "+s"(wz));          // opaque zero: stops LICM from hoisting weight loads out of the tile loop
        const x8* wa = wa_base + wz;
        if (tid < TM) {
            const int s = slot0 + tid;
            float x = 0.f, y = 0.f, z = 0.f;
            if (s < count) { x = io.bpts[3 * s]; y = io.bpts[3 * s + 1]; z = io.bpts[3 * s + 2]; }
            sm.pts[4 * tid] = x; sm.pts[4 * tid + 1] = y; sm.pts[4 * tid + 2] = z; sm.pts[4 * tid + 3] = 0.f;
        }
        __syncthreads();
        E* Xin = sm.x[0];
        E* Xout = sm.x[1];
        // ---- residual deformation net (ReLU)
        run_net<E, ACT_RELU, 10, false>(net.r, net.r4b, fr.bias_r0, fr.bias_r4, wa, ba, sm.pts, Xin, Xout, accA, accB, A, tid, wave, lane);
        if (wave < 4) {     // head: resd = tanh(z) * resd_limit; cpts = bpts + resd
            const f32x16 h = gemm_head<E>(wa + net.rhead.w, Xin, wave, lane);
            if (lane < 32) {
                const int m = wave * 32 + lane;
                const float* b = ba + net.rhead.bias;
#pragma unroll
                for (int c = 0; c < 3; ++c) sm.cpts[4 * m + c] = sm.pts[4 * m + c] + tanhf(h[c] + b[c]) * io.resd_limit;
            }
        }
        __syncthreads();
        // ---- signed distance net (softplus)
        run_net<E, ACT_SOFTPLUS, 8, true>(net.s, net.s4b, ba + net.s[0].bias, ba + net.s[4].bias, wa, ba, sm.cpts, Xin, Xout, accA, accB, A, tid, wave, lane);
        if (wave < 4) {     // head: sdf, then the HDQ blend (base_network.py:374-382)
            const f32x16 h = gemm_head<E>(wa + net.shead.w, Xin, wave, lane);
            if (lane < 32) {
                const int s = slot0 + wave * 32 + lane;
                if (s < count) {
                    float d = h[0] + ba[net.shead.bias];
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
        __syncthreads();
    }
}

}  // namespace

void launch_mlp_sdf_pipe(const GeoNet& net, const void* warena, const float* barena, const FrameState& fr, const MlpIO& io,
                         int max_slots, bool f16w, hipStream_t stream) {
    if (max_slots <= 0) return;
    const int tiles = (max_slots + TM - 1) / TM;
    const int grid = tiles < 256 ? tiles : 256;     // one 8-wave workgroup per CU, persistent over tiles
    if (f16w) hipLaunchKernelGGL((mlp_sdf_pipe_kernel<f16>), dim3(grid), dim3(PIPE_THREADS), 0, stream, net, warena, barena, fr, io);
    else hipLaunchKernelGGL((mlp_sdf_pipe_kernel<bf16>), dim3(grid), dim3(PIPE_THREADS), 0, stream, net, warena, barena, fr, io);
}
