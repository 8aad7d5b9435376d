// K3, third generation: HDQ fine query (resd + sdf MLPs) with register-resident activations and the
// weights streamed through LDS.
//
// The first two generations keep the 128-point activation tile in LDS: every layer stores 64 KB of
// activations with 8-byte ds_writes (~80 B/clk/CU on gfx950, and the store transfer is not hidden by
// interleaved loads) and every wave re-reads the whole tile as its B operand.  Compile-time ablation showed
// that the MFMA + operand-read stream alone runs at the matrix-pipe ceiling and that the LDS activation
// round trip costs as much time again.  This kernel removes it:
//
//   * a wave owns 32 points (the N = 32 columns of v_mfma_f32_32x32x16) through ALL layers and computes all
//     256 output rows of a layer itself, in 8 row blocks of 32;
//   * the D fragment of a row block (lane = point column, 16 rows per lane) is, after activation and f16
//     packing, exactly two B fragments of the next layer — provided the next layer's weights are packed with
//     the matching K permutation (ra_pack.cpp, StreamBuilder).  Activations never leave the registers;
//   * the weights are the shared operand: all 8 waves of the workgroup consume the same sequence of 1 KB A
//     fragments (1952 per 256-point tile), which one LDS-DMA stream (global_load_lds_dwordx4, 16 KB stages,
//     8-stage ring = 128 KB, 7 stages in flight, counted vmcnt + one raw s_barrier per stage) delivers in
//     consumption order.  L2 -> CU weight traffic per point is half that of the 128-point tiles;
//   * the bias/activation/pack epilogue of row block rb-1 is interleaved, element pair by element pair, with
//     the 16 MFMAs of row block rb (also across layer boundaries: the last two B fragments of a layer are
//     only needed by the final two k-steps of the next row block);
//   * softplus layers run in the scaled domain y' = y * beta*log2(e): y' = max(z',0) + log2(1 + 2^-|z'|) costs
//     5 VALU ops per element; the scale lives in the weights fed by the unscaled encoding and in the biases.
//   reference: lib/networks/deform/base_network.py:34-42,78-87,374-382; lib/utils/net_utils.py:1263-1273,1337-1352
#include "ra_common.hpp"
#include <type_traits>

#ifndef RA_SCHED
#define RA_SCHED 0     // 1: pin the per-MFMA-slot instruction order with sched_barrier(0) (A/B: 1 % slower)
#endif
#ifndef RA_ABL
#define RA_ABL 0      // compile-time ablations for timing experiments (tools/): results are garbage when != 0
#endif

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int ACT_RELU = 1, ACT_SOFTPLUS = 2;
constexpr float INV_2PI = 0.15915494309189535f;
constexpr float SP_SCALE = 144.26950408889634f;     // beta * log2(e), beta = 100 (net_utils.py:1298)
constexpr float SP_INV = 0.0069314718055994531f;    // ln(2) / beta
constexpr int ST_MAXW = 8;                          // waves per workgroup: 8 (tile = 256 points), or 4 / 2 for small launches
constexpr int ST_RING = 8;                          // ring stages
constexpr int ST_STAGE_BYTES = 16384;               // 16 fragments of 1 KB
constexpr int ST_FRAGS = 1952;                      // fragments per tile
constexpr int ST_STAGES = ST_FRAGS / 16;            // 122
constexpr int ST_AHEAD = ST_RING - 2;               // stages in flight: stage st+6 refills the slot of stage st-2, whose reads
                                                    // were all consumed by MFMAs issued before the barrier (no lgkmcnt wait needed)
constexpr int ST_PF = 4;                            // A fragments read ahead of their MFMA
constexpr int BIAS_ROWS = 18;                       // resd 0..7, rhead, sdf 0..7, shead

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <typename E> struct Tr;
template <> struct Tr<bf16> {
    typedef bf16x8 x8; typedef bf16x2 x2;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Tr<f16> {
    typedef f16x8 x8; typedef f16x2 x2;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename E>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typename Tr<E>::x2 v;
    v[0] = (E)a; v[1] = (E)b;
    return __builtin_bit_cast(unsigned, v);
}

// max(z, 0) as ONE compiler-visible instruction (v_med3_f32 z, 0, +inf): fmaxf() adds a canonicalising self-max, and an
// inline-asm v_max hides the read of a just-written MFMA accumulator from the hazard recogniser (no wait states inserted:
// wrong values as soon as the scheduler places it right behind the producing MFMA)
__device__ __forceinline__ float max0(float z) { return __builtin_amdgcn_fmed3f(z, 0.f, 3.0e38f); }     // finite bound: with +inf LLVM folds it back to two v_max

template <int ACT>
__device__ __forceinline__ float act(float z) {
    if (ACT == ACT_RELU || RA_ABL == 1) return max0(z);
    // scaled-domain softplus: z = beta*log2(e) * pre-activation, result = beta*log2(e) * softplus
    const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(z));
    return max0(z) + __builtin_amdgcn_logf(1.f + e);
}

template <typename E> struct StSmem {
    E ring[ST_RING * ST_STAGE_BYTES / 2];
    float bias[BIAS_ROWS * 256];
    int count;
};

// two consecutive 1 KB fragments: global (uniform base + per-lane offset) -> LDS (uniform base + lane * 16)
// (no instruction offset: on LDS-DMA loads it would also move the LDS destination)
__device__ __forceinline__ void glds16x2(const char* sbase, unsigned voff, unsigned voff2, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, %4, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "v"(voff2), "s"(sbase), "s"(lds_dst) : "memory", "scc");
}

// the weight stream as seen by one wave of an NW-wave workgroup (each wave moves 16 / NW fragments of every stage)
template <typename E, int NW>
struct Pipe {
    static constexpr int FPW = 16 / NW;
    const char* g;          // weight stream (uniform)
    unsigned voff;          // per lane: wave * FPW * 1024 + lane * 16
    const char* ring;       // LDS ring (generic pointer), + lane * 16
    unsigned ring_addr;     // LDS byte address of the ring + wave * FPW * 1024 (wave-uniform)
    unsigned slot;          // ring slot of the stage being read (wave-uniform)
    int sstage;             // its position in the tile's stream, 0 .. ST_STAGES-1 (wave-uniform)
    const char* rd;         // ring + slot * 16 KB + lane * 16
    typename Tr<E>::x8 af[ST_PF];

    __device__ __forceinline__ void issue(int stream_stage, unsigned ring_slot) {
        const char* sb = g;
        asm volatile("" : "+s"(sb));            // keeps the 244 per-stage addresses from being precomputed (and spilled)
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + ring_slot * ST_STAGE_BYTES);
        const char* src = sb + (size_t)(RA_ABL == 7 ? (stream_stage & 1) : stream_stage) * ST_STAGE_BYTES;
        if (RA_ABL != 5 && RA_ABL != 6) {
#pragma unroll
            for (int j = 0; j < FPW / 2; ++j) glds16x2(src, voff + j * 2048, voff + j * 2048 + 1024, dst + j * 2048);
        }
    }
    // the next stage of the stream becomes readable; the ring slot two stages back is refilled ST_AHEAD stages ahead.
    // The stage position is run-time state (SGPRs), so the code below only depends on a fragment's position in its stage.
    __device__ __forceinline__ void sync_stage() {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FPW * (ST_AHEAD - 1)) : "memory");
        if (RA_ABL != 4 && RA_ABL != 6) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = (slot + 1) & (ST_RING - 1);
        sstage = sstage + 1 == ST_STAGES ? 0 : sstage + 1;
        int ahead = sstage + ST_AHEAD;
        ahead = ahead >= ST_STAGES ? ahead - ST_STAGES : ahead;
        issue(ahead, (slot + ST_AHEAD) & (ST_RING - 1));
        rd = ring + slot * ST_STAGE_BYTES;
    }
    // FM: position of the fragment in its 16-fragment stage
    template <int FM>
    __device__ __forceinline__ void fetch() {
        if (FM == 0) sync_stage();
        af[FM % ST_PF] = *reinterpret_cast<const typename Tr<E>::x8*>(rd + FM * 1024);
    }
};

template <typename E> using X8 = typename Tr<E>::x8;

// accumulators of a row block start at the bias of their rows: lane (c, h), acc[4q + i] <-> row 8q + 4h + i
__device__ __forceinline__ void init_acc(f32x16& acc, const float* bias_rb, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_rb + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * q + i] = bv[i];
    }
}

// One row block: KS MFMAs (fragments F0.. of the tile's stream) into `acc`, interleaved with the pending epilogue
// of `accPrev` (activation ACT_PREV) into the B fragments o0, o1.  Bm: hidden-part B fragments (KS >= 16),
// Bp: encoding B fragments (KS == 4 or the last 4 k-steps of KS == 20).
template <typename E, int NW, int FM0, int KS, int ACT_PREV, bool PENDING, bool EARLY, bool TAIL>
__device__ __forceinline__ void row_block(Pipe<E, NW>& P, f32x16& acc, const f32x16& accPrev, u32x4 (&Bm)[16], const u32x4 (&Bp)[4],
                                          u32x4& o0, u32x4& o1, const float* bias_rb, int h) {
    init_acc(acc, bias_rb, h);
    float ta[16], tb[16];
    static_for<0, KS>([&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        const u32x4 bw = (KS == 4) ? Bp[ks & 3] : (ks < 16 ? Bm[ks & 15] : Bp[ks & 3]);
        acc = Tr<E>::mfma(P.af[(FM0 + ks) % ST_PF], __builtin_bit_cast(X8<E>, bw), acc);
        if constexpr (!(TAIL && ks + ST_PF >= KS)) P.template fetch<(FM0 + ks + ST_PF) % 16>();
        if constexpr (PENDING) {
            // Pending row block, element by element.  The softplus chain exp2 -> +1 -> log2 -> +max is software-pipelined
            // over four MFMA slots so that no VALU op waits on one issued in the same slot (in-order issue: a stalled
            // transcendental chain would hold back the next MFMA).  LAST: last slot whose results may still be written.
            static_for<0, 16>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                constexpr bool SP = ACT_PREV == ACT_SOFTPLUS && RA_ABL != 1 && RA_ABL != 3 && RA_ABL != 6;
                constexpr int DEPTH = SP ? 3 : 0;
                constexpr int LAST = (KS == 4) ? 3 : (EARLY ? 13 : KS - 1);
                constexpr int s0 = (KS == 4) ? 0 : (e * (LAST - DEPTH + 1)) / 16;
                if constexpr (SP && KS != 4) {
                    if constexpr (s0 == ks) { ta[e] = __builtin_amdgcn_exp2f(-__builtin_fabsf(accPrev[e])); tb[e] = max0(accPrev[e]); }
                    if constexpr (s0 + 1 == ks) ta[e] = 1.f + ta[e];
                    if constexpr (s0 + 2 == ks) ta[e] = __builtin_amdgcn_logf(ta[e]);
                    if constexpr (s0 + 3 == ks) ta[e] = ta[e] + tb[e];
                } else if constexpr (SP) {          // 4-MFMA row blocks (first layer): one stage of all 16 elements per slot
                    if constexpr (ks == 0) { ta[e] = __builtin_amdgcn_exp2f(-__builtin_fabsf(accPrev[e])); tb[e] = max0(accPrev[e]); }
                    if constexpr (ks == 1) ta[e] = 1.f + ta[e];
                    if constexpr (ks == 2) ta[e] = __builtin_amdgcn_logf(ta[e]);
                    if constexpr (ks == 3) ta[e] = ta[e] + tb[e];
                } else {
                    if constexpr ((KS == 4 ? e / 4 : s0) == ks) ta[e] = (RA_ABL == 3 || RA_ABL == 6) ? accPrev[e] : max0(accPrev[e]);
                }
                constexpr int sdone = (KS == 4) ? (SP ? 3 : e / 4) : s0 + DEPTH;
                if constexpr ((e & 1) && sdone == ks) {
                    const unsigned w = (RA_ABL == 3 || RA_ABL == 6) ? __builtin_bit_cast(unsigned, ta[e]) : pack2<E>(ta[e - 1], ta[e]);
                    if constexpr (e < 8) o0[e >> 1] = w; else o1[(e >> 1) & 3] = w;
                }
            });
        }
        if (RA_SCHED) __builtin_amdgcn_sched_barrier(0);
    });
}

// a 256-row layer: 8 row blocks; on entry `accB` holds the pending last row block of the previous layer (if PEND_IN,
// activation ACT_IN, destination Bm[14], Bm[15]); on exit accB holds this layer's pending row block 7.
template <typename E, int NW, int KS, int ACT, int ACT_IN, bool PEND_IN>
__device__ __forceinline__ void layer(Pipe<E, NW>& P, f32x16& accA, f32x16& accB, u32x4 (&Bm)[16], const u32x4 (&Bp)[4], u32x4 (&Bo)[16],
                                      const float* bias, int h) {
    row_block<E, NW, 0, KS, ACT_IN, PEND_IN, true, false>(P, accA, accB, Bm, Bp, Bm[14], Bm[15], bias, h);
    row_block<E, NW, (1 * KS) % 16, KS, ACT, true, false, false>(P, accB, accA, Bm, Bp, Bo[0], Bo[1], bias + 32, h);
    row_block<E, NW, (2 * KS) % 16, KS, ACT, true, false, false>(P, accA, accB, Bm, Bp, Bo[2], Bo[3], bias + 64, h);
    row_block<E, NW, (3 * KS) % 16, KS, ACT, true, false, false>(P, accB, accA, Bm, Bp, Bo[4], Bo[5], bias + 96, h);
    row_block<E, NW, (4 * KS) % 16, KS, ACT, true, false, false>(P, accA, accB, Bm, Bp, Bo[6], Bo[7], bias + 128, h);
    row_block<E, NW, (5 * KS) % 16, KS, ACT, true, false, false>(P, accB, accA, Bm, Bp, Bo[8], Bo[9], bias + 160, h);
    row_block<E, NW, (6 * KS) % 16, KS, ACT, true, false, false>(P, accA, accB, Bm, Bp, Bo[10], Bo[11], bias + 192, h);
    row_block<E, NW, (7 * KS) % 16, KS, ACT, true, false, false>(P, accB, accA, Bm, Bp, Bo[12], Bo[13], bias + 224, h);
}

// encoding B fragments of one point (lane half h): see pe_chan_resd / pe_chan_sdf in ra_pack.cpp
template <typename E, int L, bool LO>
__device__ __forceinline__ void pe_frags(u32x4 (&Bp)[4], const float (&x)[3], int h) {
    float rev[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rev[c] = x[c] * INV_2PI;
    float v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q < 3 * L) {
            const float a = rev[q % 3] * (float)(1 << (q / 3));
            const float sv = __builtin_amdgcn_sinf(a), cv = __builtin_amdgcn_cosf(a);
            v[q] = h ? cv : sv;
        } else if (!LO) {
            v[q] = (q == 3 * L) ? (h ? x[1] : x[0]) : ((q == 3 * L + 1) ? (h ? 0.f : x[2]) : 0.f);
        } else {
            const int r = q - 3 * L;
            if (r < 3) {
                const float hi = (float)(E)x[r];
                v[q] = h ? x[r] - hi : hi;
            } else if (r < 6) {
                const float sv = __builtin_amdgcn_sinf(rev[r - 3]), cv = __builtin_amdgcn_cosf(rev[r - 3]);
                v[q] = h ? cv - (float)(E)cv : sv - (float)(E)sv;
            } else {
                v[q] = 0.f;
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int w = 0; w < 4; ++w) Bp[ks][w] = pack2<E>(v[8 * ks + 2 * w], v[8 * ks + 2 * w + 1]);
}

// one network: L0 (encoding) .. L7, then the <= 32-row head; returns the head accumulator (bias included)
template <typename E, int NW, bool LAST, int ACT, int PEL, bool LO>
__device__ __forceinline__ f32x16 run_net(Pipe<E, NW>& P, const float (&x)[3], const float* bias /* 8 layer rows + head row */, int h) {
    u32x4 B0[16], B1[16], Bp[4];
    f32x16 accA, accB;
    pe_frags<E, PEL, LO>(Bp, x, h);
    // every layer starts on a stage boundary (32, 128, 160 and 16 fragments are multiples of 16)
    layer<E, NW, 4, ACT, ACT, false>(P, accA, accB, B0 /* unused */, Bp, B0, bias, h);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 256, h);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 512, h);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 768, h);
    layer<E, NW, 20, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 1024, h);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 1280, h);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 1536, h);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 1792, h);
    row_block<E, NW, 0, 16, ACT, true, true, LAST>(P, accA, accB, B1, Bp, B1[14], B1[15], bias + 2048, h);
    return accA;
}

template <typename E, int NW>
__global__ __launch_bounds__(64 * NW, 2) void mlp_sdf_stream_kernel(GeoNet net, const void* __restrict__ stream, const float* __restrict__ ba,
                                                                      FrameState fr, MlpIO io) {
    __shared__ __attribute__((aligned(16))) StSmem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    // bias table: resd rows (L0 / L4 carry the per-frame pose condition), resd head, scaled sdf rows, sdf head
    constexpr int ST_TM = 32 * NW;     // points per workgroup tile
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
    if (blockIdx.x == 0 && tid == 0 && io.counters) atomicAdd(&io.counters->n_fine_sdf, (unsigned long long)count);
    const int ntiles = (count + ST_TM - 1) / ST_TM;
    if ((int)blockIdx.x >= ntiles) return;

    Pipe<E, NW> P;
    P.g = reinterpret_cast<const char*>(stream);
    P.voff = wave * (16 / NW) * 1024 + lane * 16;
    P.ring = reinterpret_cast<const char*>(sm.ring) + lane * 16;
    P.ring_addr = (unsigned)(size_t)sm.ring + wave * (16 / NW) * 1024;
    P.slot = ST_RING - 1;            // the first sync_stage() advances to slot 0 / stream stage 0
    P.sstage = ST_STAGES - 1;
    P.rd = P.ring;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < ST_AHEAD; ++st) P.issue(st, st);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s = tile * ST_TM + wave * 32 + c;
        float x[3] = {0.f, 0.f, 0.f};
        int pidx = 0;
        float smpl = 0.f;
        if (s < count) {
            x[0] = io.bpts[3 * s]; x[1] = io.bpts[3 * s + 1]; x[2] = io.bpts[3 * s + 2];
            pidx = io.idx[s];
            if (io.smooth) smpl = io.sdf[pidx];
        }
        // the first ST_PF fragments of the tile (stage 0 of the stream)
        P.template fetch<0>(); P.template fetch<1>(); P.template fetch<2>(); P.template fetch<3>();
        // ---- residual deformation net (ReLU); head: resd = tanh(z) * resd_limit, cpts = bpts + resd
        const f32x16 hr = run_net<E, NW, false, ACT_RELU, 10, false>(P, x, sm.bias, h);
        float cp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float r = tanhf(hr[k]) * io.resd_limit;           // valid in lanes h = 0 (rows 0..2)
            cp[k] = x[k] + __shfl(r, c);
        }
        // ---- signed distance net (softplus, scaled domain); head row 0 = sdf
        const f32x16 hs = run_net<E, NW, true, ACT_SOFTPLUS, 8, true>(P, cp, sm.bias + 9 * 256, h);
        if (h == 0 && s < count) {
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

}  // namespace

template <int NW>
static void launch_nw(const GeoNet& net, const void* sarena, const float* barena, const FrameState& fr, const MlpIO& io,
                      int max_slots, bool f16w, hipStream_t stream) {
    const int tiles = (max_slots + 32 * NW - 1) / (32 * NW);
    const int grid = tiles < 256 ? tiles : 256;     // one workgroup per CU (the weight ring fills its LDS), persistent over tiles
    if (f16w) hipLaunchKernelGGL((mlp_sdf_stream_kernel<f16, NW>), dim3(grid), dim3(64 * NW), 0, stream, net, sarena, barena, fr, io);
    else if (NW == 8) hipLaunchKernelGGL((mlp_sdf_stream_kernel<bf16, 8>), dim3(grid), dim3(64 * NW), 0, stream, net, sarena, barena, fr, io);
}

void launch_mlp_sdf_stream(const GeoNet& net, const void* sarena, const float* barena, const FrameState& fr, const MlpIO& io,
                           int max_slots, bool f16w, hipStream_t stream) {
    if (max_slots <= 0) return;
    // A launch that cannot fill the 256 CUs with 256-point tiles is bound by the latency of ONE tile (1952 MFMAs per
    // wave): narrower workgroups put one wave on a SIMD instead of two (86 -> 67 -> 58 us per tile for 8 / 4 / 2 waves).
    // max_slots is only an upper bound of the device-side count: up to 65536 the 4-wave variant needs at most the two
    // rounds that equal one 8-wave round, and one when the real count is below half.
    static const int force = getenv("RA_STREAM_NW") ? atoi(getenv("RA_STREAM_NW")) : 0;
    const int nw = !f16w ? 8 : (force ? force : (max_slots <= 256 * 64 ? 2 : (max_slots <= 256 * 256 ? 4 : 8)));   // bf16 (A/B only): wide variant
#ifndef RA_STREAM_WIDE_ONLY
    if (nw == 2) { launch_nw<2>(net, sarena, barena, fr, io, max_slots, f16w, stream); return; }
    if (nw == 4) { launch_nw<4>(net, sarena, barena, fr, io, max_slots, f16w, stream); return; }
#endif
    launch_nw<8>(net, sarena, barena, fr, io, max_slots, f16w, stream);
}
