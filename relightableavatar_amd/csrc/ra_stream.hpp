// Shared machinery of the streamed-weight MLP kernels (ra_mlp_stream.hip: K3 distance query; ra_mlp_grad.hip: K4 full
// query in reverse mode): element traits, the LDS weight ring fed by one LDS-DMA stream per workgroup, and the row-block
// primitive whose D fragment is, after the epilogue, the next layer's B fragment.  See ra_mlp_stream.hip for the design notes.
#pragma once
#include "ra_common.hpp"
#include <type_traits>


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
constexpr int ST_FRAGS = 1952;                      // fragments per tile, every layer padded to 256 rows / 16-wide k-steps of 256 inputs
constexpr int ST_STAGES = ST_FRAGS / 16;            // 122
// the 8-wave K3 walks a TRIMMED stream: the SDF net's lin3 has 205 outputs (7 row blocks, not 8) and lin4 reads them in 14 k-steps
// (+ 4 encoding k-steps = 18, not 20): 32 MFMAs per tile less (1.6 %)
constexpr int ST_FRAGS_TRIM = ST_FRAGS - 16 - 16;    // 1920
constexpr int ST_STAGES_TRIM = ST_FRAGS_TRIM / 16;   // 120
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
    static constexpr bool is_f16 = false;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Tr<f16> {
    typedef f16x8 x8; typedef f16x2 x2;
    static constexpr bool is_f16 = true;
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

// scaled-domain softplus y' = log2(1 + 2^z'), z' = beta*log2(e) * pre-activation, in FOUR VALU ops per element:
// v_exp, v_add, v_log and ONE v_med3 that also handles the overflow of 2^z' (z' >= 128: log2(1 + inf) = inf):
// c = log2(1 + 2^z) lies in [z, 64] for z <= 64, so med3(c, z, 64) = c there, and c >= z > 64 beyond, where med3 = z =
// softplus to fp32 precision (the next term is 2^-64).  The earlier form max(z, 0) + log2(1 + 2^-|z|) took five; the kernel is
// power-limited, so every VALU op per element costs about 1 ns per MFMA slot whether or not it "fits".
__device__ __forceinline__ float sp_finish(float c, float z) { return __builtin_amdgcn_fmed3f(c, z, 64.f); }

template <int ACT>
__device__ __forceinline__ float act(float z) {
    if (ACT == ACT_RELU) return max0(z);
    return sp_finish(__builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(z)), z);
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
// PF: A fragments read ahead of their MFMA (4 with two waves per SIMD; a lone wave per SIMD needs 8 to cover the LDS latency)
template <typename E, int NW, int STAGES = ST_STAGES, int PF_ = ST_PF>
struct Pipe {
    static constexpr int FPW = 16 / NW;
    static constexpr int PF = PF_;
    static constexpr int MOD = 16;            // fetch<FM>: FM = position in a 16-fragment stage
    static constexpr bool LONE = NW < 8;      // one wave per SIMD
    const char* g;          // weight stream (uniform)
    unsigned voff;          // per lane: wave * FPW * 1024 + lane * 16
    const char* ring;       // LDS ring (generic pointer), + lane * 16
    unsigned ring_addr;     // LDS byte address of the ring + wave * FPW * 1024 (wave-uniform)
    unsigned slot;          // ring slot of the stage being read (wave-uniform)
    int sstage;             // its position in the tile's stream, 0 .. ST_STAGES-1 (wave-uniform)
    const char* rd;         // ring + slot * 16 KB + lane * 16
    typename Tr<E>::x8 af[PF];

    __device__ __forceinline__ void issue(int stream_stage, unsigned ring_slot) {
        const char* sb = g;
        asm volatile("" : "+s"(sb));            // keeps the 244 per-stage addresses from being precomputed (and spilled)
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + ring_slot * ST_STAGE_BYTES);
        const char* src = sb + (size_t)stream_stage * ST_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < FPW / 2; ++j) glds16x2(src, voff + j * 2048, voff + j * 2048 + 1024, dst + j * 2048);
    }
    // the next stage of the stream becomes readable; the ring slot two stages back is refilled ST_AHEAD stages ahead.
    // The stage position is run-time state (SGPRs), so the code below only depends on a fragment's position in its stage.
    __device__ __forceinline__ void sync_stage() {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FPW * (ST_AHEAD - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = (slot + 1) & (ST_RING - 1);
        sstage = sstage + 1 == STAGES ? 0 : sstage + 1;
        int ahead = sstage + ST_AHEAD;
        ahead = ahead >= STAGES ? ahead - STAGES : ahead;
        issue(ahead, (slot + ST_AHEAD) & (ST_RING - 1));
        rd = ring + slot * ST_STAGE_BYTES;
    }
    // FM: position of the fragment in its 16-fragment stage
    template <int FM>
    __device__ __forceinline__ void fetch() {
        if (FM == 0) sync_stage();
        af[FM % PF] = *reinterpret_cast<const typename Tr<E>::x8*>(rd + FM * 1024);
    }
    template <int FH, int FL> __device__ __forceinline__ void ready() {}       // (PipeReg, ra_k3cc.hpp: the wait for a k-step's fragments)
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
// KH: hidden k-steps of the layer (KS - KH encoding k-steps follow); ELAST: with EARLY, the last slot whose results may still be
// written (the pending outputs are this block's own inputs at k-steps 2 PRB - 2, 2 PRB - 1 for a previous layer of PRB row blocks)
template <typename E, int NW, int FM0, int KS, int ACT_PREV, bool PENDING, bool EARLY, bool TAIL, typename PipeT, int KH = (KS == 4 ? 0 : 16), int ELAST = 13>
__device__ __forceinline__ void row_block(PipeT& P, f32x16& acc, const f32x16& accPrev, u32x4 (&Bm)[16], const u32x4 (&Bp)[4],
                                          u32x4& o0, u32x4& o1, const float* bias_rb, int h) {
    init_acc(acc, bias_rb, h);
    float ta[16];
    static_for<0, KS>([&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        const u32x4 bw = ks < KH ? Bm[ks & 15] : Bp[(ks - KH) & 3];
        acc = Tr<E>::mfma(P.af[(FM0 + ks) % ST_PF], __builtin_bit_cast(X8<E>, bw), acc);
        if constexpr (!(TAIL && ks + ST_PF >= KS)) P.template fetch<(FM0 + ks + ST_PF) % 16>();
        if constexpr (PENDING) {
            // Pending row block, element by element.  The softplus chain exp2 -> +1 -> log2 -> +max is software-pipelined
            // over four MFMA slots so that no VALU op waits on one issued in the same slot (in-order issue: a stalled
            // transcendental chain would hold back the next MFMA).  LAST: last slot whose results may still be written.
            static_for<0, 16>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                constexpr bool SP = ACT_PREV == ACT_SOFTPLUS;
                constexpr int DEPTH = SP ? 3 : 0;
                constexpr int LAST = (KS == 4) ? 3 : (EARLY ? ELAST : KS - 1);
                constexpr int s0 = (KS == 4) ? 0 : (e * (LAST - DEPTH + 1)) / 16;
                if constexpr (SP && KS != 4) {
                    if constexpr (s0 == ks) ta[e] = __builtin_amdgcn_exp2f(accPrev[e]);
                    if constexpr (s0 + 1 == ks) ta[e] = 1.f + ta[e];
                    if constexpr (s0 + 2 == ks) ta[e] = __builtin_amdgcn_logf(ta[e]);
                    if constexpr (s0 + 3 == ks) ta[e] = sp_finish(ta[e], accPrev[e]);
                } else if constexpr (SP) {          // 4-MFMA row blocks (first layer): one stage of all 16 elements per slot
                    if constexpr (ks == 0) ta[e] = __builtin_amdgcn_exp2f(accPrev[e]);
                    if constexpr (ks == 1) ta[e] = 1.f + ta[e];
                    if constexpr (ks == 2) ta[e] = __builtin_amdgcn_logf(ta[e]);
                    if constexpr (ks == 3) ta[e] = sp_finish(ta[e], accPrev[e]);
                } else {
                    // f16: ReLU AFTER the pack (v_cvt_pk + v_pk_max_f16: 1.0 instead of 1.5 VALU per element).  Rounding is monotone
                    // and keeps the sign, so max(round(z), 0) == round(max(z, 0)): bitwise identical distances on 200 k points,
                    // -6.6 % VALU instructions, -0.5 .. -0.8 % kernel time (same-box A/B) — VALU count is not what bounds K3.
                    if constexpr ((KS == 4 ? e / 4 : s0) == ks) ta[e] = Tr<E>::is_f16 ? accPrev[e] : max0(accPrev[e]);
                }
                constexpr int sdone = (KS == 4) ? (SP ? 3 : e / 4) : s0 + DEPTH;
                if constexpr ((e & 1) && sdone == ks) {
                    unsigned w = pack2<E>(ta[e - 1], ta[e]);
                    if constexpr (!SP && Tr<E>::is_f16) {
                        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                        h2_t v = __builtin_bit_cast(h2_t, w);
                        v = __builtin_elementwise_max(v, h2_t{(_Float16)0, (_Float16)0});
                        w = __builtin_bit_cast(unsigned, v);
                    }
                    if constexpr (e < 8) o0[e >> 1] = w; else o1[(e >> 1) & 3] = w;
                }
            });
        }
    });
}

// a layer of NRB row blocks (8: 256 rows); on entry `accB` holds the pending last row block of the previous layer (if PEND_IN,
// activation ACT_IN; that layer had PRB row blocks: destination Bm[2 PRB - 2], Bm[2 PRB - 1]); on exit this layer's last row block is
// pending — in accB for an even NRB, in accA for an odd one (the caller swaps the two for the layers that follow).
template <typename E, int NW, int KS, int ACT, int ACT_IN, bool PEND_IN, typename PipeT, int NRB = 8, int KH = (KS == 4 ? 0 : 16), int PRB = 8>
__device__ __forceinline__ void layer(PipeT& P, f32x16& accA, f32x16& accB, u32x4 (&Bm)[16], const u32x4 (&Bp)[4], u32x4 (&Bo)[16],
                                      const float* bias, int h) {
    row_block<E, NW, 0, KS, ACT_IN, PEND_IN, true, false, PipeT, KH, 2 * PRB - 3>(P, accA, accB, Bm, Bp, Bm[2 * PRB - 2], Bm[2 * PRB - 1], bias, h);
    row_block<E, NW, (1 * KS) % 16, KS, ACT, true, false, false, PipeT, KH>(P, accB, accA, Bm, Bp, Bo[0], Bo[1], bias + 32, h);
    row_block<E, NW, (2 * KS) % 16, KS, ACT, true, false, false, PipeT, KH>(P, accA, accB, Bm, Bp, Bo[2], Bo[3], bias + 64, h);
    row_block<E, NW, (3 * KS) % 16, KS, ACT, true, false, false, PipeT, KH>(P, accB, accA, Bm, Bp, Bo[4], Bo[5], bias + 96, h);
    row_block<E, NW, (4 * KS) % 16, KS, ACT, true, false, false, PipeT, KH>(P, accA, accB, Bm, Bp, Bo[6], Bo[7], bias + 128, h);
    row_block<E, NW, (5 * KS) % 16, KS, ACT, true, false, false, PipeT, KH>(P, accB, accA, Bm, Bp, Bo[8], Bo[9], bias + 160, h);
    row_block<E, NW, (6 * KS) % 16, KS, ACT, true, false, false, PipeT, KH>(P, accA, accB, Bm, Bp, Bo[10], Bo[11], bias + 192, h);
    if constexpr (NRB == 8)
        row_block<E, NW, (7 * KS) % 16, KS, ACT, true, false, false, PipeT, KH>(P, accB, accA, Bm, Bp, Bo[12], Bo[13], bias + 224, h);
}


// ---- paired row blocks: the latency variant for launches that cannot fill the chip -------------------------------------------
// With ONE wave per SIMD the 16 MFMAs of a row block are a dependent chain on one accumulator and issue every ~64 cycles
// (measured: tools/k3_timestamps.py, 2-wave launch).  Two row blocks in flight — fragments interleaved k-step by k-step in the
// stream (ra_pack.cpp, `pair` order) — give the pipe an independent MFMA every slot.  Each accumulator still sums its k-steps
// in the same order, so the results are bit-identical to the unpaired kernel.
// NB = 2: blocks a, b (accumulators acc0, acc1); NB = 1: a lone block (the heads).  The pending epilogue is the previous PAIR
// (prev0 -> o00, o01; prev1 -> o10, o11: 32 elements over the NB * KS slots; EARLY: they are this block's own k-steps 12..15).
template <typename E, int NW, int NB, int FM0, int KS, int ACT_PREV, bool PENDING, bool EARLY, bool TAIL, typename PipeT>
__device__ __forceinline__ void row_blocks(PipeT& P, f32x16& acc0, f32x16& acc1, const f32x16& prev0, const f32x16& prev1, u32x4 (&Bm)[16],
                                           const u32x4 (&Bp)[4], u32x4& o00, u32x4& o01, u32x4& o10, u32x4& o11, const float* bias_rb, int h) {
    init_acc(acc0, bias_rb, h);
    if constexpr (NB == 2) init_acc(acc1, bias_rb + 32, h);
    float ta[32];
    constexpr int NS = NB * KS;
    static_for<0, NS>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int ks = i / NB;
        const u32x4 bw = (KS == 4) ? Bp[ks & 3] : (ks < 16 ? Bm[ks & 15] : Bp[ks & 3]);
        constexpr int PF = PipeT::PF;
        if constexpr (NB == 1 || (i & 1) == 0) acc0 = Tr<E>::mfma(P.af[(FM0 + i) % PF], __builtin_bit_cast(X8<E>, bw), acc0);
        else acc1 = Tr<E>::mfma(P.af[(FM0 + i) % PF], __builtin_bit_cast(X8<E>, bw), acc1);
        if constexpr (!(TAIL && i + PF >= NS)) P.template fetch<(FM0 + i + PF) % 16>();
        if constexpr (PENDING) {
            static_for<0, 32>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                constexpr bool SP = ACT_PREV == ACT_SOFTPLUS;
                constexpr int DEPTH = SP ? 3 : 0;
                constexpr int LAST = (KS == 4) ? NS - 1 : (EARLY ? NB * 12 - 1 : NS - 1);
                // 4-k-step blocks (first layers): the 16 elements of a block move through the chain together, one stage per slot
                constexpr int s0 = (KS == 4) ? (NB == 2 ? 4 * (e >> 4) + (SP ? 0 : (e & 15) / 4) : (SP ? 0 : e / 8)) : (e * (LAST - DEPTH + 1)) / 32;
                const float z = e < 16 ? prev0[e & 15] : prev1[e & 15];
                if constexpr (SP) {
                    if constexpr (s0 == i) ta[e] = __builtin_amdgcn_exp2f(z);
                    if constexpr (s0 + 1 == i) ta[e] = 1.f + ta[e];
                    if constexpr (s0 + 2 == i) ta[e] = __builtin_amdgcn_logf(ta[e]);
                    if constexpr (s0 + 3 == i) ta[e] = sp_finish(ta[e], z);
                } else {
                    if constexpr (s0 == i) ta[e] = Tr<E>::is_f16 ? z : max0(z);        // f16: ReLU after the pack (see row_block)
                }
                constexpr int sdone = s0 + DEPTH;
                if constexpr ((e & 1) && sdone == i) {
                    unsigned w = pack2<E>(ta[e - 1], ta[e]);
                    if constexpr (!SP && Tr<E>::is_f16) {
                        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                        h2_t v = __builtin_bit_cast(h2_t, w);
                        v = __builtin_elementwise_max(v, h2_t{(_Float16)0, (_Float16)0});
                        w = __builtin_bit_cast(unsigned, v);
                    }
                    constexpr int q = e & 15;
                    if constexpr (e < 16) { if constexpr (q < 8) o00[q >> 1] = w; else o01[(q >> 1) & 3] = w; }
                    else { if constexpr (q < 8) o10[q >> 1] = w; else o11[(q >> 1) & 3] = w; }
                }
            });
        }
        // a lone wave has nobody to hide its LDS latency: keep the slot's instructions in source order, so that the fragment of
        // slot i + PF really is requested PF slots ahead (left alone the scheduler sinks the reads to two slots before their use:
        // 61 instead of ~40 cycles per MFMA)
        __builtin_amdgcn_sched_barrier(0);
    });
}

// a 256-row layer as four pairs of row blocks.  On entry (b0, b1) hold the pending last pair of the previous layer (if PEND_IN:
// activation ACT_IN, destination Bm[12..15]); on exit they hold this layer's pending blocks 6, 7.
template <typename E, int NW, int KS, int ACT, int ACT_IN, bool PEND_IN, typename PipeT>
__device__ __forceinline__ void layer_pairs(PipeT& P, f32x16& a0, f32x16& a1, f32x16& b0, f32x16& b1, u32x4 (&Bm)[16], const u32x4 (&Bp)[4], u32x4 (&Bo)[16],
                                            const float* bias, int h) {
    row_blocks<E, NW, 2, 0, KS, ACT_IN, PEND_IN, true, false>(P, a0, a1, b0, b1, Bm, Bp, Bm[12], Bm[13], Bm[14], Bm[15], bias, h);
    row_blocks<E, NW, 2, (2 * KS) % 16, KS, ACT, true, false, false>(P, b0, b1, a0, a1, Bm, Bp, Bo[0], Bo[1], Bo[2], Bo[3], bias + 64, h);
    row_blocks<E, NW, 2, (4 * KS) % 16, KS, ACT, true, false, false>(P, a0, a1, b0, b1, Bm, Bp, Bo[4], Bo[5], Bo[6], Bo[7], bias + 128, h);
    row_blocks<E, NW, 2, (6 * KS) % 16, KS, ACT, true, false, false>(P, b0, b1, a0, a1, Bm, Bp, Bo[8], Bo[9], Bo[10], Bo[11], bias + 192, h);
}

// ---- compensated row blocks (K3C, ra_k3c.hpp): near-fp32 products from f16 MFMAs -----------------------------------------------
// Both operands are carried as hi + lo pairs of IEEE halves (x = hi + lo exactly to 22 bits; the lo parts of small values are f16
// subnormals, which the matrix pipe multiplies exactly) and a k-step is three MFMAs into ONE fp32 accumulator:
//     acc += Ah Bl;  acc += Al Bh;  acc += Ah Bh                       (Al Bl, 2^-22 of the product, is dropped)
// The weight stream holds every fragment twice, [hi | lo] per k-step (ra_pack.cpp StreamBuilder::add16); the pending epilogue of the
// previous row block (activation in fp32, then hi = f16(a), lo = f16(a - hi)) is spread over the 3 KS MFMA slots.
//
// Tile shape: v_mfma_f32_16x16x32_f16 — a wave owns 16 points, a row block is 16 output rows (16 per layer), a k-step 32 inputs (8 per
// hidden layer).  K3C serves the surface trace, whose launches never fill the chip: with 16 points per wave a wave's chain of dependent
// MFMAs is half as long as with the 32x32x16 tile of the plain kernel (16 against 32 cycles each, the same count), the two B fragment
// sets (hi, lo; in, out) are 128 registers instead of 256, so two waves share a SIMD, and a small launch spreads over twice as many SIMDs.
// (The first K3C used the 32x32x16 tile, one wave per SIMD: 150 us per 128-point tile)
// D fragment: lane (point n = lane & 15, row group g = lane >> 4) holds rows 16 rb + 4 g + i; the packed D fragments of row blocks 2 m and
// 2 m + 1 side by side are the next layer's B fragment of k-step m (ra_pack.cpp hidden_feature16).
// One accumulator chain: a second chain for the two small products (round 4, removed) measured slower on both tiles — 165 against 150 us on
// the 32x32 tile (extra AGPR traffic), 106 against 98 us for 2 400 points on this one — so the chain of dependent MFMAs is not what bounds a
// lone wave here; the ~5 non-MFMA instructions it has to issue per 16-cycle MFMA are (5808 MFMAs in 98 us = 37 cycles each).
constexpr int K3C_E0 = 2;    // MFMA slots before the pending epilogue first touches the previous accumulator (its last MFMA is still in the pipe)
__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// One row block: 3 KS MFMAs (fragments FM0.. of the stage, [hi | lo] per k-step) into `acc`, interleaved with the pending epilogue of
// `accPrev` (activation ACT_PREV), whose four values per lane go to registers 2 DH, 2 DH + 1 of fragment DI of dstH / dstL.
// KH hidden k-steps (from BmH / BmL), then KS - KH encoding k-steps (BpH / BpL).  ELAST: with EARLY, the last slot that may still write.
struct Acc16 {               // bias + sum of the three products of every k-step
    f32x4 m;
    __device__ __forceinline__ float val(int e) const { return m[e]; }
};

template <int FM0, int KS, int ACT_PREV, bool PENDING, bool EARLY, bool TAIL, int DI, int DH, typename PipeT, int KH = (KS == 2 ? 0 : 8), int ELAST = 20>
__device__ __forceinline__ void row_block_16(PipeT& P, Acc16& acc, const Acc16& accPrev, u32x4 (&BmH)[8], u32x4 (&BmL)[8], const u32x4 (&BpH)[2],
                                             const u32x4 (&BpL)[2], u32x4 (&dstH)[8], u32x4 (&dstL)[8], const float* bias_rb, int g) {
    typedef f16 E;
    acc.m = *reinterpret_cast<const f32x4*>(bias_rb + 4 * g);          // rows 16 rb + 4 g + i start at their bias
    float ta[4];
    constexpr int NS = 3 * KS, PF = PipeT::PF, PFK = PF / 2;       // PF fragments = PFK k-steps read ahead
    static_for<0, NS>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int ks = i / 3, m = i % 3;
        const u32x4 bh = ks < KH ? BmH[ks & 7] : BpH[(ks - KH) & 1];
        const u32x4 bl = ks < KH ? BmL[ks & 7] : BpL[(ks - KH) & 1];
        constexpr int fh = (FM0 + 2 * ks) % PF, fl = (FM0 + 2 * ks + 1) % PF;
        if constexpr (m == 0) P.template ready<fh, fl>();
        if constexpr (m == 0) acc.m = mfma16(P.af[fh], __builtin_bit_cast(f16x8, bl), acc.m);
        if constexpr (m == 1) acc.m = mfma16(P.af[fl], __builtin_bit_cast(f16x8, bh), acc.m);
        if constexpr (m == 2) acc.m = mfma16(P.af[fh], __builtin_bit_cast(f16x8, bh), acc.m);
        if constexpr (m == 2) {
            if constexpr (!(TAIL && ks + PFK >= KS)) {
                P.template fetch<(FM0 + 2 * (ks + PFK)) % PipeT::MOD>();            // hi first: position 0 of a stage turns the ring
                P.template fetch<(FM0 + 2 * (ks + PFK) + 1) % PipeT::MOD>();
            }
        }
        if constexpr (PENDING) {
            static_for<0, 4>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                constexpr bool SP = ACT_PREV == ACT_SOFTPLUS;
                constexpr int DA = SP ? 3 : 0;                       // slots until the activation's value exists
                constexpr int LAST = EARLY ? ELAST : NS - 1;         // last slot that may still write the outputs
                constexpr int E0 = NS >= 24 ? K3C_E0 : 0;
                constexpr int s0 = E0 + (e * (LAST - E0 - (DA + 2) + 1)) / 4;
                if constexpr (SP) {
                    if constexpr (s0 == i) ta[e] = __builtin_amdgcn_exp2f(accPrev.val(e));
                    if constexpr (s0 + 1 == i) ta[e] = 1.f + ta[e];
                    if constexpr (s0 + 2 == i) ta[e] = __builtin_amdgcn_logf(ta[e]);
                    if constexpr (s0 + 3 == i) ta[e] = sp_finish(ta[e], accPrev.val(e));
                } else {
                    if constexpr (s0 == i) ta[e] = max0(accPrev.val(e));
                }
                if constexpr ((e & 1) && s0 + DA + 1 == i) {          // hi halves of the pair (e - 1, e); ta keeps the residuals
                    f16x2 hv;
                    hv[0] = (f16)ta[e - 1]; hv[1] = (f16)ta[e];
                    dstH[DI][2 * DH + (e >> 1)] = __builtin_bit_cast(unsigned, hv);
                    ta[e - 1] -= (float)hv[0];
                    ta[e] -= (float)hv[1];
                }
                if constexpr ((e & 1) && s0 + DA + 2 == i) dstL[DI][2 * DH + (e >> 1)] = pack2<E>(ta[e - 1], ta[e]);
            });
        }
        if (PipeT::LONE) __builtin_amdgcn_sched_barrier(0);        // a lone wave per SIMD: keep the reads PF fragments ahead (see row_blocks)
    });
}

// a 256-row layer = 16 compensated row blocks; on entry accB holds the pending last row block of the previous layer (if PEND_IN: it
// completes this layer's own input fragment 7), on exit this layer's row block 15 is pending in accB
template <int KS, int ACT, int ACT_IN, bool PEND_IN, typename PipeT>
__device__ __forceinline__ void layer_16(PipeT& P, Acc16& accA, Acc16& accB, u32x4 (&BmH)[8], u32x4 (&BmL)[8], const u32x4 (&BpH)[2], const u32x4 (&BpL)[2],
                                         u32x4 (&BoH)[8], u32x4 (&BoL)[8], const float* bias, int g) {
    constexpr int KH = KS == 2 ? 0 : 8, F = 2 * KS;     // fragments per row block
    row_block_16<0, KS, ACT_IN, PEND_IN, true, false, 7, 1, PipeT, KH>(P, accA, accB, BmH, BmL, BpH, BpL, BmH, BmL, bias, g);
    static_for<1, 16>([&](auto rb_) {
        constexpr int rb = decltype(rb_)::value;
        if constexpr (rb & 1) row_block_16<(rb * F) % 16, KS, ACT, true, false, false, ((rb - 1) >> 1), ((rb - 1) & 1), PipeT, KH>(P, accB, accA, BmH, BmL, BpH, BpL, BoH, BoL, bias + 16 * rb, g);
        else row_block_16<(rb * F) % 16, KS, ACT, true, false, false, ((rb - 1) >> 1), ((rb - 1) & 1), PipeT, KH>(P, accA, accB, BmH, BmL, BpH, BpL, BoH, BoL, bias + 16 * rb, g);
    });
}

}  // namespace
