// K4: the full query (geometry with normal + material / colour heads) in REVERSE mode on the streamed-weight machinery of
// K3 (ra_stream.hpp).  Implementation header: ra_k4_{fwd,bwd}_{f16,bf16}.hip instantiate one kernel family each (parallel builds).
//
//   reference: forward_geometry + take_gradient   lib/networks/deform/base_network.py:456-494, lib/utils/net_utils.py:1215-1239
//              material heads                      lib/networks/relight/relight_network.py:45-47,91-104
//              RenderNetwork                       lib/networks/deform/base_network.py:152-171,496-515
//
// Forward-mode tangents (three columns next to every point, the round-1 kernel) cost 4x the forward FLOPs.  d sdf / d bpts is a
// gradient of ONE output, i.e. what reverse mode is for: one forward pass that remembers the activations, one backward pass
// through the transposed layers.
//
//   * forward kernel (mlp_fwd_tape_kernel) = K3 + a tape: every hidden row block's packed f16 B fragments (softplus net: the
//     activations y', from which sigma' = 1 - 2^-y' in the scaled domain; ReLU net: sign bits only, 16 per row block via
//     v_alignbit), the 256 feature rows of lin8 as B fragments, and (cpts, sdf) per point.  4.9 KB per point, written in the
//     order the row blocks complete (whole 1 KB wave stores).
//   * backward kernel (mlp_bwd_heads_kernel): u = W_l^T delta_l is the same streamed-weight row-block product with the
//     transposed matrices (packed with the same K permutation, ra_pack.cpp), the epilogue multiplies by the tape's derivative
//     instead of applying the activation — the D fragment of a transposed layer is again the next B fragment.  The
//     encoding-fed layers end in 64 rows of encoding-channel gradients, which the epilogue contracts with the encoding's
//     Jacobian on the fly (d sin(2^f x) = 2^f cos(2^f x), one v_cos per channel).  sdf net first (gradient wrt cpts), then the
//     residual net seeded with resd_limit (1 - tanh^2) g (cpts = bpts + resd(bpts)), then the material heads or the colour
//     net on the taped features, then the per-point epilogue (normal transforms, occupancy, raw channels).
//     Tape reads are plain (compiler-visible) non-temporal loads requested two row blocks ahead and first used in the MFMA slot
//     right before a stage turnover (TapeQ below explains why not inline asm with counted waits).
//   * gradients are carried times GRAD_SCALE (2^4) for f16 headroom; everything acting on them is linear.
#include "ra_stream.hpp"

namespace {

constexpr int FW_STAGES = 130;                         // 2080 fragments: gen-3 stream + 128 fragments of feature rows
constexpr float GRAD_SCALE = 16.f;
// waves per workgroup (8: two waves per SIMD, 256 registers each; 4: one wave per SIMD, 512 registers).  The tape is indexed
// by 32-point group (slot / 32), so the two kernels need not agree.  Forward: 4 — at 8 the tape stores and sign-bit
// bookkeeping push it to ~90 spilled registers (6.4 ms against 5.6 ms per 2.1 M points)
#ifndef RA_K4_NW_F
#define RA_K4_NW_F 8
#endif
#ifndef RA_K4_NW_B
#define RA_K4_NW_B 8
#endif
constexpr int FW_BIAS_ROWS = 19;                       // + feature rows' bias
// tape of one wave-tile (32 points), bytes
constexpr int TP_SDF = 0;                              // 8 layers x 8 row blocks x 2 fragments x 1 KB
constexpr int TP_FEAT = 131072;                        // 16 fragments
constexpr int TP_BITS = 147456;                        // residual net: 8 layers x 4 dwords x 64 lanes
constexpr int TP_GEO = 155648;                         // per point (lanes 0..31): cpts xyz, sdf
constexpr int TP_WAVE = 156672;

#ifdef RA_DBG_SLOTS
constexpr bool DBGE = true;
#else
constexpr bool DBGE = false;      // per-encoding-slot gradient dump (tools/dbg_grad.py RA_DBG_PE): build with -DRA_DBG_SLOTS
#endif
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_SOFTPLUS = 2, EPI_LINEAR = 3, EPI_RELU_BITS = 4, EPI_GRAD_RELU = 5, EPI_GRAD_SP = 6, EPI_PEJAC = 7 };

struct EpiAux {
    u32x4 t0, t1;        // EPI_GRAD_SP: taped activations y' of the pending row block (f16 pairs, D-fragment order)
    u32x4 n0, n1;        // ... of the NEXT pending row block, taken from the tape queue just before this block's weight-stage turnover
    unsigned bits;       // EPI_RELU_BITS: shift register of sign bits; EPI_GRAD_RELU: the pending block's 16 bits in [15:0] (bit 15-e)
    float scale;         // EPI_LINEAR: factor applied before packing
    char* st;            // forward: where the next finished row block's two fragments go (per lane), nullptr = no tape
    float g[3];          // EPI_PEJAC: running encoding-input gradient of this lane half
    float rev[3];        // EPI_PEJAC: encoding input in revolutions (x / 2 pi)
    float* dbg;          // debugging aid: this lane half's 32 encoding-slot gradients are accumulated here (nullable)
};

// number of fine slots of the sub-batch [slot0, slot0 + slot_cap) of the compacted list (device-side total *io.count)
__device__ __forceinline__ int batch_count(const FullIO& io) {
    const int left = *io.count - io.slot0;
    return left < 0 ? 0 : (left < io.slot_cap ? left : io.slot_cap);
}

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }

template <typename E>
__device__ __forceinline__ float half_of(unsigned w, int k) {
    const typename Tr<E>::x2 v = __builtin_bit_cast(typename Tr<E>::x2, w);
    return (float)v[k];
}

// one element of the pending epilogue at MFMA slot ks; L: encoding frequencies (EPI_PEJAC), LO: sdf-style slots
template <typename E, int EPI, int KS, bool EARLY, int L, bool LO, int PB>
struct Epi {
    static constexpr bool SP = EPI == EPI_SOFTPLUS;
    static constexpr int DEPTH = SP ? 3 : (EPI == EPI_GRAD_SP || EPI == EPI_PEJAC ? 1 : 0);
    static constexpr int LAST = (KS == 4) ? 3 : (EARLY ? 13 : KS - 1);
    template <int ks, int e>
    static __device__ __forceinline__ void step(const f32x16& a, float (&ta)[16], float (&tb)[16], u32x4& o0, u32x4& o1, EpiAux& x, int h) {
        constexpr int s0 = (KS == 4) ? (DEPTH ? 0 : e / 4) : (e * (LAST - DEPTH + 1)) / 16;
        if constexpr (SP) {          // ra_stream.hpp, sp_finish
            if constexpr (s0 == ks) ta[e] = __builtin_amdgcn_exp2f(a[e]);
            if constexpr (s0 + 1 == ks) ta[e] = 1.f + ta[e];
            if constexpr (s0 + 2 == ks) ta[e] = __builtin_amdgcn_logf(ta[e]);
            if constexpr (s0 + 3 == ks) ta[e] = sp_finish(ta[e], a[e]);
        } else if constexpr (EPI == EPI_GRAD_SP) {
            if constexpr (s0 == ks) ta[e] = __builtin_amdgcn_exp2f(-half_of<E>(e < 8 ? x.t0[e >> 1] : x.t1[(e >> 1) & 3], e & 1));
            if constexpr (s0 + 1 == ks) ta[e] = __builtin_fmaf(-ta[e], a[e], a[e]);          // a * (1 - 2^-y') = a * sigma'(z)
        } else if constexpr (EPI == EPI_PEJAC) {
            // the pending block holds the gradients of 16 encoding slots of this lane half (s = 16 PB + e): trig channels
            // contract with 2^f cos(2 pi (2^f x / 2 pi + h / 4))  (h = 0: d sin = cos, h = 1: d cos = -sin), identity channels add
            constexpr int s = 16 * PB + e, NT = 3 * L;
            if constexpr (DBGE && s0 == ks) { if (x.dbg) x.dbg[s] += a[e] * (1.f / GRAD_SCALE); }
            if constexpr (s < NT) {
                if constexpr (s0 == ks) ta[e] = __builtin_amdgcn_cosf(x.rev[s % 3] * (float)(1 << (s / 3)) + 0.25f * (float)h);
                if constexpr (s0 + 1 == ks) x.g[s % 3] = __builtin_fmaf(a[e] * (float)(1 << (s / 3)), ta[e], x.g[s % 3]);
            } else if constexpr (LO) {          // sdf net: slots 24..26 = x (h = 0) / its rounding residual (h = 1: same weight, counted once)
                if constexpr (s < NT + 3 && s0 == ks) x.g[s - NT] += h ? 0.f : a[e];
            } else {                            // residual net: slot 30 = x0 (h = 0) / x1 (h = 1), slot 31 = x2 (h = 0)
                if constexpr (s == NT && s0 == ks) { x.g[0] += h ? 0.f : a[e]; x.g[1] += h ? a[e] : 0.f; }
                if constexpr (s == NT + 1 && s0 == ks) x.g[2] += h ? 0.f : a[e];
            }
        } else {
            if constexpr (s0 == ks) {
                if constexpr (EPI == EPI_RELU || EPI == EPI_RELU_BITS) ta[e] = max0(a[e]);
                else if constexpr (EPI == EPI_LINEAR) ta[e] = a[e] * x.scale;
                else if constexpr (EPI == EPI_GRAD_RELU) ta[e] = __builtin_bit_cast(float, fbits(a[e]) & ~(unsigned)__builtin_amdgcn_sbfe((int)x.bits, 15 - e, 1));
                if constexpr (EPI == EPI_RELU_BITS) x.bits = __builtin_amdgcn_alignbit(x.bits, fbits(a[e]), 31);      // (bits << 1) | sign
            }
        }
        if constexpr (EPI != EPI_PEJAC && EPI != EPI_NONE) {
            constexpr int sdone = s0 + DEPTH;
            if constexpr ((e & 1) && sdone == ks) {
                const unsigned w = pack2<E>(ta[e - 1], ta[e]);
                if constexpr (e < 8) o0[e >> 1] = w; else o1[(e >> 1) & 3] = w;
            }
        }
    }
};

// One row block: KS MFMAs into `acc` (bias-initialised or zero), interleaved with the pending epilogue EPI of `accPrev`
// into the B fragments o0, o1 (or into aux.g for EPI_PEJAC).  After the block, the finished fragments go to the tape if aux.st.
// accumulators start at the bias of their rows; bias_l = table row + 4 * (lane half) ALREADY added and laundered through an empty
// asm by the caller: with the lane part folded into ~150 distinct loop-invariant LDS addresses the compiler hoists them all
// out of the tile loop and spills them (90 spilled registers in the 8-wave forward kernel)
__device__ __forceinline__ void init_acc_l(f32x16& acc, const float* bias_l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * q + i] = bv[i];
    }
}

struct NoTape { template <int K> __device__ __forceinline__ void take(u32x4&, u32x4&) {} };

// TK >= 0: the tape fragments of the NEXT pending block are taken from the queue (slot TK) in the MFMA slot right before
// this block's stage turnover (P.fetch<0>: counted wait + barrier + the next weight DMA): the compiler's wait for those loads
// cannot tell them from the weight DMA it does not see, so it drains the queue — cheap only where the youngest DMA is a whole
// block old, i.e. exactly there.
// PIN: four prefetched dwords (the ReLU net's sign-bit words of a later layer) are waited for in that same slot instead of at
// their first use, which would fall a few MFMAs AFTER a turnover — the worst place for a wait that drains the weight DMA.
template <typename E, int FM0, int KS, int EPI, bool EARLY, bool TAIL, bool BIAS, int L, bool LO, int PB = 0, int TK = -1, bool PIN = false, typename PipeT, typename TQ = NoTape>
__device__ __forceinline__ void rbg(PipeT& P, f32x16& acc, const f32x16& accPrev, u32x4 (&Bm)[16], const u32x4 (&Bp)[4], u32x4& o0, u32x4& o1,
                                    const float* bias_rb, int h, EpiAux& aux, TQ* tq = nullptr, unsigned* pin = nullptr) {
    if constexpr (BIAS) init_acc_l(acc, bias_rb);
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
    float ta[16], tb[16];
    static_for<0, KS>([&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        const u32x4 bw = (KS == 4) ? Bp[ks & 3] : (ks < 16 ? Bm[ks & 15] : Bp[ks & 3]);
        acc = Tr<E>::mfma(P.af[(FM0 + ks) % ST_PF], __builtin_bit_cast(X8<E>, bw), acc);
        if constexpr (TK >= 0 && (FM0 + ks + ST_PF) % 16 == 0) {
            tq->template take<TK>(aux.n0, aux.n1);
            asm volatile("" : "+v"(aux.n0), "+v"(aux.n1));          // the loads are waited for HERE
        }
        if constexpr (PIN && (FM0 + ks + ST_PF) % 16 == 0) asm volatile("" : "+v"(pin[0]), "+v"(pin[1]), "+v"(pin[2]), "+v"(pin[3]));
        if constexpr (!(TAIL && ks + ST_PF >= KS)) P.template fetch<(FM0 + ks + ST_PF) % 16>();
        if constexpr (EPI != EPI_NONE) {
            static_for<0, 16>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                Epi<E, EPI, KS, EARLY, L, LO, PB>::template step<ks, e>(accPrev, ta, tb, o0, o1, aux, h);
            });
        }
    });
}

// the pending epilogue alone (no MFMAs to hide behind): end of a phase
template <typename E, int EPI, int L, bool LO, int PB = 0>
__device__ __forceinline__ void flush(const f32x16& accPrev, u32x4& o0, u32x4& o1, int h, EpiAux& aux) {
    float ta[16], tb[16];
    static_for<0, 4>([&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        static_for<0, 16>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
            Epi<E, EPI, 4, false, L, LO, PB>::template step<ks, e>(accPrev, ta, tb, o0, o1, aux, h);
        });
    });
}

__device__ __forceinline__ void tape_store2(EpiAux& aux, const u32x4& a, const u32x4& b) {
    if (aux.st) {
        __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(aux.st));          // written once, read once by another kernel:
        __builtin_nontemporal_store(b, reinterpret_cast<u32x4*>(aux.st + 1024));   // streaming stores, -13 % forward time
        aux.st += 2048;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
//  forward with tape
// ---------------------------------------------------------------------------------------------------------------------
// a 256-row forward layer (8 row blocks).  On entry accB holds the pending last row block of the previous layer (EPI_IN,
// destination Bm[14], Bm[15]).  TAPE: 0 none, 1 fragments (softplus net), 2 sign bits (ReLU net, one dword per two blocks)
template <typename E, int KS, int EPI, int EPI_IN, int TAPE, typename PipeT>
__device__ __forceinline__ void fwd_layer(PipeT& P, f32x16& accA, f32x16& accB, u32x4 (&Bm)[16], const u32x4 (&Bp)[4], u32x4 (&Bo)[16],
                                          const float* bias, int h, EpiAux& aux, unsigned*& bits_out) {
    auto done = [&](auto n_, u32x4& a, u32x4& b) {       // the block whose epilogue just finished: n-th of its layer
        constexpr int n = decltype(n_)::value;
        if constexpr (TAPE == 1) tape_store2(aux, a, b);
        if constexpr (TAPE == 2 && (n & 1)) { *bits_out = aux.bits; bits_out += 64; }
    };
    rbg<E, 0, KS, EPI_IN, true, false, true, 0, false>(P, accA, accB, Bm, Bp, Bm[14], Bm[15], bias, h, aux);
    if constexpr (EPI_IN != EPI_NONE) done(std::integral_constant<int, 7>{}, Bm[14], Bm[15]);
    rbg<E, (1 * KS) % 16, KS, EPI, false, false, true, 0, false>(P, accB, accA, Bm, Bp, Bo[0], Bo[1], bias + 32, h, aux);
    done(std::integral_constant<int, 0>{}, Bo[0], Bo[1]);
    rbg<E, (2 * KS) % 16, KS, EPI, false, false, true, 0, false>(P, accA, accB, Bm, Bp, Bo[2], Bo[3], bias + 64, h, aux);
    done(std::integral_constant<int, 1>{}, Bo[2], Bo[3]);
    rbg<E, (3 * KS) % 16, KS, EPI, false, false, true, 0, false>(P, accB, accA, Bm, Bp, Bo[4], Bo[5], bias + 96, h, aux);
    done(std::integral_constant<int, 2>{}, Bo[4], Bo[5]);
    rbg<E, (4 * KS) % 16, KS, EPI, false, false, true, 0, false>(P, accA, accB, Bm, Bp, Bo[6], Bo[7], bias + 128, h, aux);
    done(std::integral_constant<int, 3>{}, Bo[6], Bo[7]);
    rbg<E, (5 * KS) % 16, KS, EPI, false, false, true, 0, false>(P, accB, accA, Bm, Bp, Bo[8], Bo[9], bias + 160, h, aux);
    done(std::integral_constant<int, 4>{}, Bo[8], Bo[9]);
    rbg<E, (6 * KS) % 16, KS, EPI, false, false, true, 0, false>(P, accA, accB, Bm, Bp, Bo[10], Bo[11], bias + 192, h, aux);
    done(std::integral_constant<int, 5>{}, Bo[10], Bo[11]);
    rbg<E, (7 * KS) % 16, KS, EPI, false, false, true, 0, false>(P, accB, accA, Bm, Bp, Bo[12], Bo[13], bias + 224, h, aux);
    done(std::integral_constant<int, 6>{}, Bo[12], Bo[13]);
}

// encoding B fragments of one point (lane half h): see pe_chan_resd / pe_chan_sdf in ra_pack.cpp (same as K3's)
template <typename E, int L, bool LO>
__device__ __forceinline__ void pe_frags_g(u32x4 (&Bp)[4], const float (&x)[3], int h) {
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

template <typename E> struct FwSmem {
    E ring[ST_RING * ST_STAGE_BYTES / 2];
    float bias[FW_BIAS_ROWS * 256];
    int count;
};

template <typename PipeT, typename SM>
__device__ __forceinline__ void pipe_init(PipeT& P, const void* stream, SM& sm, int wave, int lane, int nw, int stages) {
    P.g = reinterpret_cast<const char*>(stream);
    P.voff = wave * (16 / nw) * 1024 + lane * 16;
    P.ring = reinterpret_cast<const char*>(sm.ring) + lane * 16;
    P.ring_addr = (unsigned)(size_t)sm.ring + wave * (16 / nw) * 1024;
    P.slot = ST_RING - 1;            // the first sync_stage() advances to slot 0 / stream stage 0
    P.sstage = stages - 1;
    P.rd = P.ring;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < ST_AHEAD; ++st) P.issue(st, st);
}

template <typename E, int NW, bool DBG>
__global__ __launch_bounds__(64 * NW, 2) void mlp_fwd_tape_kernel(GeoNet net, const void* __restrict__ stream, const float* __restrict__ ba, FrameState fr,
                                                                    FullIO io, char* __restrict__ tape) {
    __shared__ __attribute__((aligned(16))) FwSmem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int TM = 32 * NW;
    for (int i = tid; i < FW_BIAS_ROWS * 256; i += 64 * NW) {
        const int row = i >> 8, r = i & 255;
        float v = 0.f;
        if (row < 8) v = row == 0 ? fr.bias_r0[r] : (row == 4 ? fr.bias_r4[r] : ba[net.r[row].bias + r]);
        else if (row == 8) v = r < 32 ? ba[net.rhead.bias + r] : 0.f;
        else if (row < 17) v = ba[net.s[row - 9].bias + r] * SP_SCALE;
        else if (row == 17) v = r < 32 ? ba[net.shead.bias + r] * SP_SCALE : 0.f;
        else v = ba[net.sfeat.bias + r] * SP_SCALE;
        sm.bias[i] = v;
    }
    if (tid == 0) sm.count = batch_count(io);
    __syncthreads();
    const int count = sm.count;             // slots of THIS sub-batch (slot0 .. slot0 + count)
    if (blockIdx.x == 0 && tid == 0 && io.counters) atomicAdd(&io.counters->n_fine_full, (unsigned long long)count);
    const int ntiles = (count + TM - 1) / TM;
    if ((int)blockIdx.x >= ntiles) return;

    Pipe<E, NW, FW_STAGES> P;
    pipe_init(P, stream, sm, wave, lane, NW, FW_STAGES);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int sl = tile * TM + wave * 32 + c;       // slot inside the sub-batch (the tape's index)
        const bool live = sl < count;
        const int s = io.slot0 + sl;                     // slot of the compacted fine list
        float x[3] = {0.f, 0.f, 0.f};
        if (live) { x[0] = io.bpts[3 * s]; x[1] = io.bpts[3 * s + 1]; x[2] = io.bpts[3 * s + 2]; }
        char* tw = tape + ((size_t)tile * NW + wave) * TP_WAVE;
        EpiAux aux;
        aux.bits = 0u; aux.scale = SP_INV; aux.st = nullptr; aux.dbg = nullptr;
        unsigned* bits_out = reinterpret_cast<unsigned*>(tw + TP_BITS) + lane;
        P.template fetch<0>(); P.template fetch<1>(); P.template fetch<2>(); P.template fetch<3>();
        u32x4 B0[16], B1[16], Bp[4];
        f32x16 accA, accB;
        // ---- residual deformation net (ReLU, sign bits to the tape)
        pe_frags_g<E, 10, false>(Bp, x, h);
        // the lane part goes through an empty asm as an INTEGER: the table rows then are one base register + immediates (a
        // laundered pointer loses its LDS address space: flat loads, each followed by s_waitcnt vmcnt(0) lgkmcnt(0) — which
        // drains the weight DMA and the tape stores once per row block)
        unsigned boff = 4 * h;
        asm volatile("" : "+v"(boff));
        const float* bias = sm.bias + boff;
        fwd_layer<E, 4, EPI_RELU_BITS, EPI_NONE, 2>(P, accA, accB, B0, Bp, B0, bias, h, aux, bits_out);
        fwd_layer<E, 16, EPI_RELU_BITS, EPI_RELU_BITS, 2>(P, accA, accB, B0, Bp, B1, bias + 256, h, aux, bits_out);
        fwd_layer<E, 16, EPI_RELU_BITS, EPI_RELU_BITS, 2>(P, accA, accB, B1, Bp, B0, bias + 512, h, aux, bits_out);
        fwd_layer<E, 16, EPI_RELU_BITS, EPI_RELU_BITS, 2>(P, accA, accB, B0, Bp, B1, bias + 768, h, aux, bits_out);
        fwd_layer<E, 20, EPI_RELU_BITS, EPI_RELU_BITS, 2>(P, accA, accB, B1, Bp, B0, bias + 1024, h, aux, bits_out);
        fwd_layer<E, 16, EPI_RELU_BITS, EPI_RELU_BITS, 2>(P, accA, accB, B0, Bp, B1, bias + 1280, h, aux, bits_out);
        fwd_layer<E, 16, EPI_RELU_BITS, EPI_RELU_BITS, 2>(P, accA, accB, B1, Bp, B0, bias + 1536, h, aux, bits_out);
        fwd_layer<E, 16, EPI_RELU_BITS, EPI_RELU_BITS, 2>(P, accA, accB, B0, Bp, B1, bias + 1792, h, aux, bits_out);
        rbg<E, 0, 16, EPI_RELU_BITS, true, false, true, 0, false>(P, accA, accB, B1, Bp, B1[14], B1[15], bias + 2048, h, aux);
        *bits_out = aux.bits;                                     // layer 7, blocks 6 and 7
        float cp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float r = tanhf(accA[k]) * io.resd_limit;           // valid in lanes h = 0 (rows 0..2)
            cp[k] = x[k] + __shfl(r, c);
        }
        // ---- signed distance net (softplus, scaled domain, activations to the tape), head, feature rows
        aux.st = tw + TP_SDF + lane * 16;
        asm volatile("" : "+v"(boff));
        bias = sm.bias + 9 * 256 + boff;
        pe_frags_g<E, 8, true>(Bp, cp, h);
        fwd_layer<E, 4, EPI_SOFTPLUS, EPI_NONE, 1>(P, accA, accB, B0, Bp, B0, bias, h, aux, bits_out);
        fwd_layer<E, 16, EPI_SOFTPLUS, EPI_SOFTPLUS, 1>(P, accA, accB, B0, Bp, B1, bias + 256, h, aux, bits_out);
        fwd_layer<E, 16, EPI_SOFTPLUS, EPI_SOFTPLUS, 1>(P, accA, accB, B1, Bp, B0, bias + 512, h, aux, bits_out);
        fwd_layer<E, 16, EPI_SOFTPLUS, EPI_SOFTPLUS, 1>(P, accA, accB, B0, Bp, B1, bias + 768, h, aux, bits_out);
        fwd_layer<E, 20, EPI_SOFTPLUS, EPI_SOFTPLUS, 1>(P, accA, accB, B1, Bp, B0, bias + 1024, h, aux, bits_out);
        fwd_layer<E, 16, EPI_SOFTPLUS, EPI_SOFTPLUS, 1>(P, accA, accB, B0, Bp, B1, bias + 1280, h, aux, bits_out);
        fwd_layer<E, 16, EPI_SOFTPLUS, EPI_SOFTPLUS, 1>(P, accA, accB, B1, Bp, B0, bias + 1536, h, aux, bits_out);
        fwd_layer<E, 16, EPI_SOFTPLUS, EPI_SOFTPLUS, 1>(P, accA, accB, B0, Bp, B1, bias + 1792, h, aux, bits_out);
        rbg<E, 0, 16, EPI_SOFTPLUS, true, false, true, 0, false>(P, accA, accB, B1, Bp, B1[14], B1[15], bias + 2048, h, aux);
        tape_store2(aux, B1[14], B1[15]);
        const float sdf = accA[0] * SP_INV;                        // head row 0, lanes h = 0
        // feature rows: lin8 rows 1..256 on the same inputs, no activation; packed as the heads' B fragments
        aux.st = tw + TP_FEAT + lane * 16;
        {
            const float* fb = bias + 2304;
            rbg<E, 0, 16, EPI_NONE, false, false, true, 0, false>(P, accB, accA, B1, Bp, B0[0], B0[1], fb, h, aux);
            rbg<E, 0, 16, EPI_LINEAR, false, false, true, 0, false>(P, accA, accB, B1, Bp, B0[0], B0[1], fb + 32, h, aux);
            tape_store2(aux, B0[0], B0[1]);
            rbg<E, 0, 16, EPI_LINEAR, false, false, true, 0, false>(P, accB, accA, B1, Bp, B0[2], B0[3], fb + 64, h, aux);
            tape_store2(aux, B0[2], B0[3]);
            rbg<E, 0, 16, EPI_LINEAR, false, false, true, 0, false>(P, accA, accB, B1, Bp, B0[4], B0[5], fb + 96, h, aux);
            tape_store2(aux, B0[4], B0[5]);
            rbg<E, 0, 16, EPI_LINEAR, false, false, true, 0, false>(P, accB, accA, B1, Bp, B0[6], B0[7], fb + 128, h, aux);
            tape_store2(aux, B0[6], B0[7]);
            rbg<E, 0, 16, EPI_LINEAR, false, false, true, 0, false>(P, accA, accB, B1, Bp, B0[8], B0[9], fb + 160, h, aux);
            tape_store2(aux, B0[8], B0[9]);
            rbg<E, 0, 16, EPI_LINEAR, false, false, true, 0, false>(P, accB, accA, B1, Bp, B0[10], B0[11], fb + 192, h, aux);
            tape_store2(aux, B0[10], B0[11]);
            rbg<E, 0, 16, EPI_LINEAR, false, true, true, 0, false>(P, accA, accB, B1, Bp, B0[12], B0[13], fb + 224, h, aux);
            tape_store2(aux, B0[12], B0[13]);
            flush<E, EPI_LINEAR, 0, false>(accA, B0[14], B0[15], h, aux);
            tape_store2(aux, B0[14], B0[15]);
        }
        if (DBG && io.dbg_feat && live) {          // test hook: features as the heads will see them (f16 values)
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = 32 * (k >> 1) + 16 * (k & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
                    io.dbg_feat[(size_t)s * 256 + f] = half_of<E>(B0[k][j >> 1], j & 1);
                }
        }
        if (h == 0) {
            float4 gv = make_float4(cp[0], cp[1], cp[2], sdf);
            *reinterpret_cast<float4*>(tw + TP_GEO + c * 16) = gv;
            if (DBG && io.dbg_sdf && live) io.dbg_sdf[s] = sdf;
            if (DBG && io.dbg_resd && live) { io.dbg_resd[3 * s] = cp[0] - x[0]; io.dbg_resd[3 * s + 1] = cp[1] - x[1]; io.dbg_resd[3 * s + 2] = cp[2] - x[2]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}


// ---------------------------------------------------------------------------------------------------------------------
//  backward + heads
// ---------------------------------------------------------------------------------------------------------------------
// tape reads of the softplus net, two row blocks ahead of their use: slot k & 1 holds the fragments of tape block k.
// Blocks are consumed in the order (layer 6, rb 0..7), (5, 0..7), ..., (0, 0..7); `next` counts requested blocks.
// Plain loads on purpose: an inline-asm load's destination is, for the compiler, defined at once — it may copy it (AGPR
// spill slots, register renaming at joins) or reuse the register of a request that is never consumed before the bytes
// arrive, and the late write-back then lands in whatever lives there (seen: garbage in the next block's accumulators).
// The compiler's own s_waitcnt for these loads does not know about the LDS-DMA weight stream issued in between, so it waits
// for more than it needs (the DMA of the last block stays in flight, older stages must have landed): safe, slightly early.
template <int FPW>
struct TapeQ {
    u32x4 q[2][2];
    const char* base;       // wave-tile tape + lane * 16
    int next;               // next block to request, 0 .. 55
    __device__ __forceinline__ int off(int k) const { const int l = 6 - (k >> 3), rb = k & 7; return TP_SDF + ((l * 8 + rb) * 2) * 1024; }
    template <int S>
    __device__ __forceinline__ void issue() {
        const char* p = base + off(next < 56 ? next : 55);
        q[S][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        q[S][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 1024));
        ++next;
    }
    template <int K01>
    __device__ __forceinline__ void take(u32x4& d0, u32x4& d1) {
        d0 = q[K01][0]; d1 = q[K01][1];
        issue<K01>();
    }
};

// a transposed 256-row layer (8 row blocks, K = 16 k-steps): Bo = EPI(W^T Bm).  On entry accB holds a pending block:
//   IN = 0 nothing, 1 the previous transposed layer's last block (same EPI, goes to Bm[14], Bm[15]), 2 an encoding-gradient
//   block (EPI_PEJAC, block PB_IN).  K01 is the parity of the first tape block this layer consumes (EPI_GRAD_SP).
template <typename E, int EPI, int IN, int L, bool LO, bool PF = false, typename PipeT, typename TQ>
__device__ __forceinline__ void bwd_layer(PipeT& P, f32x16& accA, f32x16& accB, u32x4 (&Bm)[16], const u32x4 (&Bp)[4], u32x4 (&Bo)[16], int h, EpiAux& aux,
                                          TQ& tq, const unsigned (&bw_in)[4], const unsigned (&bw)[4], unsigned* bw_next = nullptr,
                                          const unsigned* bt_next = nullptr) {
    // PF: the sign-bit words of the layer after next (bt_next, per lane) are requested now and pinned in block 7
    if constexpr (PF) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bw_next[k] = bt_next[k * 64];
    }
    // bw_in: sign-bit words of the pending block's layer (EPI_GRAD_RELU, IN == 1); bw: of this layer's output features.
    // EPI_GRAD_SP: on entry aux.t0/t1 hold the tape of the pending block (IN == 1); every row block i takes the tape of block
    // i of THIS layer's output (its epilogue is pending during block i + 1) — on exit aux.t0/t1 = block 7's.
    constexpr bool SP = EPI == EPI_GRAD_SP;
    auto bits = [&](auto n_, const unsigned (&words)[4]) {
        constexpr int n = decltype(n_)::value;
        if constexpr (EPI == EPI_GRAD_RELU) aux.bits = (n & 1) ? words[n >> 1] : (words[n >> 1] >> 16);
    };
    auto adv = [&]() { if constexpr (SP) { aux.t0 = aux.n0; aux.t1 = aux.n1; } };
    u32x4 dum0, dum1;
    if constexpr (IN == 1) {
        bits(std::integral_constant<int, 7>{}, bw_in);
        rbg<E, 0, 16, EPI, true, false, false, L, LO, 0, SP ? 0 : -1>(P, accA, accB, Bm, Bp, Bm[14], Bm[15], nullptr, h, aux, &tq);
    } else {
        rbg<E, 0, 16, EPI_NONE, false, false, false, L, LO, 0, SP ? 0 : -1>(P, accA, accB, Bm, Bp, dum0, dum1, nullptr, h, aux, &tq);
    }
    adv(); bits(std::integral_constant<int, 0>{}, bw);
    rbg<E, 0, 16, EPI, false, false, false, L, LO, 0, SP ? 1 : -1>(P, accB, accA, Bm, Bp, Bo[0], Bo[1], nullptr, h, aux, &tq);
    adv(); bits(std::integral_constant<int, 1>{}, bw);
    rbg<E, 0, 16, EPI, false, false, false, L, LO, 0, SP ? 0 : -1>(P, accA, accB, Bm, Bp, Bo[2], Bo[3], nullptr, h, aux, &tq);
    adv(); bits(std::integral_constant<int, 2>{}, bw);
    rbg<E, 0, 16, EPI, false, false, false, L, LO, 0, SP ? 1 : -1>(P, accB, accA, Bm, Bp, Bo[4], Bo[5], nullptr, h, aux, &tq);
    adv(); bits(std::integral_constant<int, 3>{}, bw);
    rbg<E, 0, 16, EPI, false, false, false, L, LO, 0, SP ? 0 : -1>(P, accA, accB, Bm, Bp, Bo[6], Bo[7], nullptr, h, aux, &tq);
    adv(); bits(std::integral_constant<int, 4>{}, bw);
    rbg<E, 0, 16, EPI, false, false, false, L, LO, 0, SP ? 1 : -1>(P, accB, accA, Bm, Bp, Bo[8], Bo[9], nullptr, h, aux, &tq);
    adv(); bits(std::integral_constant<int, 5>{}, bw);
    rbg<E, 0, 16, EPI, false, false, false, L, LO, 0, SP ? 0 : -1>(P, accA, accB, Bm, Bp, Bo[10], Bo[11], nullptr, h, aux, &tq);
    adv(); bits(std::integral_constant<int, 6>{}, bw);
    rbg<E, 0, 16, EPI, false, false, false, L, LO, 0, SP ? 1 : -1, PF>(P, accB, accA, Bm, Bp, Bo[12], Bo[13], nullptr, h, aux, &tq, bw_next);
    adv();
}

// the two encoding-gradient row blocks that follow a transposed layer fed by the encoding (W_pe^T delta: 64 rows):
// on entry accB = the layer's pending block 7 (EPI, into o14, o15; EARLY when those are this block's own inputs).  The two
// blocks' accumulators are contracted with the encoding's Jacobian right away (8 such blocks per tile: not worth a pending
// epilogue); on exit nothing is pending.  Bin: the delta fragments to multiply.
template <typename E, int EPI, bool EARLY, int L, bool LO, typename PipeT, typename TQ>
__device__ __forceinline__ void bwd_pe_blocks(PipeT& P, f32x16& accA, f32x16& accB, u32x4 (&Bin)[16], const u32x4 (&Bp)[4], u32x4& o14, u32x4& o15, int h,
                                              EpiAux& aux, TQ& tq, const unsigned (&bw)[4]) {
    if constexpr (EPI == EPI_GRAD_RELU) aux.bits = bw[3];        // EPI_GRAD_SP: aux.t0/t1 already hold block 7's tape
    flush<E, EPI, L, LO>(accB, o14, o15, h, aux);              // the layer's last row block, stand-alone
    u32x4 d0, d1;
    rbg<E, 0, 16, EPI_NONE, false, false, false, L, LO>(P, accA, accB, Bin, Bp, d0, d1, nullptr, h, aux);
    rbg<E, 0, 16, EPI_NONE, false, false, false, L, LO>(P, accB, accA, Bin, Bp, d0, d1, nullptr, h, aux);
    flush<E, EPI_PEJAC, L, LO, 0>(accA, d0, d1, h, aux);
    flush<E, EPI_PEJAC, L, LO, 1>(accB, d0, d1, h, aux);
}

__device__ __forceinline__ void inv3r(const float R[9], float M[9]) {   // blend_utils.py:125-165
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
__device__ __forceinline__ void normalize3r(float v[3]) {   // net_utils.py:1626-1628
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-8f;
    v[0] /= n; v[1] /= n; v[2] /= n;
}
__device__ __forceinline__ float sdf_to_occ_r(float sdf, float beta) {   // net_utils.py:852-893
    const float x = -sdf;
    float sigma;
    if (x <= 0.f) sigma = 1.f / beta * (0.5f * expf(x / beta));
    else sigma = 1.f / beta * (1.f - 0.5f * expf(-x / beta));
    return 1.f - expf(-fmaxf(sigma, 0.f) * 0.005f);
}

constexpr int BW_BIAS_ROWS = 6;       // heads: up to 4 full rows + head row; row 5: lin8 row 0 (fp32, the backward seed)
template <typename E> struct BwSmem {
    E ring[ST_RING * ST_STAGE_BYTES / 2];
    float bias[BW_BIAS_ROWS * 256];
    int count;
};

template <typename E, int NW, int STAGES, bool RELIGHT, bool DBG>
__global__ __launch_bounds__(64 * NW, 2) void mlp_bwd_heads_kernel(MatNet mat, ColNet col, const void* __restrict__ stream, const float* __restrict__ ba,
                                                                     const float* __restrict__ shead_row, FrameState fr, FullIO io, const char* __restrict__ tape) {
    __shared__ __attribute__((aligned(16))) BwSmem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int TM = 32 * NW;
    constexpr int FPW = 16 / NW;
    for (int i = tid; i < BW_BIAS_ROWS * 256; i += 64 * NW) {
        const int row = i >> 8, r = i & 255;
        float v = 0.f;
        if (row == 5) v = shead_row[r];
        else if (RELIGHT) {
            if (row == 0) v = ba[mat.m0.bias + r] * SP_SCALE;
            else if (row == 1) v = ba[mat.m1.bias + r] * SP_SCALE;
            else if (row == 2) v = r < 32 ? ba[mat.mhead.bias + r] * SP_SCALE : 0.f;
        } else {
            if (row == 0) v = ba[col.c0a.bias + r];
            else if (row == 1) v = ba[col.c1.bias + r];
            else if (row == 2) v = ba[col.c2.bias + r];
            else if (row == 3) v = fr.bias_c3[r];
            else if (row == 4) v = r < 32 ? ba[col.chead.bias + r] : 0.f;
        }
        sm.bias[i] = v;
    }
    if (tid == 0) sm.count = batch_count(io);
    __syncthreads();
    const int count = sm.count;
    const int ntiles = (count + TM - 1) / TM;
    if ((int)blockIdx.x >= ntiles) return;

    Pipe<E, NW, STAGES> P;
    pipe_init(P, stream, sm, wave, lane, NW, STAGES);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int sl = tile * TM + wave * 32 + c;
        const bool live = sl < count;
        const int s = io.slot0 + sl;
        const char* tw = tape + ((size_t)tile * NW + wave) * TP_WAVE;
        float x[3] = {0.f, 0.f, 0.f};
        if (live) { x[0] = io.bpts[3 * s]; x[1] = io.bpts[3 * s + 1]; x[2] = io.bpts[3 * s + 2]; }
        const float4 geo = *reinterpret_cast<const float4*>(tw + TP_GEO + c * 16);
        const float cp[3] = {geo.x, geo.y, geo.z};
        const float sdfv = geo.w;
        u32x4 B0[16], B1[16], Bp[4];
        f32x16 accA, accB;
        EpiAux aux;
        aux.st = nullptr; aux.bits = 0u; aux.scale = 1.f;
        float* const dbg_base = (DBG && io.dbg_pe && io.dbg_layer < 0 && live) ? io.dbg_pe + (size_t)s * 128 + h * 32 : nullptr;      // [sdf: 2 x 32 | resd: 2 x 32]
        aux.dbg = (DBG && io.dbg_layer == -1) ? dbg_base : nullptr;
        const unsigned nob[4] = {0u, 0u, 0u, 0u};
        // ---- seed of the sdf net: delta_7 = sigma'(z_7) * lin8[0, :] (times GRAD_SCALE), straight into B fragments
        {
            unsigned toff = lane * 16, woff = 4 * h;             // laundered as integers: the pointers keep their address spaces
            asm volatile("" : "+v"(woff), "+v"(toff));
            const char* t7 = tw + TP_SDF + (7 * 8 * 2) * 1024 + toff;
            const float* w8 = sm.bias + 5 * 256 + woff;
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(t7 + (rb * 2) * 1024);
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(t7 + (rb * 2 + 1) * 1024);
                float d[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float y = half_of<E>(e < 8 ? a0[e >> 1] : a1[(e >> 1) & 3], e & 1);
                    const int row = 32 * rb + 8 * (e >> 2) + (e & 3);                 // + 4 h inside w8
                    d[e] = (1.f - __builtin_amdgcn_exp2f(-y)) * (w8[row] * GRAD_SCALE);
                }
#pragma unroll
                for (int w = 0; w < 4; ++w) { B0[2 * rb][w] = pack2<E>(d[2 * w], d[2 * w + 1]); B0[2 * rb + 1][w] = pack2<E>(d[8 + 2 * w], d[8 + 2 * w + 1]); }
            }
        }
        auto dumpB = [&](const u32x4 (&Bf)[16], int which) {          // debugging aid: delta fragments as fp32, feature-major
            if (DBG && io.dbg_pe && io.dbg_layer == which && live) {
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int f = 32 * (k >> 1) + 16 * (k & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
                        io.dbg_pe[(size_t)s * 256 + f] = half_of<E>(Bf[k][j >> 1], j & 1) * (1.f / GRAD_SCALE);
                    }
            }
        };
        dumpB(B0, 7);
        TapeQ<FPW> tq;
        tq.base = tw + lane * 16; tq.next = 0;
        tq.template issue<0>(); tq.template issue<1>();
        P.template fetch<0>(); P.template fetch<1>(); P.template fetch<2>(); P.template fetch<3>();
        // ---- sdf net backward: gradient wrt the encoding's input cpts accumulates in aux.g
#pragma unroll
        for (int k = 0; k < 3; ++k) { aux.g[k] = 0.f; aux.rev[k] = cp[k] * INV_2PI; }
        bwd_layer<E, EPI_GRAD_SP, 0, 8, true>(P, accA, accB, B0, Bp, B1, h, aux, tq, nob, nob);       // lin7^T -> delta_6
        dumpB(B1, 6);
        bwd_layer<E, EPI_GRAD_SP, 1, 8, true>(P, accA, accB, B1, Bp, B0, h, aux, tq, nob, nob);       // lin6^T -> delta_5
        dumpB(B0, 5);
        bwd_layer<E, EPI_GRAD_SP, 1, 8, true>(P, accA, accB, B0, Bp, B1, h, aux, tq, nob, nob);       // lin5^T -> delta_4
        dumpB(B1, 4);
        bwd_layer<E, EPI_GRAD_SP, 1, 8, true>(P, accA, accB, B1, Bp, B0, h, aux, tq, nob, nob);       // lin4^T (205 rows) -> delta_3
        if (DBG && io.dbg_layer == -2) aux.dbg = dbg_base;
        bwd_pe_blocks<E, EPI_GRAD_SP, false, 8, true>(P, accA, accB, B1, Bp, B0[14], B0[15], h, aux, tq, nob);   // lin4's encoding columns
        if (DBG && io.dbg_layer == -2) aux.dbg = nullptr;
        if (DBG && io.dbg_layer == -3) aux.dbg = dbg_base;
        dumpB(B0, 3);
        bwd_layer<E, EPI_GRAD_SP, 0, 8, true>(P, accA, accB, B0, Bp, B1, h, aux, tq, nob, nob);       // lin3^T -> delta_2
        dumpB(B1, 2);
        bwd_layer<E, EPI_GRAD_SP, 1, 8, true>(P, accA, accB, B1, Bp, B0, h, aux, tq, nob, nob);       // lin2^T -> delta_1
        dumpB(B0, 1);
        bwd_layer<E, EPI_GRAD_SP, 1, 8, true>(P, accA, accB, B0, Bp, B1, h, aux, tq, nob, nob);       // lin1^T -> delta_0
        dumpB(B1, 0);
        bwd_pe_blocks<E, EPI_GRAD_SP, true, 8, true>(P, accA, accB, B1, Bp, B1[14], B1[15], h, aux, tq, nob);    // lin0^T
        float gc[3];        // GRAD_SCALE * d sdf / d cpts
#pragma unroll
        for (int k = 0; k < 3; ++k) gc[k] = aux.g[k] + __shfl_xor(aux.g[k], 32);
        if (DBG && io.dbg_gc && live && h == 0) { io.dbg_gc[3 * s] = gc[0] / GRAD_SCALE; io.dbg_gc[3 * s + 1] = gc[1] / GRAD_SCALE; io.dbg_gc[3 * s + 2] = gc[2] / GRAD_SCALE; }
        if (DBG && aux.dbg) aux.dbg += 64;
        // ---- residual net backward, seeded with d resd_k / d z_k * g_k = resd_limit (1 - tanh^2 z_k) g_k
        {
            float sd[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float th = (cp[k] - x[k]) / io.resd_limit;
                sd[k] = io.resd_limit * (1.f - th * th) * gc[k];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int w = 0; w < 4; ++w) Bp[ks][w] = 0u;
            Bp[0][0] = h ? 0u : pack2<E>(sd[0], sd[1]);
            Bp[0][1] = h ? 0u : pack2<E>(sd[2], 0.f);
        }
        const unsigned* bt = reinterpret_cast<const unsigned*>(tw + TP_BITS) + lane;
        unsigned bwA[4], bwB[4], bwC[4];            // three rotating sets: pending block's layer, this layer, prefetch
#pragma unroll
        for (int k = 0; k < 4; ++k) { bwA[k] = bt[(7 * 4 + k) * 64]; bwB[k] = bt[(6 * 4 + k) * 64]; }      // layers 7 and 6
        // head^T: 8 row blocks of 4 k-steps (3 real columns), epilogue = layer 7's ReLU mask -> delta_7
        {
            u32x4 d0, d1;
            rbg<E, 0, 4, EPI_NONE, false, false, false, 10, false>(P, accA, accB, B0, Bp, d0, d1, nullptr, h, aux);
            aux.bits = bwA[0] >> 16;
            rbg<E, 4, 4, EPI_GRAD_RELU, false, false, false, 10, false>(P, accB, accA, B0, Bp, B0[0], B0[1], nullptr, h, aux);
            aux.bits = bwA[0];
            rbg<E, 8, 4, EPI_GRAD_RELU, false, false, false, 10, false>(P, accA, accB, B0, Bp, B0[2], B0[3], nullptr, h, aux);
            aux.bits = bwA[1] >> 16;
            rbg<E, 12, 4, EPI_GRAD_RELU, false, false, false, 10, false>(P, accB, accA, B0, Bp, B0[4], B0[5], nullptr, h, aux);
            aux.bits = bwA[1];
            rbg<E, 0, 4, EPI_GRAD_RELU, false, false, false, 10, false>(P, accA, accB, B0, Bp, B0[6], B0[7], nullptr, h, aux);
            aux.bits = bwA[2] >> 16;
            rbg<E, 4, 4, EPI_GRAD_RELU, false, false, false, 10, false>(P, accB, accA, B0, Bp, B0[8], B0[9], nullptr, h, aux);
            aux.bits = bwA[2];
            rbg<E, 8, 4, EPI_GRAD_RELU, false, false, false, 10, false>(P, accA, accB, B0, Bp, B0[10], B0[11], nullptr, h, aux);
            aux.bits = bwA[3] >> 16;
            rbg<E, 12, 4, EPI_GRAD_RELU, false, false, false, 10, false, 0, -1, true>(P, accB, accA, B0, Bp, B0[12], B0[13], nullptr, h, aux, (NoTape*)nullptr, bwB);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { aux.g[k] = 0.f; aux.rev[k] = x[k] * INV_2PI; }
        auto words = [&](int l) { return bt + (l * 4) * 64; };
        bwd_layer<E, EPI_GRAD_RELU, 1, 10, false, true>(P, accA, accB, B0, Bp, B1, h, aux, tq, bwA, bwB, bwC, words(5));   // W7^T -> delta_6
        bwd_layer<E, EPI_GRAD_RELU, 1, 10, false, true>(P, accA, accB, B1, Bp, B0, h, aux, tq, bwB, bwC, bwA, words(4));   // W6^T -> delta_5
        bwd_layer<E, EPI_GRAD_RELU, 1, 10, false, true>(P, accA, accB, B0, Bp, B1, h, aux, tq, bwC, bwA, bwB, words(3));   // W5^T -> delta_4
        bwd_layer<E, EPI_GRAD_RELU, 1, 10, false, true>(P, accA, accB, B1, Bp, B0, h, aux, tq, bwA, bwB, bwC, words(2));   // W4^T (hidden columns) -> delta_3
        bwd_pe_blocks<E, EPI_GRAD_RELU, false, 10, false>(P, accA, accB, B1, Bp, B0[14], B0[15], h, aux, tq, bwB);
        bwd_layer<E, EPI_GRAD_RELU, 0, 10, false, true>(P, accA, accB, B0, Bp, B1, h, aux, tq, bwB, bwC, bwA, words(1));   // W3^T -> delta_2
        bwd_layer<E, EPI_GRAD_RELU, 1, 10, false, true>(P, accA, accB, B1, Bp, B0, h, aux, tq, bwC, bwA, bwB, words(0));   // W2^T -> delta_1
        bwd_layer<E, EPI_GRAD_RELU, 1, 10, false>(P, accA, accB, B0, Bp, B1, h, aux, tq, bwA, bwB);                        // W1^T -> delta_0
        bwd_pe_blocks<E, EPI_GRAD_RELU, true, 10, false>(P, accA, accB, B1, Bp, B1[14], B1[15], h, aux, tq, bwB);   // W0^T
        float g[3];         // d sdf / d bpts = g_c + J_resd^T g_c
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = (gc[k] + aux.g[k] + __shfl_xor(aux.g[k], 32)) * (1.f / GRAD_SCALE);
        if (DBG && io.dbg_grad && live && h == 0) { io.dbg_grad[3 * s] = g[0]; io.dbg_grad[3 * s + 1] = g[1]; io.dbg_grad[3 * s + 2] = g[2]; }
        // ---- per-point geometry outputs (both lane halves compute them: the colour net's encoding needs them in both)
        const float occ = sdf_to_occ_r(sdfv, io.beta);
        float nrm[3], bv[3] = {0.f, 0.f, 0.f};
        {
            float gn[3] = {g[0], g[1], g[2]};
            normalize3r(gn);
            float A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Bm9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (io.mats && live) {
                const float* M = io.mats + (size_t)s * 24;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) { A[3 * i + j] = M[4 * i + j]; Bm9[3 * i + j] = M[12 + 4 * i + j]; }
            }
            float Ai[9], Bi[9];
            inv3r(A, Ai);
            inv3r(Bm9, Bi);
            // normal: big-pose -> T (big_R^T), T -> pose (R_inv^T), pose -> world (R), base_network.py:471-475
            float nt_[3], np_[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) nt_[i] = Bm9[0 + i] * gn[0] + Bm9[3 + i] * gn[1] + Bm9[6 + i] * gn[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) np_[i] = Ai[0 + i] * nt_[0] + Ai[3 + i] * nt_[1] + Ai[6 + i] * nt_[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) nrm[i] = np_[0] * fr.R[3 * i] + np_[1] * fr.R[3 * i + 1] + np_[2] * fr.R[3 * i + 2];
            normalize3r(nrm);
            if (!RELIGHT && io.view && live) {   // view dirs to big-pose space, base_network.py:324-334 (not re-normalised)
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
        // ---- heads on the taped features
        {
            const char* tf = tw + TP_FEAT + lane * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) B0[k] = *reinterpret_cast<const u32x4*>(tf + k * 1024);
        }
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned* nobits = nullptr;
        aux.st = nullptr;
        unsigned hoff = 4 * h;
        asm volatile("" : "+v"(hoff));          // integer, not the pointer: see the forward kernel
        const float* hb = sm.bias + hoff;
        if constexpr (RELIGHT) {
            fwd_layer<E, 16, EPI_SOFTPLUS, EPI_NONE, 0>(P, accA, accB, B0, Bp, B1, hb, h, aux, nobits);
            fwd_layer<E, 16, EPI_SOFTPLUS, EPI_SOFTPLUS, 0>(P, accA, accB, B1, Bp, B0, hb + 256, h, aux, nobits);
            rbg<E, 0, 16, EPI_SOFTPLUS, true, true, true, 0, false>(P, accA, accB, B0, Bp, B0[14], B0[15], hb + 512, h, aux);
#pragma unroll
            for (int k = 0; k < 3; ++k) o4[k] = io.albedo_slope / (1.f + expf(-accA[k] * SP_INV)) + io.albedo_bias;
            o4[3] = io.rough_slope / (1.f + expf(-accA[3] * SP_INV)) + io.rough_bias;
        } else {
            {   // [PE4(bvds) | world normal] as encoding fragments (pe_chan_col in ra_pack.cpp)
                float rev[3], v[32];
#pragma unroll
                for (int k = 0; k < 3; ++k) rev[k] = bv[k] * INV_2PI;
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    if (q < 12) {
                        const float a = rev[q % 3] * (float)(1 << (q / 3));
                        v[q] = h ? __builtin_amdgcn_cosf(a) : __builtin_amdgcn_sinf(a);
                    } else if (q < 15) v[q] = h ? nrm[q - 12] : bv[q - 12];
                    else v[q] = 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int w = 0; w < 4; ++w) Bp[ks][w] = pack2<E>(v[8 * ks + 2 * w], v[8 * ks + 2 * w + 1]);
            }
            fwd_layer<E, 20, EPI_RELU, EPI_NONE, 0>(P, accA, accB, B0, Bp, B1, hb, h, aux, nobits);
            fwd_layer<E, 16, EPI_RELU, EPI_RELU, 0>(P, accA, accB, B1, Bp, B0, hb + 256, h, aux, nobits);
            fwd_layer<E, 16, EPI_RELU, EPI_RELU, 0>(P, accA, accB, B0, Bp, B1, hb + 512, h, aux, nobits);
            fwd_layer<E, 16, EPI_RELU, EPI_RELU, 0>(P, accA, accB, B1, Bp, B0, hb + 768, h, aux, nobits);
            rbg<E, 0, 16, EPI_RELU, true, true, true, 0, false>(P, accA, accB, B0, Bp, B0[14], B0[15], hb + 1024, h, aux);
#pragma unroll
            for (int k = 0; k < 3; ++k) o4[k] = 1.f / (1.f + expf(-accA[k]));
        }
        // ---- raw channels (lanes h = 0 hold head rows 0..3)
        if (h == 0 && live) {
            float* o = io.raw + (size_t)io.idx[s] * io.C;
#pragma unroll
            for (int k = 0; k < 3; ++k) { o[k] = cp[k]; o[3 + k] = x[k]; o[6 + k] = cp[k] - x[k]; }
            if (RELIGHT) {
                o[9] = o4[0]; o[10] = o4[1]; o[11] = o4[2]; o[12] = o4[3];
                o[13] = nrm[0]; o[14] = nrm[1]; o[15] = nrm[2]; o[16] = occ;
            } else {
                o[9] = nrm[0]; o[10] = nrm[1]; o[11] = nrm[2];
                o[12] = o4[0]; o[13] = o4[1]; o[14] = o4[2]; o[15] = occ;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace

// tape of a sub-batch of `slots` fine slots (whole 256-point tiles)
static size_t k4_tape_bytes(int slots) { return (size_t)((slots + 255) / 256) * 8 * TP_WAVE + 4096; }

template <typename E>
static void launch_k4_fwd(const GeoNet& net, const void* fwd_arena, const float* barena, const FrameState& fr, const FullIO& io, char* tape, hipStream_t stream) {
    if (io.slot_cap <= 0) return;
    constexpr int NW = RA_K4_NW_F;
    const int tiles = (io.slot_cap + 32 * NW - 1) / (32 * NW);
    const int grid = tiles < 256 ? tiles : 256;
    if (io.dbg_feat || io.dbg_sdf || io.dbg_resd) hipLaunchKernelGGL((mlp_fwd_tape_kernel<E, NW, true>), dim3(grid), dim3(64 * NW), 0, stream, net, fwd_arena, barena, fr, io, tape);
    else hipLaunchKernelGGL((mlp_fwd_tape_kernel<E, NW, false>), dim3(grid), dim3(64 * NW), 0, stream, net, fwd_arena, barena, fr, io, tape);
}

constexpr int BW_STAGES_RELIGHT = 139, BW_STAGES_ANISDF = 157;      // backward stream: geometry (transposed) + material heads / colour net

template <typename E>
static void launch_k4_bwd(const MatNet& mat, const ColNet& col, const void* bwd_arena, const float* barena, const float* shead_row, const FrameState& fr,
                          const FullIO& io, const char* tape, hipStream_t stream) {
    if (io.slot_cap <= 0) return;
    constexpr int NW = RA_K4_NW_B;
    const int tiles = (io.slot_cap + 32 * NW - 1) / (32 * NW);
    const int grid = tiles < 256 ? tiles : 256;
    const bool dbg = io.dbg_grad || io.dbg_gc || io.dbg_pe;
    if (io.relight) {
        if (dbg) hipLaunchKernelGGL((mlp_bwd_heads_kernel<E, NW, BW_STAGES_RELIGHT, true, true>), dim3(grid), dim3(64 * NW), 0, stream, mat, col, bwd_arena, barena, shead_row, fr, io, tape);
        else hipLaunchKernelGGL((mlp_bwd_heads_kernel<E, NW, BW_STAGES_RELIGHT, true, false>), dim3(grid), dim3(64 * NW), 0, stream, mat, col, bwd_arena, barena, shead_row, fr, io, tape);
    } else {
        if (dbg) hipLaunchKernelGGL((mlp_bwd_heads_kernel<E, NW, BW_STAGES_ANISDF, false, true>), dim3(grid), dim3(64 * NW), 0, stream, mat, col, bwd_arena, barena, shead_row, fr, io, tape);
        else hipLaunchKernelGGL((mlp_bwd_heads_kernel<E, NW, BW_STAGES_ANISDF, false, false>), dim3(grid), dim3(64 * NW), 0, stream, mat, col, bwd_arena, barena, shead_row, fr, io, tape);
    }
}
