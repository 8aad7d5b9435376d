// K3CC: the compensated distance query (K3C, ra_k3c.hpp) with FOUR WAVES COOPERATING on one 16-point tile — the latency variant for the
// small launches of a surface-trace loop (one rank of eight, small images: a few hundred to a few thousand points per iteration).
//
// Why: a K3C tile is one wave walking 5808 dependent-issue MFMAs: 77 us of MFMA issue + 25 us of exposed weight-stream latency = 102 us
// whatever the launch size (tools/bench_k3c.py; round 4's stream ablation), and a 2 400-point launch occupies 150 of the chip's 1024 SIMDs.  Sixteen
// iterations of that are the longest dependent chain of a rank's frame.  Here the 16 row blocks of a layer are dealt to the four SIMDs of
// a CU (wave w owns row blocks 4 i + w), so a layer is four row blocks deep instead of sixteen:
//   * weights: every wave walks a PRIVATE stream (ra_pack.cpp: its own row blocks in order, both heads in every stream), and because no
//     other wave needs its fragments they do not pass through LDS at all: plain `global_load_dwordx4` straight into a rotating file of
//     62 fragment registers (248 of the lone wave's 512 registers; loads may target AGPRs and MFMAs read A operands from them), each
//     refilled 31 k-steps ahead right after the MFMAs that consumed it.  62 KB in flight per wave instead of the 24 KB a quarter of the
//     LDS can hold (first version, private LDS-DMA rings: 63 us per tile, 36 us without the stream — the stream's latency, 1.2-1.4 us per
//     round trip, was all that mattered); no barrier in the weight path, no LDS operand reads; one counted `s_waitcnt vmcnt(60)` per
//     k-step (below: the registers are addressed by name from inline assembly).  The per-wave stream is 992 = 16 x 62 fragments long, so
//     a fragment's register is its position mod 62 in every tile.  56 us per launch; 37 us without the loads, 51 us from a cache-hot 4 KB:
//     what remains is the CU's vector-memory path (3.97 MB per tile at 64 B/clk = 27 us at best);
//   * activations: the D fragments of a wave's row blocks — after bias / activation / hi + lo split exactly as in K3C — go to a 16 KB LDS
//     array in the next layer's B-fragment layout (lane (c, g) of k-step ks holds features 32 ks + 16 (j >> 2) + 4 g + (j & 3): row block
//     2 ks + (j >> 2), elements j & 3 — 8 bytes per row block and lane), two barriers per layer (array free / array complete), every wave reads all eight k-steps back;
//   * both heads are computed by every wave (24 MFMAs each) instead of being broadcast.
// Every output is the same chain of operations on the same operands as in K3C: bit-identical distances (test_compensated_distance_query...).
// reference: as ra_k3c.hpp
#include "ra_k3c.hpp"

namespace {

constexpr int CC_W = 4;                          // cooperating waves = SIMDs of a CU
constexpr int CC_FRAGS = 992;                    // per wave and tile: 2 x (16 + 3 x 64 + 80 + 3 x 64 + 16)
constexpr int CC_PF = 62;                        // fragment registers = loads in flight (vmcnt counts to 63); divides CC_FRAGS

struct CcSmem {
    u32x4 act[8 * 2 * 64];                       // [k-step][hi | lo][lane]
    float bias[BIAS_ROWS * 256];
    int count;
};

// The fragment file: slot n = a[4 n : 4 n + 3], addressed BY NAME.  Left to the compiler (plain loads, or asm loads whose results are C++
// values) 62 fragments in flight end badly either way: it "spills" each load result to an AGPR through a VGPR, waiting for it on the spot,
// or moves an asm load's destination aside before the data has arrived.  So the loads have no C++ result at all; a k-step's two fragments
// become values at the counted wait that precedes their MFMAs (62 loads in flight, the two oldest are this k-step's), born in their
// physical registers.  The accumulators are kept out of the AGPR file (Makefile: -amdgpu-mfma-vgpr-form=1), so nothing else wants one;
// tools/check_k3cc_isa.py verifies on the compiled kernel that no instruction outside these statements writes an AGPR.
template <int N> struct CcSlot;
#define RA_CC_SLOT(n, r0, r1, r2, r3)                                                                                                   \
    template <> struct CcSlot<n> {                                                                                                      \
        static __device__ __forceinline__ void load(unsigned voff, const char* cur) {                                                   \
            asm volatile("global_load_dwordx4 a[" #r0 ":" #r3 "], %0, %1" :: "v"(voff), "s"(cur) : "a" #r0, "a" #r1, "a" #r2, "a" #r3); \
        }                                                                                                                               \
    };
RA_CC_SLOT(0, 0, 1, 2, 3) RA_CC_SLOT(1, 4, 5, 6, 7) RA_CC_SLOT(2, 8, 9, 10, 11) RA_CC_SLOT(3, 12, 13, 14, 15)
RA_CC_SLOT(4, 16, 17, 18, 19) RA_CC_SLOT(5, 20, 21, 22, 23) RA_CC_SLOT(6, 24, 25, 26, 27) RA_CC_SLOT(7, 28, 29, 30, 31)
RA_CC_SLOT(8, 32, 33, 34, 35) RA_CC_SLOT(9, 36, 37, 38, 39) RA_CC_SLOT(10, 40, 41, 42, 43) RA_CC_SLOT(11, 44, 45, 46, 47)
RA_CC_SLOT(12, 48, 49, 50, 51) RA_CC_SLOT(13, 52, 53, 54, 55) RA_CC_SLOT(14, 56, 57, 58, 59) RA_CC_SLOT(15, 60, 61, 62, 63)
RA_CC_SLOT(16, 64, 65, 66, 67) RA_CC_SLOT(17, 68, 69, 70, 71) RA_CC_SLOT(18, 72, 73, 74, 75) RA_CC_SLOT(19, 76, 77, 78, 79)
RA_CC_SLOT(20, 80, 81, 82, 83) RA_CC_SLOT(21, 84, 85, 86, 87) RA_CC_SLOT(22, 88, 89, 90, 91) RA_CC_SLOT(23, 92, 93, 94, 95)
RA_CC_SLOT(24, 96, 97, 98, 99) RA_CC_SLOT(25, 100, 101, 102, 103) RA_CC_SLOT(26, 104, 105, 106, 107) RA_CC_SLOT(27, 108, 109, 110, 111)
RA_CC_SLOT(28, 112, 113, 114, 115) RA_CC_SLOT(29, 116, 117, 118, 119) RA_CC_SLOT(30, 120, 121, 122, 123) RA_CC_SLOT(31, 124, 125, 126, 127)
RA_CC_SLOT(32, 128, 129, 130, 131) RA_CC_SLOT(33, 132, 133, 134, 135) RA_CC_SLOT(34, 136, 137, 138, 139) RA_CC_SLOT(35, 140, 141, 142, 143)
RA_CC_SLOT(36, 144, 145, 146, 147) RA_CC_SLOT(37, 148, 149, 150, 151) RA_CC_SLOT(38, 152, 153, 154, 155) RA_CC_SLOT(39, 156, 157, 158, 159)
RA_CC_SLOT(40, 160, 161, 162, 163) RA_CC_SLOT(41, 164, 165, 166, 167) RA_CC_SLOT(42, 168, 169, 170, 171) RA_CC_SLOT(43, 172, 173, 174, 175)
RA_CC_SLOT(44, 176, 177, 178, 179) RA_CC_SLOT(45, 180, 181, 182, 183) RA_CC_SLOT(46, 184, 185, 186, 187) RA_CC_SLOT(47, 188, 189, 190, 191)
RA_CC_SLOT(48, 192, 193, 194, 195) RA_CC_SLOT(49, 196, 197, 198, 199) RA_CC_SLOT(50, 200, 201, 202, 203) RA_CC_SLOT(51, 204, 205, 206, 207)
RA_CC_SLOT(52, 208, 209, 210, 211) RA_CC_SLOT(53, 212, 213, 214, 215) RA_CC_SLOT(54, 216, 217, 218, 219) RA_CC_SLOT(55, 220, 221, 222, 223)
RA_CC_SLOT(56, 224, 225, 226, 227) RA_CC_SLOT(57, 228, 229, 230, 231) RA_CC_SLOT(58, 232, 233, 234, 235) RA_CC_SLOT(59, 236, 237, 238, 239)
RA_CC_SLOT(60, 240, 241, 242, 243) RA_CC_SLOT(61, 244, 245, 246, 247)
#undef RA_CC_SLOT
template <int K> struct CcPair;          // the [hi | lo] fragments of a k-step: slots 2 K, 2 K + 1
#define RA_CC_PAIR(k, h0, h3, l0, l3)                                                                                                   \
    template <> struct CcPair<k> {                                                                                                      \
        static __device__ __forceinline__ void wait(f16x8& hi, f16x8& lo) {                                                             \
            asm volatile("s_waitcnt vmcnt(%2)" : "={a[" #h0 ":" #h3 "]}"(hi), "={a[" #l0 ":" #l3 "]}"(lo) : "n"(CC_PF - 2));           \
        }                                                                                                                               \
    };
RA_CC_PAIR(0, 0, 3, 4, 7) RA_CC_PAIR(1, 8, 11, 12, 15) RA_CC_PAIR(2, 16, 19, 20, 23) RA_CC_PAIR(3, 24, 27, 28, 31)
RA_CC_PAIR(4, 32, 35, 36, 39) RA_CC_PAIR(5, 40, 43, 44, 47) RA_CC_PAIR(6, 48, 51, 52, 55) RA_CC_PAIR(7, 56, 59, 60, 63)
RA_CC_PAIR(8, 64, 67, 68, 71) RA_CC_PAIR(9, 72, 75, 76, 79) RA_CC_PAIR(10, 80, 83, 84, 87) RA_CC_PAIR(11, 88, 91, 92, 95)
RA_CC_PAIR(12, 96, 99, 100, 103) RA_CC_PAIR(13, 104, 107, 108, 111) RA_CC_PAIR(14, 112, 115, 116, 119) RA_CC_PAIR(15, 120, 123, 124, 127)
RA_CC_PAIR(16, 128, 131, 132, 135) RA_CC_PAIR(17, 136, 139, 140, 143) RA_CC_PAIR(18, 144, 147, 148, 151) RA_CC_PAIR(19, 152, 155, 156, 159)
RA_CC_PAIR(20, 160, 163, 164, 167) RA_CC_PAIR(21, 168, 171, 172, 175) RA_CC_PAIR(22, 176, 179, 180, 183) RA_CC_PAIR(23, 184, 187, 188, 191)
RA_CC_PAIR(24, 192, 195, 196, 199) RA_CC_PAIR(25, 200, 203, 204, 207) RA_CC_PAIR(26, 208, 211, 212, 215) RA_CC_PAIR(27, 216, 219, 220, 223)
RA_CC_PAIR(28, 224, 227, 228, 231) RA_CC_PAIR(29, 232, 235, 236, 239) RA_CC_PAIR(30, 240, 243, 244, 247)
#undef RA_CC_PAIR

// one wave's private weight stream, global memory -> fragment file: the interface row_block_16 expects, fetch<position mod 62>
struct PipeReg {
    static constexpr int PF = CC_PF;
    static constexpr int MOD = CC_PF;
    static constexpr bool LONE = true;
    const char* cur;        // next fragment of this wave's stream (uniform)
    const char* beg;
    const char* end;
    unsigned voff;          // lane * 16
    f16x8 af[PF];           // values only between ready<>() and the k-step's last MFMA

    template <int FM>
    __device__ __forceinline__ void fetch() {
        CcSlot<FM>::load(voff, cur);
        cur += 1024;
        if (cur == end) cur = beg;
    }
    template <int FH, int FL>
    __device__ __forceinline__ void ready() {
        static_assert((FH & 1) == 0 && FL == FH + 1, "a k-step's fragments are an aligned slot pair");
        CcPair<FH / 2>::wait(af[FH], af[FL]);
    }
};

// the epilogue of a layer's last row block (nothing left to hide it behind): the operations of row_block_16's pending epilogue
template <int ACT>
__device__ __forceinline__ void finish_16(const Acc16& a, u32x4 (&oH)[8], u32x4 (&oL)[8]) {
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float z = a.val(e);
        if (ACT == ACT_SOFTPLUS) {
            float u = __builtin_amdgcn_exp2f(z);
            u = 1.f + u;
            u = __builtin_amdgcn_logf(u);
            t[e] = sp_finish(u, z);
        } else {
            t[e] = max0(z);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f16x2 hv;
        hv[0] = (f16)t[2 * p]; hv[1] = (f16)t[2 * p + 1];
        oH[0][p] = __builtin_bit_cast(unsigned, hv);
        oL[0][p] = pack2<f16>(t[2 * p] - (float)hv[0], t[2 * p + 1] - (float)hv[1]);
    }
}

__device__ __forceinline__ void act_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// this wave's four row blocks of a 256-row layer, then the exchange: on return Bm holds the layer's 256 outputs as B fragments
template <int KS, int ACT, int BASE, typename PipeT>
__device__ __forceinline__ void coop_layer(PipeT& P, u32x4 (&BmH)[8], u32x4 (&BmL)[8], const u32x4 (&BpH)[2], const u32x4 (&BpL)[2], const float* bias,
                                           u32x4* act, int w, int g, int lane) {
    constexpr int KH = KS == 2 ? 0 : 8, F = 2 * KS;
    Acc16 a0, a1;
    u32x4 oH[8], oL[8];                                  // [0][0..1]: the four outputs of a row block, packed
    char* const my = reinterpret_cast<char*>(act) + (w >> 1) * 2048 + lane * 16 + 8 * (w & 1);
    auto store = [&](int i) {                            // row block 4 i + w = k-step 2 i + (w >> 1), half w & 1
        uint2 h, l;
        h.x = oH[0][0]; h.y = oH[0][1]; l.x = oL[0][0]; l.y = oL[0][1];
        *reinterpret_cast<uint2*>(my + i * 4096) = h;
        *reinterpret_cast<uint2*>(my + i * 4096 + 1024) = l;
    };
    const float* b = bias + 16 * w;
    row_block_16<BASE % CC_PF, KS, ACT, false, false, false, 0, 0, PipeT, KH>(P, a0, a1, BmH, BmL, BpH, BpL, oH, oL, b, g);
    row_block_16<(BASE + F) % CC_PF, KS, ACT, true, false, false, 0, 0, PipeT, KH>(P, a1, a0, BmH, BmL, BpH, BpL, oH, oL, b + 64, g);
    act_barrier();                                       // every wave has read the previous layer's outputs (long ago): the array is free
    store(0);
    row_block_16<(BASE + 2 * F) % CC_PF, KS, ACT, true, false, false, 0, 0, PipeT, KH>(P, a0, a1, BmH, BmL, BpH, BpL, oH, oL, b + 128, g);
    store(1);
    row_block_16<(BASE + 3 * F) % CC_PF, KS, ACT, true, false, false, 0, 0, PipeT, KH>(P, a1, a0, BmH, BmL, BpH, BpL, oH, oL, b + 192, g);
    store(2);
    finish_16<ACT>(a1, oH, oL);
    store(3);
    act_barrier();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        BmH[ks] = act[(2 * ks) * 64 + lane];
        BmL[ks] = act[(2 * ks + 1) * 64 + lane];
    }
}

// one network, every wave: returns the head accumulator (rows 4 g + i of lane group g), the same in all four waves
template <int ACT, int PEL, bool SDFNET, typename PipeT>
__device__ __forceinline__ f32x4 coop_net(PipeT& P, const float (&x)[3], const float* bias, u32x4* act, int w, int g, int lane) {
    constexpr int NB = SDFNET ? CC_FRAGS / 2 : 0;        // the net's first fragment in the wave's stream
    u32x4 Bh[8], Bl[8], Bph[2], Bpl[2];
    pe_frags_16<PEL, SDFNET>(Bph, Bpl, x, g);
    coop_layer<2, ACT, NB>(P, Bh, Bl, Bph, Bpl, bias, act, w, g, lane);
    coop_layer<8, ACT, NB + 16>(P, Bh, Bl, Bph, Bpl, bias + 256, act, w, g, lane);
    coop_layer<8, ACT, NB + 80>(P, Bh, Bl, Bph, Bpl, bias + 512, act, w, g, lane);
    coop_layer<8, ACT, NB + 144>(P, Bh, Bl, Bph, Bpl, bias + 768, act, w, g, lane);
    coop_layer<10, ACT, NB + 208>(P, Bh, Bl, Bph, Bpl, bias + 1024, act, w, g, lane);
    coop_layer<8, ACT, NB + 288>(P, Bh, Bl, Bph, Bpl, bias + 1280, act, w, g, lane);
    coop_layer<8, ACT, NB + 352>(P, Bh, Bl, Bph, Bpl, bias + 1536, act, w, g, lane);
    coop_layer<8, ACT, NB + 416>(P, Bh, Bl, Bph, Bpl, bias + 1792, act, w, g, lane);
    Acc16 ah, unused;
    u32x4 oH[8], oL[8];
    row_block_16<(NB + 480) % CC_PF, 8, ACT, false, false, false, 0, 0, PipeT>(P, ah, unused, Bh, Bl, Bph, Bpl, oH, oL, bias + 2048, g);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = ah.val(e);
    return out;
}

__global__ __launch_bounds__(64 * CC_W, 1) void mlp_sdf_coop_kernel(GeoNet net, const void* __restrict__ stream, const float* __restrict__ ba, FrameState fr, MlpIO io) {
    __shared__ __attribute__((aligned(16))) CcSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    for (int i = tid; i < BIAS_ROWS * 256; i += 64 * CC_W) {          // the bias table of K3C
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
    const int ntiles = (count + 15) / 16;
    if ((int)blockIdx.x >= ntiles) return;

    PipeReg P;
    P.beg = reinterpret_cast<const char*>(stream) + (size_t)wave * CC_FRAGS * 1024;
    P.end = P.beg + (size_t)CC_FRAGS * 1024;
    P.cur = P.beg;
    P.voff = lane * 16;
    static_for<0, CC_PF>([&](auto f_) { P.template fetch<decltype(f_)::value>(); });      // the first 62 fragments; from here on the file rotates

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s = tile * 16 + c;                        // every wave, and the four lane groups of a column, hold the same point
        float x[3] = {0.f, 0.f, 0.f};
        int pidx = 0;
        float smpl = 0.f;
        if (s < count) {
            x[0] = io.bpts[3 * s]; x[1] = io.bpts[3 * s + 1]; x[2] = io.bpts[3 * s + 2];
            pidx = io.idx[s];
            if (io.smooth) smpl = io.sdf[pidx];
        }
        const f32x4 hr = coop_net<ACT_RELU, 10, false>(P, x, sm.bias, sm.act, wave, g, lane);
        float cp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float r = tanhf(hr[k]) * io.resd_limit;           // valid in lane group 0 (rows 0..2)
            cp[k] = x[k] + __shfl(r, c);
        }
        const f32x4 hs = coop_net<ACT_SOFTPLUS, 8, true>(P, cp, sm.bias + 9 * 256, sm.act, wave, g, lane);
        if (wave == 0 && g == 0 && s < count) {
            float d = hs[0] * SP_INV;
            if (io.smooth) {
                const float r = fminf(fmaxf(fabsf(d) / io.dist_th, 0.f), 1.f);
                d = smpl * r + d * (1.f - r);
            }
            io.sdf[pidx] = d;
        }
    }
}

void launch_coop(const GeoNet& net, const void* sarena_c, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream, int grid_slots) {
    const int grid = mlp_grid(max_slots, grid_slots, 16);
    hipLaunchKernelGGL(mlp_sdf_coop_kernel, dim3(grid), dim3(64 * CC_W), 0, stream, net, reinterpret_cast<const char*>(sarena_c) + (size_t)STC_FRAGS * 1024, barena, fr, io);
}

}  // namespace
