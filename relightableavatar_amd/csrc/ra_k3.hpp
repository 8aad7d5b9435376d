// K3: HDQ fine query (resd + sdf MLPs) with register-resident activations and the weights streamed through LDS.
// Implementation header: ra_k3_f16.hip / ra_k3_bf16.hip instantiate it for one operand type each (parallel builds).
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
#include "ra_stream.hpp"
#include "ra_k3_pe.hpp"

#ifdef RA_TIMESTAMPS
// Instrumented variant (tools/build_variant.sh ... "-DRA_TIMESTAMPS", tools/k3_timestamps.py): wave 0 of every workgroup records
// s_memtime at the layer boundaries of its first tile.  48 slots per workgroup: [0] tile start, [1..9] after resd L0..L7 + head,
// [10..18] after sdf L0..L7 + head, [19] tile end, [20] last tile end, [21] tiles done, [22] XCC id.  Costs a few % (MI355X_MICROARCH.md).
__device__ long long ra_k3_ts[256 * 48];
extern "C" int ra_k3_read_timestamps(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ra_k3_ts), sizeof(ra_k3_ts)); }
#define RA_STAMP(p, k) do { if (p) { __builtin_amdgcn_sched_barrier(0); (p)[k] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define RA_STAMP(p, k) do { } while (0)
#endif

namespace {

// one network: L0 (encoding) .. L7, then the <= 32-row head; returns the head accumulator (bias included)
template <typename E, int NW, bool LAST, int ACT, int PEL, bool LO, typename PipeT>
__device__ __forceinline__ f32x16 run_net(PipeT& P, const float (&x)[3], const float* bias /* 8 layer rows + head row */, int h, long long* ts) {
    u32x4 B0[16], B1[16], Bp[4];
    f32x16 accA, accB;
    pe_frags<E, PEL, LO>(Bp, x, h);
    // every layer starts on a stage boundary (32, 128, 160 / 144, 112 and 16 fragments are multiples of 16)
    layer<E, NW, 4, ACT, ACT, false>(P, accA, accB, B0 /* unused */, Bp, B0, bias, h);
    RA_STAMP(ts, 0);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 256, h);
    RA_STAMP(ts, 1);
    layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 512, h);
    RA_STAMP(ts, 2);
    if constexpr (LO) {
        // SDF net on the trimmed stream: lin3 has 205 outputs = 7 row blocks (its last one stays pending in accA: the accumulators
        // swap roles from here on), lin4 reads them in 14 hidden k-steps + 4 encoding k-steps
        layer<E, NW, 16, ACT, ACT, true, PipeT, 7>(P, accA, accB, B0, Bp, B1, bias + 768, h);
        RA_STAMP(ts, 3);
        layer<E, NW, 18, ACT, ACT, true, PipeT, 8, 14, 7>(P, accB, accA, B1, Bp, B0, bias + 1024, h);
        RA_STAMP(ts, 4);
        layer<E, NW, 16, ACT, ACT, true>(P, accB, accA, B0, Bp, B1, bias + 1280, h);
        RA_STAMP(ts, 5);
        layer<E, NW, 16, ACT, ACT, true>(P, accB, accA, B1, Bp, B0, bias + 1536, h);
        RA_STAMP(ts, 6);
        layer<E, NW, 16, ACT, ACT, true>(P, accB, accA, B0, Bp, B1, bias + 1792, h);
        RA_STAMP(ts, 7);
        row_block<E, NW, 0, 16, ACT, true, true, LAST>(P, accB, accA, B1, Bp, B1[14], B1[15], bias + 2048, h);
        RA_STAMP(ts, 8);
        return accB;
    } else {
        layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 768, h);
        RA_STAMP(ts, 3);
        layer<E, NW, 20, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 1024, h);
        RA_STAMP(ts, 4);
        layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 1280, h);
        RA_STAMP(ts, 5);
        layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 1536, h);
        RA_STAMP(ts, 6);
        layer<E, NW, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 1792, h);
        RA_STAMP(ts, 7);
        row_block<E, NW, 0, 16, ACT, true, true, LAST>(P, accA, accB, B1, Bp, B1[14], B1[15], bias + 2048, h);
        RA_STAMP(ts, 8);
        return accA;
    }
}

// the same network as pairs of row blocks (ra_stream.hpp, row_blocks): the latency variant of the 2- and 4-wave workgroups
template <typename E, int NW, bool LAST, int ACT, int PEL, bool LO, typename PipeT>
__device__ __forceinline__ f32x16 run_net_pairs(PipeT& P, const float (&x)[3], const float* bias, int h, long long* ts) {
    u32x4 B0[16], B1[16], Bp[4];
    f32x16 a0, a1, b0, b1;
    pe_frags<E, PEL, LO>(Bp, x, h);
    layer_pairs<E, NW, 4, ACT, ACT, false>(P, a0, a1, b0, b1, B0 /* unused */, Bp, B0, bias, h);
    RA_STAMP(ts, 0);
    layer_pairs<E, NW, 16, ACT, ACT, true>(P, a0, a1, b0, b1, B0, Bp, B1, bias + 256, h);
    RA_STAMP(ts, 1);
    layer_pairs<E, NW, 16, ACT, ACT, true>(P, a0, a1, b0, b1, B1, Bp, B0, bias + 512, h);
    RA_STAMP(ts, 2);
    layer_pairs<E, NW, 16, ACT, ACT, true>(P, a0, a1, b0, b1, B0, Bp, B1, bias + 768, h);
    RA_STAMP(ts, 3);
    layer_pairs<E, NW, 20, ACT, ACT, true>(P, a0, a1, b0, b1, B1, Bp, B0, bias + 1024, h);
    RA_STAMP(ts, 4);
    layer_pairs<E, NW, 16, ACT, ACT, true>(P, a0, a1, b0, b1, B0, Bp, B1, bias + 1280, h);
    RA_STAMP(ts, 5);
    layer_pairs<E, NW, 16, ACT, ACT, true>(P, a0, a1, b0, b1, B1, Bp, B0, bias + 1536, h);
    RA_STAMP(ts, 6);
    layer_pairs<E, NW, 16, ACT, ACT, true>(P, a0, a1, b0, b1, B0, Bp, B1, bias + 1792, h);
    RA_STAMP(ts, 7);
    row_blocks<E, NW, 1, 0, 16, ACT, true, true, LAST>(P, a0, a1, b0, b1, B1, Bp, B1[12], B1[13], B1[14], B1[15], bias + 2048, h);
    RA_STAMP(ts, 8);
    return a0;
}

template <typename E, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void mlp_sdf_stream_kernel(GeoNet net, const void* __restrict__ stream, const float* __restrict__ ba,
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
    if (blockIdx.x == 0 && tid == 0 && io.counters) {
        atomicAdd(&io.counters->n_fine_sdf, (unsigned long long)count);
        if (NW == 8) atomicAdd(&io.counters->n_fine_sdf_wide, (unsigned long long)count);
    }
    const int ntiles = (count + ST_TM - 1) / ST_TM;
    if ((int)blockIdx.x >= ntiles) return;

    constexpr bool PAIRS = NW < 8;          // one wave per SIMD: two row blocks in flight (pair-ordered stream), A fragments read 8 ahead
    constexpr int STAGES = PAIRS ? ST_STAGES : ST_STAGES_TRIM;          // the 8-wave kernel walks the trimmed stream
    Pipe<E, NW, STAGES, PAIRS ? 8 : ST_PF> P;
    P.g = reinterpret_cast<const char*>(stream);
    P.voff = wave * (16 / NW) * 1024 + lane * 16;
    P.ring = reinterpret_cast<const char*>(sm.ring) + lane * 16;
    P.ring_addr = (unsigned)(size_t)sm.ring + wave * (16 / NW) * 1024;
    P.slot = ST_RING - 1;            // the first sync_stage() advances to slot 0 / stream stage 0
    P.sstage = STAGES - 1;
    P.rd = P.ring;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < ST_AHEAD; ++st) P.issue(st, st);

    long long* ts = nullptr;
#ifdef RA_TIMESTAMPS
    int tiles_done = 0;
#endif
    // Whole rounds of 256-point tiles first (tile = round * grid + workgroup).  The LAST round is usually partly filled: with the plain
    // deal its tiles keep a few CUs busy for a full tile time (107 us with two waves per SIMD) while the others idle.  When at most a quarter
    // of the workgroups would get a tile, the 8-wave kernel SPREADS the remaining points over all of them instead: each takes 32 * wact
    // consecutive points on its first wact <= 2 waves — one wave per SIMD, a tile then takes ~65 us — and its other waves only keep the
    // weight stream and the stage barriers going (Pipe::sync_stage).  Which wave computes a point does not change its arithmetic.
    // Measured on one box (tools/bench_mlp.py): 470 400 points 0.894 -> 0.852 ms; with three active waves (tried) it is a wash.
    const int G = gridDim.x;
    const int full = ntiles / G, remt = ntiles - full * G;
    const bool spread = NW == 8 && full >= 1 && remt > 0 && 4 * remt <= G;
    int wact = NW, tail_start = 0;
    if (spread) {
        const int base = full * G * ST_TM;
        wact = (count - base + 32 * G - 1) / (32 * G);          // 1 .. 2
        tail_start = base + (int)blockIdx.x * 32 * wact;
    }
    const int n_it = full + ((spread ? tail_start < count : (int)blockIdx.x < remt) ? 1 : 0);
    for (int it = 0; it < n_it; ++it) {
        const bool spread_tile = spread && it == full;
#ifdef RA_TIMESTAMPS
        ts = (wave == 0 && lane == 0 && it == 0) ? ra_k3_ts + blockIdx.x * 48 : nullptr;
        RA_STAMP(ts, 0);
        if (ts) ts += 1;
#endif
        if (spread_tile && wave >= wact) {          // wave-uniform: this wave has no points in the spread tile
            for (int st = 0; st < STAGES; ++st) P.sync_stage();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        const int s = spread_tile ? tail_start + wave * 32 + c : (it * G + (int)blockIdx.x) * ST_TM + wave * 32 + c;
        float x[3] = {0.f, 0.f, 0.f};
        int pidx = 0;
        float smpl = 0.f;
        if (s < count) {
            x[0] = io.bpts[3 * s]; x[1] = io.bpts[3 * s + 1]; x[2] = io.bpts[3 * s + 2];
            pidx = io.idx[s];
            if (io.smooth) smpl = io.sdf[pidx];
        }
        // the first ST_PF fragments of the tile (stage 0 of the stream)
        static_for<0, decltype(P)::PF>([&](auto f_) { P.template fetch<decltype(f_)::value>(); });
        // ---- residual deformation net (ReLU); head: resd = tanh(z) * resd_limit, cpts = bpts + resd
        f32x16 hr;
        if constexpr (PAIRS) hr = run_net_pairs<E, NW, false, ACT_RELU, 10, false>(P, x, sm.bias, h, ts);
        else hr = run_net<E, NW, false, ACT_RELU, 10, false>(P, x, sm.bias, h, ts);
        float cp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float r = tanhf(hr[k]) * io.resd_limit;           // valid in lanes h = 0 (rows 0..2)
            cp[k] = x[k] + __shfl(r, c);
        }
        // ---- signed distance net (softplus, scaled domain); head row 0 = sdf
        f32x16 hs;
        if constexpr (PAIRS) hs = run_net_pairs<E, NW, true, ACT_SOFTPLUS, 8, true>(P, cp, sm.bias + 9 * 256, h, ts ? ts + 9 : nullptr);
        else hs = run_net<E, NW, true, ACT_SOFTPLUS, 8, true>(P, cp, sm.bias + 9 * 256, h, ts ? ts + 9 : nullptr);
        if (h == 0 && s < count) {
            float d = hs[0] * SP_INV;                                 // head accumulates beta*log2(e) * sdf
            if (io.smooth) {                                          // HDQ blend (base_network.py:374-382)
                const float r = fminf(fmaxf(fabsf(d) / io.dist_th, 0.f), 1.f);
                d = smpl * r + d * (1.f - r);
            }
            io.sdf[pidx] = d;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RA_TIMESTAMPS
        RA_STAMP(ts, 18);
        ++tiles_done;
        if (wave == 0 && lane == 0) {
            long long* t0 = ra_k3_ts + blockIdx.x * 48;
            t0[20] = __builtin_readcyclecounter(); t0[21] = tiles_done;
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            t0[22] = xcc & 0xf;
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace

template <typename E, int NW>
static void launch_nw(const GeoNet& net, const void* sarena, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream, int grid_slots) {
    const int grid = mlp_grid(max_slots, grid_slots, 32 * NW);
    hipLaunchKernelGGL((mlp_sdf_stream_kernel<E, NW>), dim3(grid), dim3(64 * NW), 0, stream, net, sarena, barena, fr, io);
}

// A launch that cannot fill the 256 CUs with 256-point tiles is bound by the latency of ONE tile (1952 MFMAs per
// wave): narrower workgroups put one wave on a SIMD instead of two (86 -> 67 -> 58 us per tile for 8 / 4 / 2 waves).
// max_slots is only an upper bound of the device-side count: up to 65536 the 4-wave variant needs at most the two
// rounds that equal one 8-wave round, and one when the real count is below half.  All widths are bit-identical.
template <typename E>
static void launch_k3(const GeoNet& net, const void* sarena, const void* sarena_pairs, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots,
                      hipStream_t stream, int grid_slots) {
    if (max_slots <= 0) return;
    int nw = k3_waves(max_slots);
#ifdef RA_TESTING            // test / experiment builds only (tools/build_variant.sh): force the workgroup width
    static const int force = getenv("RA_STREAM_NW") ? atoi(getenv("RA_STREAM_NW")) : 0;
    if (force) nw = force;
#endif
    if (nw == 2) launch_nw<E, 2>(net, sarena_pairs, barena, fr, io, max_slots, stream, grid_slots);
    else if (nw == 4) launch_nw<E, 4>(net, sarena_pairs, barena, fr, io, max_slots, stream, grid_slots);
    else launch_nw<E, 8>(net, sarena, barena, fr, io, max_slots, stream, grid_slots);
}
