// K3 with 64 points per wave (experiment, see ra_stream.hpp row_block2): its own translation unit so that a variant builds in two minutes.
#include "ra_stream.hpp"
#include "ra_k3_pe.hpp"

namespace {

// the network for a wave that owns 64 points (two column sets), on the trimmed stream of the 8-wave kernel: same fragments, same order,
// same arithmetic per point -> bit-identical distances
template <typename E, bool LAST, int ACT, int PEL, bool LO, typename PipeT>
__device__ __forceinline__ void run_net2(PipeT& P, const float (&x)[2][3], const float* bias, int h, f32x16 (&head)[2]) {
    u32x4 B0[2][16], B1[2][16], Bp[2][4];
    f32x16 accA[2], accB[2];
    pe_frags<E, PEL, LO>(Bp[0], x[0], h);
    pe_frags<E, PEL, LO>(Bp[1], x[1], h);
    layer2<E, 4, ACT, ACT, false>(P, accA, accB, B0 /* unused */, Bp, B0, bias, h);
    layer2<E, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 256, h);
    layer2<E, 16, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 512, h);
    if constexpr (LO) {
        layer2<E, 16, ACT, ACT, true, PipeT, 7>(P, accA, accB, B0, Bp, B1, bias + 768, h);
        layer2<E, 18, ACT, ACT, true, PipeT, 8, 14, 7>(P, accB, accA, B1, Bp, B0, bias + 1024, h);
        layer2<E, 16, ACT, ACT, true>(P, accB, accA, B0, Bp, B1, bias + 1280, h);
        layer2<E, 16, ACT, ACT, true>(P, accB, accA, B1, Bp, B0, bias + 1536, h);
        layer2<E, 16, ACT, ACT, true>(P, accB, accA, B0, Bp, B1, bias + 1792, h);
        row_block2<E, 0, 16, ACT, true, true, LAST, 14, PipeT>(P, accB, accA, B1, Bp, B1, bias + 2048, h);
        head[0] = accB[0]; head[1] = accB[1];
    } else {
        layer2<E, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 768, h);
        layer2<E, 20, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 1024, h);
        layer2<E, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 1280, h);
        layer2<E, 16, ACT, ACT, true>(P, accA, accB, B1, Bp, B0, bias + 1536, h);
        layer2<E, 16, ACT, ACT, true>(P, accA, accB, B0, Bp, B1, bias + 1792, h);
        row_block2<E, 0, 16, ACT, true, true, LAST, 14, PipeT>(P, accA, accB, B1, Bp, B1, bias + 2048, h);
        head[0] = accA[0]; head[1] = accA[1];
    }
}

template <typename E>
__global__ __launch_bounds__(256, 1) void mlp_sdf_stream64_kernel(GeoNet net, const void* __restrict__ stream, const float* __restrict__ ba, FrameState fr, MlpIO io) {
    constexpr int NW = 4;
    __shared__ __attribute__((aligned(16))) StSmem<E> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    constexpr int ST_TM = 64 * NW;     // points per workgroup tile
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
        atomicAdd(&io.counters->n_fine_sdf_wide, (unsigned long long)count);
    }
    const int ntiles = (count + ST_TM - 1) / ST_TM;
    if ((int)blockIdx.x >= ntiles) return;
    Pipe<E, NW, ST_STAGES_TRIM, 8> P;
    P.g = reinterpret_cast<const char*>(stream);
    P.voff = wave * (16 / NW) * 1024 + lane * 16;
    P.ring = reinterpret_cast<const char*>(sm.ring) + lane * 16;
    P.ring_addr = (unsigned)(size_t)sm.ring + wave * (16 / NW) * 1024;
    P.slot = ST_RING - 1;
    P.sstage = ST_STAGES_TRIM - 1;
    P.rd = P.ring;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < ST_AHEAD; ++st) P.issue(st, st);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float x[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        int pidx[2] = {0, 0}, sl[2];
        float smpl[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            sl[k] = tile * ST_TM + wave * 64 + k * 32 + c;
            if (sl[k] < count) {
                x[k][0] = io.bpts[3 * sl[k]]; x[k][1] = io.bpts[3 * sl[k] + 1]; x[k][2] = io.bpts[3 * sl[k] + 2];
                pidx[k] = io.idx[sl[k]];
                if (io.smooth) smpl[k] = io.sdf[pidx[k]];
            }
        }
        static_for<0, 8>([&](auto f_) { P.template fetch<decltype(f_)::value>(); });
        f32x16 hr[2];
        run_net2<E, false, ACT_RELU, 10, false>(P, x, sm.bias, h, hr);
        float cp[2][3];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float r = tanhf(hr[k][j]) * io.resd_limit;
                cp[k][j] = x[k][j] + __shfl(r, c);
            }
        f32x16 hs[2];
        run_net2<E, true, ACT_SOFTPLUS, 8, true>(P, cp, sm.bias + 9 * 256, h, hs);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (h == 0 && sl[k] < count) {
                float d = hs[k][0] * SP_INV;
                if (io.smooth) {
                    const float r = fminf(fmaxf(fabsf(d) / io.dist_th, 0.f), 1.f);
                    d = smpl[k] * r + d * (1.f - r);
                }
                io.sdf[pidx[k]] = d;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace
