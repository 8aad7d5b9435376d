// K3 inner-loop laboratory (stand-alone gfx950 micro-benchmark; not part of the product path).
// Reproduces the structure of mlp_sdf_stream_kernel's steady state — a wave owns N = 32*NC point columns, a layer is 8 row
// blocks of 16 dependent v_mfma_f32_32x32x16_f16 on one accumulator (per column set), the A fragment of every MFMA is a
// 1 KB ds_read_b128 from an LDS ring, the bias/activation/pack epilogue of row block rb-1 runs interleaved with the MFMAs of
// row block rb and its packed output is the next layer's B operand — without the weight DMA, so that the cost of each
// ingredient can be priced on its own:  waves per SIMD (1 | 2), column sets per wave (1 | 2), epilogue kind, barrier per stage.
// Prints ns per MFMA (SIMD view), TFLOP/s, cycles per MFMA (s_memtime) and the effective clock.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/k3lab.hip -o tools/k3lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__device__ __forceinline__ float max0(float z) { return __builtin_amdgcn_fmed3f(z, 0.f, 3.0e38f); }
__device__ __forceinline__ unsigned pack2(float a, float b) { f16x2 v; v[0] = (f16)a; v[1] = (f16)b; return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned pk_max0(unsigned w) {
    f16x2 v = __builtin_bit_cast(f16x2, w);
    v = __builtin_elementwise_max(v, f16x2{(f16)0, (f16)0});
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pk_fma(unsigned a, unsigned b, unsigned c) {
    f16x2 r = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned pk_const(float x) { return pack2(x, x); }

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_SP = 2, EPI_SP_MED3 = 3, EPI_SP_POLY = 4, EPI_SP_PIPE = 5, EPI_SP_MED3_PIPE = 6 };

// epilogue of one element pair (e-1, e) of a pending accumulator -> one packed word
template <int EPI>
__device__ __forceinline__ unsigned epi_pair(float z0, float z1) {
    if constexpr (EPI == EPI_RELU) {
        return pk_max0(pack2(z0, z1));
    } else if constexpr (EPI == EPI_SP) {
        const float e0 = __builtin_amdgcn_exp2f(-__builtin_fabsf(z0)), e1 = __builtin_amdgcn_exp2f(-__builtin_fabsf(z1));
        const float y0 = max0(z0) + __builtin_amdgcn_logf(1.f + e0), y1 = max0(z1) + __builtin_amdgcn_logf(1.f + e1);
        return pack2(y0, y1);
    } else if constexpr (EPI == EPI_SP_MED3) {
        // log2(1 + 2^z) overflows to +inf for z >= 128; med3(C, z, 64) = C for z <= 64 (C in [z, 64]) and z beyond (C >= z > 64)
        const float c0 = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(z0)), c1 = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(z1));
        return pack2(__builtin_amdgcn_fmed3f(c0, z0, 64.f), __builtin_amdgcn_fmed3f(c1, z1, 64.f));
    } else if constexpr (EPI == EPI_SP_POLY) {
        // y = max(z, 0) + g(u), u = 2^-|z|, g(u) = log2(1 + u) ~ u (c1 + u (c2 + u c3)) evaluated on packed f16 pairs
        const float u0 = __builtin_amdgcn_exp2f(-__builtin_fabsf(z0)), u1 = __builtin_amdgcn_exp2f(-__builtin_fabsf(z1));
        const unsigned uh = pack2(u0, u1), m = pk_max0(pack2(z0, z1));
        unsigned p = pk_fma(uh, pk_const(0.16538342f), pk_const(-0.58920629f));
        p = pk_fma(p, uh, pk_const(1.42459377f));
        return pk_fma(p, uh, m);
    } else {
        return pack2(z0, z1);
    }
}

// the product's weight stream (ra_stream.hpp Pipe, NW = 8): 16 KB stages, 8-slot ring, 6 stages in flight, each wave moves 2 fragments
struct LabPipe {
    const char* g; unsigned voff; const char* ring; unsigned ring_addr; unsigned slot; int sstage; const char* rd;
    static constexpr int STAGES = 122, AHEAD = 6, RING = 8;
    __device__ __forceinline__ void issue(int stream_stage, unsigned ring_slot) {
        const char* sb = g;
        asm volatile("" : "+s"(sb));
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + ring_slot * 16384);
        const char* src = sb + (size_t)stream_stage * 16384;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                     "s_add_u32 m0, %4, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "v"(voff + 1024), "s"(src), "s"(dst) : "memory", "scc");
    }
    __device__ __forceinline__ void sync_stage() {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (AHEAD - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = (slot + 1) & (RING - 1);
        sstage = sstage + 1 == STAGES ? 0 : sstage + 1;
        int ahead = sstage + AHEAD;
        ahead = ahead >= STAGES ? ahead - STAGES : ahead;
        issue(ahead, (slot + AHEAD) & (RING - 1));
        rd = ring + slot * 16384;
    }
};

template <int NC, int EPI, bool LDSA, bool BAR, int rb, bool BIAS = false, bool DMA = false, typename PipeT = int>
__device__ __forceinline__ void row_block(PipeT& pipe, const char* rd, const f16x8& areg, f32x16 (&acc)[NC], f32x16 (&accPrev)[NC], const u32x4 (&Bin)[16][NC], u32x4 (&o0)[NC], u32x4 (&o1)[NC]) {
        float ta[NC][16], tb[NC][16];
        unsigned off = threadIdx.x & 63;
        asm volatile("" : "+v"(off));          // keeps the (loop-invariant) fragment reads inside the loop
        extern __shared__ __attribute__((aligned(16))) char smem_[];
        const char* rdl = smem_ + off * 16;
        if constexpr (BIAS) {                  // accumulators start at the bias of their rows: 4 ds_read_b128 per row block (as the product)
            const float* bt = reinterpret_cast<const float*>(smem_ + 131072) + rb * 32 + 4 * (off >> 5);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bt + 8 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[c][4 * q + i] = bv[i];
                }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c][i] = 0.25f;            // stands for the bias row
        }
        static_for<0, 16>([&](auto ks_) {
            constexpr int ks = decltype(ks_)::value;
            f16x8 a = areg;
            if constexpr (DMA) {
                if constexpr (ks == 0) pipe.sync_stage();              // counted vmcnt + barrier + the next stage's DMA (product: Pipe::sync_stage)
                a = *reinterpret_cast<const f16x8*>(pipe.rd + ks * 1024);
            } else if constexpr (LDSA) a = *reinterpret_cast<const f16x8*>(rdl + ((rb * 16 + ks) & 63) * 1024);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, Bin[ks][c]), acc[c], 0, 0, 0);
            if constexpr (BAR && !DMA) if constexpr (ks == 0) __builtin_amdgcn_s_barrier();
            if constexpr (EPI == EPI_SP_PIPE || EPI == EPI_SP_MED3_PIPE) {
                // the product kernel's software pipeline: element e starts at slot s0 = e * 13 / 16 and takes 4 slots
                static_for<0, 16>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    constexpr int s0 = (e * 13) / 16;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if constexpr (EPI == EPI_SP_PIPE) {
                            if constexpr (s0 == ks) { ta[c][e] = __builtin_amdgcn_exp2f(-__builtin_fabsf(accPrev[c][e])); tb[c][e] = max0(accPrev[c][e]); }
                            if constexpr (s0 + 1 == ks) ta[c][e] = 1.f + ta[c][e];
                            if constexpr (s0 + 2 == ks) ta[c][e] = __builtin_amdgcn_logf(ta[c][e]);
                            if constexpr (s0 + 3 == ks) ta[c][e] = ta[c][e] + tb[c][e];
                        } else {
                            if constexpr (s0 == ks) ta[c][e] = __builtin_amdgcn_exp2f(accPrev[c][e]);
                            if constexpr (s0 + 1 == ks) ta[c][e] = 1.f + ta[c][e];
                            if constexpr (s0 + 2 == ks) ta[c][e] = __builtin_amdgcn_logf(ta[c][e]);
                            if constexpr (s0 + 3 == ks) ta[c][e] = __builtin_amdgcn_fmed3f(ta[c][e], accPrev[c][e], 64.f);
                        }
                        if constexpr ((e & 1) && s0 + 3 == ks) {
                            const unsigned wv = pack2(ta[c][e - 1], ta[c][e]);
                            if constexpr (e < 8) o0[c][e >> 1] = wv; else o1[c][(e >> 1) & 3] = wv;
                        }
                    }
                });
            } else if constexpr (EPI != EPI_NONE) {
                if constexpr (ks >= 4 && ks < 12) {            // one element pair per slot in slots 4..11
                    constexpr int e = 2 * (ks - 4);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const unsigned wv = epi_pair<EPI>(accPrev[c][e], accPrev[c][e + 1]);
                        if constexpr (e < 8) o0[c][e >> 1] = wv; else o1[c][(e >> 1) & 3] = wv;
                    }
                }
            }
        });
        if constexpr (EPI == EPI_NONE) {
#pragma unroll
            for (int c = 0; c < NC; ++c) asm volatile("" :: "v"(accPrev[c]));
        }
}

template <int NC, int EPI, bool LDSA, bool BAR, bool BIAS = false, bool DMA = false, typename PipeT = int>
__device__ __forceinline__ void layer(PipeT& pipe, const char* rd, const f16x8& areg, f32x16 (&accA)[NC], f32x16 (&accB)[NC], u32x4 (&Bin)[16][NC], u32x4 (&Bout)[16][NC]) {
        row_block<NC, EPI, LDSA, BAR, 0, BIAS, DMA>(pipe, rd, areg, accA, accB, Bin, Bin[14], Bin[15]);     // pending rb 7 of the previous layer -> Bin[14], Bin[15]
        row_block<NC, EPI, LDSA, BAR, 1, BIAS, DMA>(pipe, rd, areg, accB, accA, Bin, Bout[0], Bout[1]);
        row_block<NC, EPI, LDSA, BAR, 2, BIAS, DMA>(pipe, rd, areg, accA, accB, Bin, Bout[2], Bout[3]);
        row_block<NC, EPI, LDSA, BAR, 3, BIAS, DMA>(pipe, rd, areg, accB, accA, Bin, Bout[4], Bout[5]);
        row_block<NC, EPI, LDSA, BAR, 4, BIAS, DMA>(pipe, rd, areg, accA, accB, Bin, Bout[6], Bout[7]);
        row_block<NC, EPI, LDSA, BAR, 5, BIAS, DMA>(pipe, rd, areg, accB, accA, Bin, Bout[8], Bout[9]);
        row_block<NC, EPI, LDSA, BAR, 6, BIAS, DMA>(pipe, rd, areg, accA, accB, Bin, Bout[10], Bout[11]);
        row_block<NC, EPI, LDSA, BAR, 7, BIAS, DMA>(pipe, rd, areg, accB, accA, Bin, Bout[12], Bout[13]);
}

template <int WPS, int NC, int EPI, bool LDSA, bool BAR, bool BIAS = false, bool DMA = false>
__global__ __launch_bounds__(256 * WPS, WPS) void lab(const f16* __restrict__ w, float* __restrict__ out, long long* __restrict__ cyc, int iters, const char* __restrict__ wstream) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KB: 4 "stages" of 16 fragments (DMA: 128 KB ring); + 8 KB bias table at 128 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 16; i += 256 * WPS) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(w)[i];
    for (int i = tid; i < 2048; i += 256 * WPS) reinterpret_cast<float*>(smem + 131072)[i] = 0.01f * ((i * 7) % 41 - 20);
    __syncthreads();
    const char* rd = smem + lane * 16;
    LabPipe pipe;
    if constexpr (DMA) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        pipe.g = wstream; pipe.voff = wave * 2048 + lane * 16; pipe.ring = smem + lane * 16;
        pipe.ring_addr = (unsigned)(size_t)smem + wave * 2048; pipe.slot = 7; pipe.sstage = 121; pipe.rd = pipe.ring;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int st = 0; st < LabPipe::AHEAD; ++st) pipe.issue(st, st);
    }
    u32x4 B[16][NC], Bo[16][NC];
    f32x16 accA[NC], accB[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { B[k][c][j] = pack2(0.01f * ((tid * 7 + k * 5 + j * 3 + c) % 23 - 11), 0.013f * ((tid * 3 + k + j * 7 + c) % 19 - 9)); Bo[k][c][j] = B[k][c][j]; }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accA[c][i] = 0.f; accB[c][i] = 0.1f * ((tid + i) % 7 - 3); }
    f16x8 areg;
#pragma unroll
    for (int i = 0; i < 8; ++i) areg[i] = (f16)(0.02f * ((tid * 5 + i * 3) % 17 - 8));
    const long long t0 = __builtin_readcyclecounter();

    for (int it = 0; it < iters; ++it) {
        layer<NC, EPI, LDSA, BAR, BIAS, DMA>(pipe, rd, areg, accA, accB, B, Bo);
        layer<NC, EPI, LDSA, BAR, BIAS, DMA>(pipe, rd, areg, accA, accB, Bo, B);
    }
    if constexpr (DMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s += accA[c][i] + accB[c][i];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += (float)__builtin_bit_cast(f16x2, B[k][c][0])[0] + (float)__builtin_bit_cast(f16x2, Bo[k][c][1])[1];
    }
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

static int g_long = 1;
static f16* d_w; static char* d_stream; static float* d_out; static long long* d_cyc;

template <int WPS, int NC, int EPI, bool LDSA, bool BAR, bool BIAS = false, bool DMA = false>
void run(const char* name) {
    const int iters = (g_long ? 2400 : 60) / NC;
    auto kern = lab<WPS, NC, EPI, LDSA, BAR, BIAS, DMA>;
    constexpr int LDSB = 131072 + 8192;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WPS), LDSB, 0, d_w, d_out, d_cyc, 4, d_stream);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WPS), LDSB, 0, d_w, d_out, d_cyc, iters, d_stream);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<long long> cyc(256);
    hipMemcpy(cyc.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
    double cavg = 0; for (auto c : cyc) cavg += c; cavg /= 256;
    const double mfma_per_wave = (double)iters * 2 * 8 * 16 * NC;
    const double mfma_per_simd = mfma_per_wave * WPS;
    const double ns = best * 1e6 / mfma_per_simd;
    // s_memtime counts at a fixed 100 MHz on gfx950 (REFCLK); report both raw ticks and derived values
    const double cyc_simd = cavg / mfma_per_wave / WPS;
    printf("%-34s wps %d nc %d lds %d bar %d bias %d dma %d: %8.3f ms  %6.2f ns/MFMA(SIMD)  %5.0f TF  %6.2f cycles/MFMA(SIMD)  clock %.2f GHz\n", name, WPS, NC, (int)LDSA, (int)BAR, (int)BIAS, (int)DMA, best, ns,
           mfma_per_simd * 1024 * 32768.0 / (best * 1e-3) / 1e12, cyc_simd, cyc_simd / ns);
    fflush(stdout);
}

int main(int argc, char** argv) {
    std::vector<f16> w(32768);
    unsigned s = 12345u;
    for (auto& x : w) { s = s * 1664525u + 1013904223u; x = (f16)(((int)(s >> 16) % 2001 - 1000) * (1.0f / 1000.f) * 0.08f); }
    hipMalloc(&d_w, 65536); hipMemcpy(d_w, w.data(), 65536, hipMemcpyHostToDevice);
    hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 256 * 8);
    {   // 2 MB weight stream (122 stages of 16 KB) for the DMA variants
        std::vector<f16> ws(122 * 8192);
        for (auto& x : ws) { s = s * 1664525u + 1013904223u; x = (f16)(((int)(s >> 16) % 2001 - 1000) * (1.0f / 1000.f) * 0.08f); }
        hipMalloc(&d_stream, ws.size() * 2); hipMemcpy(d_stream, ws.data(), ws.size() * 2, hipMemcpyHostToDevice);
    }
    const int sel = argc > 1 ? atoi(argv[1]) : 0;
    if (argc > 2) g_long = atoi(argv[2]);
    if (sel == 0 || sel == 1) {
        run<2, 1, EPI_NONE, true, false>("mfma + ds_read");
        run<2, 1, EPI_NONE, true, true>("mfma + ds_read + barrier");
        run<2, 1, EPI_RELU, true, false>("relu");
        run<2, 1, EPI_SP, true, false>("softplus (exp,add,log,med3,add)");
        run<2, 1, EPI_SP_PIPE, true, false>("softplus, product pipeline");
        run<2, 1, EPI_SP_MED3, true, false>("softplus med3 form");
        run<2, 1, EPI_SP_MED3_PIPE, true, false>("softplus med3 form, pipelined");
        run<2, 1, EPI_SP_POLY, true, false>("softplus 1 trans + pk_f16 poly");
    }
    if (sel == 3) {          // what the product's machinery adds to the bare loop: bias reads, barrier, the LDS-DMA weight stream
        run<2, 1, EPI_RELU, true, false>("relu");
        run<2, 1, EPI_RELU, true, false, true>("relu + bias reads");
        run<2, 1, EPI_RELU, true, true, true>("relu + bias + barrier");
        run<2, 1, EPI_RELU, true, true, true, true>("relu + bias + barrier + DMA");
        run<2, 1, EPI_SP_MED3_PIPE, true, false>("softplus med3");
        run<2, 1, EPI_SP_MED3_PIPE, true, false, true>("softplus med3 + bias reads");
        run<2, 1, EPI_SP_MED3_PIPE, true, true, true>("softplus med3 + bias + barrier");
        run<2, 1, EPI_SP_MED3_PIPE, true, true, true, true>("softplus med3 + bias + barrier + DMA");
        run<2, 1, EPI_NONE, true, true, false, true>("mfma + ds_read + barrier + DMA");
    }
    if (sel == 0 || sel == 2) {
        run<1, 1, EPI_NONE, true, false>("1 wave/SIMD: mfma + ds_read");
        run<1, 1, EPI_RELU, true, false>("1 wave/SIMD: relu");
        run<1, 1, EPI_SP_PIPE, true, false>("1 wave/SIMD: softplus pipeline");
        run<1, 2, EPI_NONE, true, false>("1 wave/SIMD x 64 cols: mfma+ds");
        run<1, 2, EPI_RELU, true, false>("1 wave/SIMD x 64 cols: relu");
        run<1, 2, EPI_SP_PIPE, true, false>("1 wave/SIMD x 64 cols: softplus");
        run<1, 2, EPI_SP_MED3_PIPE, true, false>("1 wave/SIMD x 64 cols: sp med3");
        run<1, 2, EPI_SP_POLY, true, false>("1 wave/SIMD x 64 cols: sp poly");
    }
    return 0;
}
