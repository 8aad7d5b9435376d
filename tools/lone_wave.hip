// What limits ONE wave per SIMD that feeds its MFMAs from LDS?  (the 2- / 4-wave K3 variants run 61 cycles per MFMA in the ReLU
// layers although a register-fed chain issues every 32.)  Loop of 16 x { v_mfma_f32_32x32x16_f16 ; ds_read_b128 of the A fragment
// used DIST slots later ; K plain VALU ops }, everything in inline asm so that the compiler cannot reorder it.
//   hipcc --offload-arch=gfx950 -O3 tools/lone_wave.hip -o tools/lone_wave.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// MODE 0: MFMA only (A from registers); 1: + ds_read_b128 per MFMA, waited for DIST slots later; 2: as 1 but the read result is never
// used by an MFMA (A stays in registers): the read's issue cost alone; 3: as 1 with two ds_read_b64 instead of one b128
template <int WPS, int MODE, int DIST, int K>
__global__ __launch_bounds__(256 * WPS, WPS) void k(const unsigned* __restrict__ w, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256 * WPS) reinterpret_cast<unsigned*>(smem)[i] = w[i];
    __syncthreads();
    const unsigned lds = (unsigned)(size_t)smem + (threadIdx.x & 63) * 16;
    f32x16 acc0 = {}, acc1 = {};
    u32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const u32x4*>(smem + i * 1024 + (threadIdx.x & 63) * 16);
    u32x4 b = *reinterpret_cast<const u32x4*>(smem + 9000 + (threadIdx.x & 63) * 16);
    u32x4 junk = a[0];
    float v[4] = {1.f, 2.f, 3.f, 4.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // the fragment of slot s was requested DIST slots ago: wait until at most DIST - 1 younger reads are outstanding
            if (MODE == 1 || MODE == 3) {
                if (DIST == 4) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(MODE == 3 ? 6 : 3));
                else asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(MODE == 3 ? 14 : 7));
            }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a[s % DIST]), "v"(b));
            if (MODE == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[s % DIST]) : "v"(lds), "n"((s * 1024) & 0xffff));
            if (MODE == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(junk) : "v"(lds), "n"((s * 1024) & 0xffff));
            if (MODE == 3) {
                unsigned long long lo, hi;
                asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=v"(lo), "=v"(hi) : "v"(lds), "n"((s * 1024) & 0xffff), "n"(((s * 1024) & 0xffff) + 8));
                a[s % DIST] = u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
            }
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j % 4]));
        }
        if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = v[0] + v[1] + v[2] + v[3] + __builtin_bit_cast(float, junk[0]);
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static unsigned* d_w; static float* d_out; static long long* d_cyc;
template <int WPS, int MODE, int DIST, int K>
void run(const char* name) {
    auto kern = k<WPS, MODE, DIST, K>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WPS), 65536, 0, d_w, d_out, d_cyc, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WPS), 65536, 0, d_w, d_out, d_cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[256]; hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
    double ca = 0; for (auto x : c) ca += x; ca /= 256;
    const double per = ca / (iters * 16.0);
    printf("%-44s waves/SIMD %d: %6.1f cycles per MFMA per wave, %6.1f per SIMD, %.2f ns/MFMA(SIMD), clock %.2f GHz\n", name, WPS, per, per / WPS,
           ms * 1e6 / (iters * 16.0 * WPS), ca / (ms * 1e6));
}

int main() {
    unsigned h[16384]; unsigned s = 1;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = 0x2c002c00u ^ (s & 0x83ff83ffu); }       // f16 pairs, |x| in [0.06, 0.12)
    hipMalloc(&d_w, sizeof(h)); hipMemcpy(d_w, h, sizeof(h), hipMemcpyHostToDevice);
    hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 256 * 8);
    run<1, 0, 4, 0>("MFMA only");
    run<1, 0, 4, 2>("MFMA + 2 v_fma");
    run<1, 0, 4, 5>("MFMA + 5 v_fma");
    run<1, 2, 4, 0>("MFMA + ds_read_b128 (result unused)");
    run<1, 1, 4, 0>("MFMA + ds_read_b128, 4 ahead");
    run<1, 1, 8, 0>("MFMA + ds_read_b128, 8 ahead");
    run<1, 3, 8, 0>("MFMA + 2 ds_read_b64, 8 ahead");
    run<1, 1, 8, 2>("MFMA + ds_read_b128 (8 ahead) + 2 v_fma");
    run<2, 0, 4, 0>("MFMA only");
    run<2, 1, 4, 0>("MFMA + ds_read_b128, 4 ahead");
    run<2, 1, 8, 2>("MFMA + ds_read_b128 (8 ahead) + 2 v_fma");
    return 0;
}
