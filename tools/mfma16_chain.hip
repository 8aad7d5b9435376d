// Issue rate of v_mfma_f32_16x16x32_f16 for ONE wave per SIMD (and two): NACC independent accumulators used round-robin, optionally one
// ds_read_b128 pair per three MFMAs (the compensated kernels' k-step: tools -> DESIGN.md section 8 item 5).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, int NT, bool LDS>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float seed) {
    __shared__ f16x8 sm[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += NT) for (int j = 0; j < 8; ++j) sm[i][j] = (_Float16)(0.001f * ((i + j) % 17));
    __syncthreads();
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 4; ++i) acc[a][i] = 0.f;
    f16x8 a, b, a2;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * ((threadIdx.x * 7 + i * 3) % 13 - 6) * 0.01f); b[i] = (_Float16)(seed * ((threadIdx.x * 5 + i) % 11 - 5) * 0.02f); }
    a2 = a;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 48; ++j) {
            if (LDS && j % 3 == 0) { a = sm[((it + j) & 63) * 64 + lane]; a2 = sm[((it + j + 7) & 63) * 64 + lane]; }
            acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16((j % 3 == 1) ? a2 : a, b, acc[j % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) for (int i = 0; i < 4; ++i) s += acc[q][i];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC, int NT, bool LDS>
void run() {
    float* d; hipMalloc(&d, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, NT, LDS>), dim3(256), dim3(NT), 0, 0, d, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NT, LDS>), dim3(256), dim3(NT), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 48.0);         // ns per MFMA per wave
    printf("accumulators %d, waves/SIMD %d, lds reads %d: %.2f ns per MFMA per wave (%.1f cycles at 2.4 GHz), %.2f ns per MFMA per SIMD\n", NACC, NT / 256, (int)LDS, per, per * 2.4,
           per / (NT / 256));
}

int main() {
    run<1, 256, false>(); run<2, 256, false>(); run<4, 256, false>();
    run<1, 512, false>(); run<2, 512, false>();
    run<1, 256, true>(); run<2, 256, true>(); run<1, 512, true>(); run<2, 512, true>();
    return 0;
}
