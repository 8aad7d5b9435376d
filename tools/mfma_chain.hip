// MFMA issue rate vs accumulator dependence and waves per SIMD: v_mfma_f32_32x32x16_f16, NACC independent accumulators
// used round-robin, NT threads per workgroup (one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * ((threadIdx.x * 7 + i * 3) % 13 - 6) * 0.01f); b[i] = (_Float16)(seed * ((threadIdx.x * 5 + i) % 11 - 5) * 0.02f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC, int NT>
void run() {
    float* d; hipMalloc(&d, 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, NT>), dim3(256), dim3(NT), 0, 0, d, 200, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NT>), dim3(256), dim3(NT), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)iters * 16 * (NT / 64) * 256;
    printf("accumulators %d, waves/SIMD %d: %.0f TF, %.1f ns per MFMA per wave\n", NACC, NT / 256, mfma * 32768 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 16.0));
    hipFree(d);
}

int main() {
    run<1, 256>(); run<2, 256>(); run<4, 256>();
    run<1, 512>(); run<2, 512>(); run<4, 512>();
    return 0;
}
