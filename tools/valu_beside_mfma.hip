// How many VALU ops of each kind fit beside one v_mfma_f32_32x32x16_f16 per wave, 2 waves per SIMD (8 waves per CU)?
// Each variant: loop of {1 MFMA + K ops of kind T on independent registers}; prints ns per MFMA slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KIND, int K>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float seed) {
    f32x16 acc0 = {}, acc1 = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * (threadIdx.x % 7 + i) * 0.01f); b[i] = (_Float16)(seed * (threadIdx.x % 5 + i) * 0.02f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (threadIdx.x + i) * 1e-3f + 0.5f;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float& x = v[j % 8];
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
            if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            if (KIND == 2) asm volatile("v_log_f32 %0, %0" : "+v"(x));
            if (KIND == 3) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x));
            if (KIND == 4) { f32x2 p = {v[(2 * j) % 8], v[(2 * j + 1) % 8]}; asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p)); v[(2 * j) % 8] = p[0]; v[(2 * j + 1) % 8] = p[1]; }
            if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(x));
            if (KIND == 6) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x));
        }
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float& x = v[(j + 4) % 8];
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
            if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            if (KIND == 2) asm volatile("v_log_f32 %0, %0" : "+v"(x));
            if (KIND == 3) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x));
            if (KIND == 4) { f32x2 p = {v[(2 * j) % 8], v[(2 * j + 1) % 8]}; asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p)); v[(2 * j) % 8] = p[0]; v[(2 * j + 1) % 8] = p[1]; }
            if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(x));
            if (KIND == 6) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int K>
void run(const char* name) {
    float* d; hipMalloc(&d, 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(512), 0, 0, d, 1000, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(512), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x 2 MFMAs per iteration
    const double ns_per_mfma = ms * 1e6 / (iters * 4.0);
    printf("%-14s K=%2d: %.2f ns per MFMA (SIMD view), %.0f TF\n", name, K, ns_per_mfma, 32768.0 * 1024 * iters * 4 / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main() {
    run<0, 0>("mfma only");
    run<0, 2>("v_fma"); run<0, 4>("v_fma"); run<0, 6>("v_fma"); run<0, 8>("v_fma");
    run<1, 1>("v_exp"); run<1, 2>("v_exp"); run<1, 3>("v_exp"); run<1, 4>("v_exp");
    run<2, 1>("v_log"); run<2, 2>("v_log");
    run<3, 4>("v_max"); run<3, 8>("v_max");
    run<4, 2>("v_pk_fma"); run<4, 4>("v_pk_fma");
    run<5, 4>("v_cvt_pk"); run<6, 4>("v_add");
    return 0;
}
