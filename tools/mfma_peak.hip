// Measures the sustained dense f16 MFMA rate of this GPU (v_mfma_f32_32x32x16_f16, register operands only):
// the practical ceiling the fused MLP kernel is compared against.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x * 3 + e)); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* name) {
    float* d; hipMalloc(&d, blocks * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(d, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * (double)NACC * iters * blocks * 4;
    printf("%s: %d blocks x 4 waves, %d accumulators: %.3f ms, %.0f TFLOP/s\n", name, blocks, NACC, ms, flop / ms * 1e-9);
    hipFree(d);
}
int main() { run<8>(256, "1 wave/SIMD"); run<8>(512, "2 waves/SIMD"); run<4>(1024, "4 waves/SIMD"); run<2>(512, "2 waves/SIMD, 2 acc"); run<1>(512, "2 waves/SIMD, 1 acc"); run<2>(256, "1 wave/SIMD, 2 acc"); run<8>(512, "2 waves/SIMD (again)"); return 0; }
