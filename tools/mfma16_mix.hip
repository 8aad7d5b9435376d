// What raises a lone wave's 16x16x32 MFMA period from 19 to 32+ cycles in the compensated kernels?  One dependent chain per wave, 3 MFMAs per
// k-step, variants: A-fragment ds_read_b128 pair per k-step read AHEAD k-steps before use (ring of registers), V VALU ops per MFMA slot
// (independent fma chain), sched_barrier(0) per slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int AHEAD, int V, bool SB, int NT, int NACC = 1>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float seed) {
    __shared__ f16x8 sm[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += NT) for (int j = 0; j < 8; ++j) sm[i][j] = (_Float16)(0.001f * ((i + j) % 17));
    __syncthreads();
    f32x4 accs[NACC];
    for (int a = 0; a < NACC; ++a) accs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(seed * ((threadIdx.x * 5 + i) % 11 - 5) * 0.02f);
    const int lane = threadIdx.x & 63;
    constexpr int R = AHEAD > 0 ? AHEAD : 1;
    f16x8 ah[R], al[R];
    for (int r = 0; r < R; ++r) { ah[r] = sm[r * 64 + lane]; al[r] = sm[(r + 9) * 64 + lane]; }
    float v[4] = {seed, seed * 2, seed * 3, seed * 4};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            f16x8 h, l;
            if (AHEAD == 0) { h = sm[((it + ks) & 63) * 64 + lane]; l = sm[((it + ks + 7) & 63) * 64 + lane]; }
            else { h = ah[ks % R]; l = al[ks % R]; }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                accs[(ks * 3 + m) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(m == 1 ? l : h, b, accs[(ks * 3 + m) % NACC], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < V; ++q) v[(m + q) & 3] = __builtin_fmaf(v[(m + q) & 3], 1.0001f, 0.5f);
                if (m == 2 && AHEAD > 0) { ah[ks % R] = sm[((it + ks + 3) & 63) * 64 + lane]; al[ks % R] = sm[((it + ks + 11) & 63) * 64 + lane]; }
                if (SB) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int a = 0; a < NACC; ++a) s += accs[a][0] + accs[a][1] + accs[a][2] + accs[a][3];
    if (s == 12345.678f) out[0] = s;
}

template <int AHEAD, int V, bool SB, int NT, int NACC = 1>
void run() {
    float* d; (void)hipMalloc(&d, 4);
    const int iters = 3000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<AHEAD, V, SB, NT, NACC>), dim3(256), dim3(NT), 0, 0, d, 100, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<AHEAD, V, SB, NT, NACC>), dim3(256), dim3(NT), 0, 0, d, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 48.0);
    printf("reads ahead %d k-steps, %d VALU per slot, sched_barrier %d, waves/SIMD %d, chains %d: %.2f ns per MFMA per wave (%.1f cycles at 2.4 GHz)\n", AHEAD, V, (int)SB, NT / 256, NACC, per, per * 2.4);
}

int main() {
    run<0, 0, false, 256>(); run<2, 0, false, 256>(); run<4, 0, false, 256>(); run<4, 0, true, 256>();
    run<4, 1, true, 256>(); run<4, 2, true, 256>(); run<4, 2, false, 256>(); run<4, 4, true, 256>();
    run<4, 2, true, 512>(); run<4, 0, true, 512>();
    run<4, 2, true, 256, 2>(); run<4, 2, true, 256, 3>(); run<4, 1, true, 256, 2>(); run<4, 2, true, 512, 2>(); run<4, 2, false, 256, 2>();
    return 0;
}
