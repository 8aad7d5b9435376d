// sanity check of the LDS-DMA mechanics the streamed K3 relies on: global_load_lds_dwordx4 into LDS offsets
// above 64 KB, counted vmcnt, raw s_barrier, then ds_read_b128 of the landed fragments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int stages) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 8 stages x 16 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring = (unsigned)(size_t)smem;
    const char* g = reinterpret_cast<const char*>(src) + wave * 2048 + lane * 16;
    auto issue = [&](int st) {
        const unsigned slot = st & 7;
        glds16(g + (size_t)st * 16384, ring + slot * 16384 + wave * 2048);
        glds16(g + (size_t)st * 16384 + 1024, ring + slot * 16384 + wave * 2048 + 1024);
    };
    for (int st = 0; st < 7; ++st) issue(st);
    for (int st = 0; st < stages; ++st) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(st + 7 < stages ? st + 7 : st);      // keep the in-flight count constant (tail re-loads the same stage: harmless)
        const u32x4* rp = reinterpret_cast<const u32x4*>(smem + (st & 7) * 16384);
        // every wave reads all 16 fragments; wave w writes fragments 2w, 2w+1 back out
        u32x4 acc = {0, 0, 0, 0};
        for (int f = 0; f < 16; ++f) { const u32x4 v = rp[f * 64 + lane]; acc += v;
            if (f == 2 * wave || f == 2 * wave + 1) dst[((size_t)st * 16 + f) * 64 + lane] = v; }
        if (acc.x == 0xdeadbeef) dst[0] = acc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

int main() {
    const int stages = 122;
    const size_t n = (size_t)stages * 1024;   // u32x4 elements
    std::vector<unsigned> h(n * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    u32x4 *s, *d;
    hipMalloc(&s, n * 16); hipMalloc(&d, n * 16);
    hipMemcpy(s, h.data(), n * 16, hipMemcpyHostToDevice);
    hipMemset(d, 0, n * 16);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384 + 16384);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 8 * 16384 + 16384, 0, s, d, stages);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned> o(n * 4);
    hipMemcpy(o.data(), d, n * 16, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < o.size(); ++i) bad += o[i] != h[i];
    printf("glds_test: %s, mismatches %zu of %zu\n", hipGetErrorString(e), bad, o.size());
    return bad != 0;
}
