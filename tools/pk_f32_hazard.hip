// Stand-alone probe of round 5's finding: does a compiler-style packed-fp32 chain (v_pk_mul_f32 with op_sel swizzles ->
// v_pk_fma_f32 -> v_pk_fma_f32 op_sel_hi -> v_pk_add_f32: the instructions SLP formed in lbs_verts_kernel) return wrong halves when its
// waves share SIMDs with another kernel's MFMA stream?  The victim kernel computes the chain with the packed instructions (inline asm: the
// exact forms) AND with scalar v_mul / v_fma, compares bit for bit and counts mismatches per component; the aggressor is a persistent MFMA
// loop, one wave per SIMD (140 KB of LDS per workgroup keeps it at one workgroup per CU and leaves registers for the victim's waves).
//   hipcc --offload-arch=gfx950 -O3 tools/pk_f32_hazard.hip -o tools/pk_f32_hazard.bin && tools/pk_f32_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// FORM 0: the chain SLP formed in lbs_verts_kernel; 1: the same chain without any op_sel; 2: only the swizzled v_pk_mul_f32; 3: only the
// v_pk_fma_f32 with op_sel_hi:[1,0,1]; 4: only a plain v_pk_fma_f32; 5: v_pk_mul_f32 of a VGPR pair with an SGPR pair, op_sel_hi:[1,0]
// (a form of the coarse level's hand-written scan)
template <int FORM>
__global__ void victim(const float* __restrict__ in, int n, unsigned* bad_lo, unsigned* bad_hi, unsigned* first_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = in + (size_t)i * 12;
    const f2 sg = {in[0], in[1]};                    // wave-uniform: lives in an SGPR pair
    for (int rep = 0; rep < 64; ++rep) {
        f2 a = {p[0], p[1]}, b = {p[2], p[3]}, c = {p[4], p[5]}, d = {p[6], p[7]}, e = {p[8] + rep, p[9]}, f = {p[10], p[11]};
        f2 t, u, r;
        float rl, rh;
        if (FORM == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(u) : "v"(c), "v"(b), "v"(t));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(d), "v"(e), "v"(u));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(f), "v"(r));
            const float tl = a.x * b.y, th = a.y * b.x;
            const float ul = __builtin_fmaf(c.x, b.x, tl), uh = __builtin_fmaf(c.y, b.y, th);
            rl = f.x + __builtin_fmaf(d.x, e.x, ul); rh = f.y + __builtin_fmaf(d.y, e.x, uh);
        } else if (FORM == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(u) : "v"(c), "v"(b), "v"(t));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(d), "v"(e), "v"(u));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(f), "v"(r));
            const float tl = a.x * b.x, th = a.y * b.y;
            const float ul = __builtin_fmaf(c.x, b.x, tl), uh = __builtin_fmaf(c.y, b.y, th);
            rl = f.x + __builtin_fmaf(d.x, e.x, ul); rh = f.y + __builtin_fmaf(d.y, e.y, uh);
        } else if (FORM == 2) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(e));
            rl = a.x * e.y; rh = a.y * e.x;
        } else if (FORM == 3) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(d), "v"(e), "v"(c));
            rl = __builtin_fmaf(d.x, e.x, c.x); rh = __builtin_fmaf(d.y, e.x, c.y);
        } else if (FORM == 4) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(d), "v"(e), "v"(c));
            rl = __builtin_fmaf(d.x, e.x, c.x); rh = __builtin_fmaf(d.y, e.y, c.y);
        } else if (FORM == 5) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(e), "s"(sg));
            rl = e.x * sg.x; rh = e.y * sg.x;
        } else if (FORM == 6) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(e));
            rl = a.y + e.x; rh = a.x + e.y;
        } else if (FORM == 7) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(d), "v"(e), "v"(c));
            rl = __builtin_fmaf(d.x, e.x, c.y); rh = __builtin_fmaf(d.y, e.y, c.x);
        } else if (FORM == 8) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(e));
            rl = a.y * e.y; rh = a.x * e.x;
        } else {
            // packed f16 with the same swizzle (one dword per operand): compared as raw bits with the scalar f16 products
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 ha = {(_Float16)a.x, (_Float16)a.y}, hb = {(_Float16)e.x, (_Float16)e.y}, hr;
            asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(hr) : "v"(ha), "v"(hb));
            const _Float16 l = ha.x * hb.y, hgh = ha.y * hb.x;
            r.x = (float)hr.x; r.y = (float)hr.y; rl = (float)l; rh = (float)hgh;
        }
        if (__float_as_uint(r.x) != __float_as_uint(rl)) { atomicAdd(bad_lo, 1u); atomicMin(first_bad, (unsigned)i); }
        if (__float_as_uint(r.y) != __float_as_uint(rh)) { atomicAdd(bad_hi, 1u); atomicMin(first_bad, (unsigned)i); }
    }
}

__global__ __launch_bounds__(256) void aggressor(float* out, const int* stop, int max_iters) {
    __shared__ float hog[35 * 1024];                 // 140 KB: one workgroup per CU, its four waves on the four SIMDs
    hog[threadIdx.x] = 0.f;
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (k + 1)); }
    f16v acc = {};
    for (int it = 0; it < max_iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if ((it & 63) == 0 && *(volatile const int*)stop) break;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + hog[threadIdx.x];
}

int main() {
    const int n = 1 << 16;
    std::vector<float> h((size_t)n * 12);
    unsigned s = 1u;
    for (float& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.f / 16777216.f) - 0.5f) * 2.f; }
    float *din, *dout; unsigned* cnt; int* stop;
    (void)hipMalloc(&din, h.size() * 4); (void)hipMalloc(&dout, 256 * 256 * 4); (void)hipMalloc(&cnt, 12); (void)hipMalloc(&stop, 4);
    (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipStream_t sv, sa; (void)hipStreamCreateWithFlags(&sv, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    const char* names[10] = {"0: SLP's chain (mul op_sel:[0,1] op_sel_hi:[1,0] -> fma -> fma op_sel_hi:[1,0,1] -> add)", "1: the chain without op_sel",
                            "2: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] alone", "3: v_pk_fma_f32 op_sel_hi:[1,0,1] alone", "4: plain v_pk_fma_f32 alone",
                            "5: v_pk_mul_f32 v, s op_sel_hi:[1,0] (coarse-level form)", "6: v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]",
                            "7: v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0] (only the addend swizzled)", "8: v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0] (both halves swapped)",
                            "9: v_pk_mul_f16 op_sel:[0,1] op_sel_hi:[1,0] (packed f16)"};
    for (int form = 0; form < 10; ++form)
        for (int with = 0; with < 2; ++with) {
            unsigned init[3] = {0, 0, 0xffffffffu};
            (void)hipMemcpy(cnt, init, 12, hipMemcpyHostToDevice);
            int zero = 0; (void)hipMemcpy(stop, &zero, 4, hipMemcpyHostToDevice);
            if (with) hipLaunchKernelGGL(aggressor, dim3(256), dim3(256), 0, sa, dout, stop, 1 << 22);
            for (int launch = 0; launch < 400; ++launch) {
                switch (form) {
                    case 0: hipLaunchKernelGGL(victim<0>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 1: hipLaunchKernelGGL(victim<1>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 2: hipLaunchKernelGGL(victim<2>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 3: hipLaunchKernelGGL(victim<3>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 4: hipLaunchKernelGGL(victim<4>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 5: hipLaunchKernelGGL(victim<5>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 6: hipLaunchKernelGGL(victim<6>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 7: hipLaunchKernelGGL(victim<7>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    case 8: hipLaunchKernelGGL(victim<8>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                    default: hipLaunchKernelGGL(victim<9>, dim3(n / 256), dim3(256), 0, sv, din, n, cnt, cnt + 1, cnt + 2); break;
                }
            }
            (void)hipStreamSynchronize(sv);
            int one = 1; (void)hipMemcpy(stop, &one, 4, hipMemcpyHostToDevice);
            (void)hipStreamSynchronize(sa);
            unsigned res[3]; (void)hipMemcpy(res, cnt, 12, hipMemcpyDeviceToHost);
            printf("form %-100s %s MFMAs: low-half mismatches %9u, high-half %9u of %lld (first thread %d)\n", names[form], with ? "BESIDE " : "without",
                   res[0], res[1], 400ll * n * 64, (int)res[2]);
        }
    return 0;
}
