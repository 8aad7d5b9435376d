// N4, third item: the map -> image normalisations of the reference visualiser
// (lib/visualizers/base_visualizer.py:54-201, Visualizer.generate_image): per output type a normalisation of the rendered map
// (percentile stretch for depth / shading / specular / residual, camera-space normals, big-pose-box coordinates, sRGB albedo),
// then the scatter of the in-box rays into the H x W image over cfg.bg_brightness and the alpha plane.
// HBM streaming (<= 28 B read + 16 B written per ray) plus one hipcub radix sort for the types that need a percentile.
#include "ra_kernels.hpp"
#include <hipcub/hipcub.hpp>

namespace {

constexpr int TPB = 256;
constexpr float PI_F = 3.14159265358979323846f;
inline dim3 grid_for(long long n) { return dim3((unsigned)((n + TPB - 1) / TPB)); }

__device__ __forceinline__ float srgb(float x) {                // relight_utils.py:179-192
    x = fminf(fmaxf(x, 0.f), 1.f);
    return (x <= 0.0031308f) ? x * 12.92f : 1.055f * powf(x + 1e-7f, 1.f / 2.4f) - (1.055f - 1.f);
}

__global__ void flags_kernel(const float* __restrict__ acc, int P, unsigned char* __restrict__ flag) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < P) flag[i] = acc[i] != 0.f;                          // acc_map.bool()
}
__global__ void diff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ o) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) o[i] = a[i] - b[i];
}
// torch.topk ranks every NaN above +inf whatever its sign bit (x86 produces 0/0 = -nan, the GPU +nan); the radix sort orders
// bit patterns, so NaNs are made the canonical positive quiet NaN first, and the unselected tail gets the largest pattern of all
__global__ void canon_nan_kernel(const float* __restrict__ v, long long n, float* __restrict__ o) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) { const float x = v[i]; o[i] = x != x ? __uint_as_float(0x7FC00000u) : x; }
}
__global__ void pad_tail_kernel(float* __restrict__ p, const int* __restrict__ cnt, int n) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n && i >= *cnt) p[i] = __uint_as_float(0x7FFFFFFFu);
}
// "a simple version of percentile" (:108-109): lo = the k-th smallest, hi = the k-th largest of the n sorted values.
// Deviation, on purpose: the Depth type selects among the rays with acc != 0, whose count is only known on the device.  Where
// fewer than k = int(0.01 * P) rays hit, the reference's topk(k) raises; checking that here would cost a read-back per image,
// so k is clamped to the count instead (a nearly empty frame is stretched between its own extremes; documented in
// include/relightableavatar.h and visualizers/base_visualizer.py)
__global__ void pick_kernel(const float* __restrict__ sorted, const int* __restrict__ n_dev, int n_host, int k, float* __restrict__ stats) {
    const int n = n_dev ? *n_dev : n_host;
    const int kk = k < 1 ? 1 : (k > n ? n : k);
    stats[0] = n > 0 ? sorted[kk - 1] : 0.f;
    stats[1] = n > 0 ? sorted[n - kk] : 1.f;
}

__global__ void compose_kernel(ImageJob j) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= j.P) return;
    const float acc = j.acc ? j.acc[i] : 1.f;
    float rgb[3];
    const float* a = j.a;
    switch (j.type) {
    case RA_IMG_SURFACE:                                                         // :150-154
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = acc * ((a[3 * i + c] - j.tbounds[c]) / (j.tbounds[3 + c] - j.tbounds[c]));
        break;
    case RA_IMG_RESIDUAL:                                                        // :156-165
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = acc * ((a[3 * i + c] - j.b[3 * i + c]) / j.stats[1]);
        break;
    case RA_IMG_DEPTH: {                                                         // :98-114
        const float lo = fminf(j.stats[0], j.min_clip), hi = j.stats[1];
        float v = (a[i] - lo) / (hi - lo);
        v = v != v ? v : fminf(fmaxf(v, 0.f), 1.f);                              // torch.clip keeps NaN (the d_x = 0 rays of quirk 4)
        rgb[0] = rgb[1] = rgb[2] = v;
        break;
    }
    case RA_IMG_ALPHA:                                                           // :93-96
        rgb[0] = rgb[1] = rgb[2] = acc;
        break;
    case RA_IMG_NORMAL: {                                                        // :57-66
        float n[3] = {a[3 * i], a[3 * i + 1], a[3 * i + 2]};
        const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) + 1e-8f;
        n[0] /= nn; n[1] /= nn; n[2] /= nn;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = n[0] * j.cam_R[3 * c] + n[1] * j.cam_R[3 * c + 1] + n[2] * j.cam_R[3 * c + 2];       // norm @ cam_R^T
            if (c > 0) v = -v;
            rgb[c] = (v * 0.5f + 0.5f) * acc;
        }
        break;
    }
    case RA_IMG_SPECULAR:
    case RA_IMG_SHADING:                                                         // :116-126, :176-186
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = j.normalize ? a[3 * i + c] / j.stats[1] : a[3 * i + c];
        break;
    case RA_IMG_ALBEDO:                                                          // :128-132
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = j.tonemap ? srgb(a[3 * i + c]) : a[3 * i + c];
        break;
    case RA_IMG_ROUGHNESS:                                                       // :134-136
        rgb[0] = rgb[1] = rgb[2] = a[i];
        break;
    default:                                                                     // RA_IMG_RENDERING :167-171
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = a[3 * i + c];
        break;
    }
    const long long p = j.pix ? j.pix[i] : i;
    j.image[3 * p] = rgb[0]; j.image[3 * p + 1] = rgb[1]; j.image[3 * p + 2] = rgb[2];
    if (j.alpha) j.alpha[p] = acc;
}

}  // namespace

size_t image_sort_temp_bytes(long long n) {
    size_t a = 0, b = 0;
    hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const float*)nullptr, (float*)nullptr, (int)n);
    hipcub::DeviceSelect::Flagged(nullptr, b, (const float*)nullptr, (const unsigned char*)nullptr, (float*)nullptr, (int*)nullptr, (int)n);
    return a > b ? a : b;
}

// stats[0], stats[1] <- k-th smallest / k-th largest of vals[0..n) (flag != nullptr: only where flag is set)
int launch_percentiles(const float* vals, long long n, const float* acc_flags, int k, float* scratch_a, float* scratch_b, unsigned char* flag,
                       int* count_dev, void* temp, size_t temp_bytes, float* stats, hipStream_t s) {
    const int* n_dev = nullptr;
    hipLaunchKernelGGL(canon_nan_kernel, grid_for(n), dim3(TPB), 0, s, vals, n, scratch_b);
    const float* keys = scratch_b;
    if (acc_flags) {
        hipLaunchKernelGGL(flags_kernel, grid_for(n), dim3(TPB), 0, s, acc_flags, (int)n, flag);
        if (hipcub::DeviceSelect::Flagged(temp, temp_bytes, scratch_b, flag, scratch_a, count_dev, (int)n, s) != hipSuccess) return 1;
        // the radix sort takes a host-side count: all n slots are sorted, the tail beyond the selected count padded with +inf
        // (no read-back of the count); pick_kernel indexes with the device count
        hipLaunchKernelGGL(pad_tail_kernel, grid_for(n), dim3(TPB), 0, s, scratch_a, count_dev, (int)n);
        keys = scratch_a;
        n_dev = count_dev;
    }
    float* sorted = keys == scratch_b ? scratch_a : scratch_b;
    if (hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, sorted, (int)n, 0, 32, s) != hipSuccess) return 1;
    hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(1), 0, s, sorted, n_dev, (int)n, k, stats);
    return 0;
}

void launch_diff(const float* a, const float* b, long long n, float* o, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(diff_kernel, grid_for(n), dim3(TPB), 0, s, a, b, n, o);
}

void launch_compose_image(const ImageJob& j, long long n_pixels, hipStream_t s) {
    launch_fill(j.image, (size_t)n_pixels * 3, j.bg, s);
    if (j.alpha) hipMemsetAsync(j.alpha, 0, (size_t)n_pixels * sizeof(float), s);
    if (j.P > 0) hipLaunchKernelGGL(compose_kernel, grid_for(j.P), dim3(TPB), 0, s, j);
}
