// positional-encoding B fragments of the distance-query kernels (ra_k3.hpp)
#pragma once
#include "ra_stream.hpp"

namespace {

// encoding B fragments of one point (lane half h): see pe_chan_resd / pe_chan_sdf in ra_pack.cpp
template <typename E, int L, bool LO>
__device__ __forceinline__ void pe_frags(u32x4 (&Bp)[4], const float (&x)[3], int h) {
    float rev[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rev[c] = x[c] * INV_2PI;
    float v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q < 3 * L) {
            const float a = rev[q % 3] * (float)(1 << (q / 3));
            const float sv = __builtin_amdgcn_sinf(a), cv = __builtin_amdgcn_cosf(a);
            v[q] = h ? cv : sv;
        } else if (!LO) {
            v[q] = (q == 3 * L) ? (h ? x[1] : x[0]) : ((q == 3 * L + 1) ? (h ? 0.f : x[2]) : 0.f);
        } else {
            const int r = q - 3 * L;
            if (r < 3) {
                const float hi = (float)(E)x[r];
                v[q] = h ? x[r] - hi : hi;
            } else if (r < 6) {
                const float sv = __builtin_amdgcn_sinf(rev[r - 3]), cv = __builtin_amdgcn_cosf(rev[r - 3]);
                v[q] = h ? cv - (float)(E)cv : sv - (float)(E)sv;
            } else {
                v[q] = 0.f;
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int w = 0; w < 4; ++w) Bp[ks][w] = pack2<E>(v[8 * ks + 2 * w], v[8 * ks + 2 * w + 1]);
}

}  // namespace
