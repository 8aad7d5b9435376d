// K3, 64 points per wave, IEEE half operands (ra_k3w.hpp)
#include "ra_k3w.hpp"
void launch_mlp_sdf_stream64_f16(const GeoNet& net, const void* sarena, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream) {
    if (max_slots <= 0) return;
    const int tiles = (max_slots + 255) / 256;
    hipLaunchKernelGGL((mlp_sdf_stream64_kernel<f16>), dim3(tiles < 256 ? tiles : 256), dim3(256), 0, stream, net, sarena, barena, fr, io);
}
