// K3 (ra_k3.hpp) for bfloat16 operands (cfg.mlp_dtype = 'bf16': three mantissa bits fewer, same MFMA rate).
#include "ra_k3.hpp"
void launch_mlp_sdf_stream_bf16(const GeoNet& net, const void* sarena, const void* sarena_pairs, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots,
                                hipStream_t stream, int grid_slots) {
    launch_k3<bf16>(net, sarena, sarena_pairs, barena, fr, io, max_slots, stream, grid_slots);
}
