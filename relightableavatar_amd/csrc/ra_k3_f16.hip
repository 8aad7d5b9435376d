// K3 (ra_k3.hpp) for IEEE half operands: the production type.
#include "ra_k3.hpp"
void launch_mlp_sdf_stream_f16(const GeoNet& net, const void* sarena, const void* sarena_pairs, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots,
                               hipStream_t stream, int grid_slots) {
    launch_k3<f16>(net, sarena, sarena_pairs, barena, fr, io, max_slots, stream, grid_slots);
}
