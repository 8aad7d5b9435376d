// K3CC (ra_k3cc.hpp): the compensated distance query with four waves sharing a 16-point tile.
#include "ra_k3cc.hpp"
void launch_mlp_sdf_coop(const GeoNet& net, const void* sarena_c, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream, int grid_slots) {
    if (max_slots <= 0) return;
    launch_coop(net, sarena_c, barena, fr, io, max_slots, stream, grid_slots);
}
