// K3C (ra_k3c.hpp): the compensated distance query; IEEE-half hi + lo operand pairs whatever cfg.mlp_dtype says.
#include "ra_k3c.hpp"
void launch_mlp_sdf_comp(const GeoNet& net, const void* sarena_c, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream) {
    if (max_slots <= 0) return;
    const int nw = k3c_waves(max_slots);
    if (nw == 2) launch_c_nw<2>(net, sarena_c, barena, fr, io, max_slots, stream);
    else if (nw == 4) launch_c_nw<4>(net, sarena_c, barena, fr, io, max_slots, stream);
    else launch_c_nw<8>(net, sarena_c, barena, fr, io, max_slots, stream);
}
