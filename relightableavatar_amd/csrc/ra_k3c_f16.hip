// K3C (ra_k3c.hpp): the compensated distance query; IEEE-half hi + lo operand pairs whatever cfg.mlp_dtype says.
// Launches of at most k3c_coop_max points go to K3CC (ra_k3cc.hpp, its own translation unit): four waves sharing a 16-point tile.
#include "ra_k3c.hpp"
#ifdef RA_TESTING
#include <cstdlib>
#endif
void launch_mlp_sdf_comp(const GeoNet& net, const void* sarena_c, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream,
                         bool allow_coop, int grid_slots) {
    if (max_slots <= 0) return;
    int coop_max = allow_coop ? k3c_coop_max : 0;
#ifdef RA_TESTING            // test / experiment builds only (tools/build_variant.sh, tools/ab_coop.sh): RA_K3C_COOP_MAX=0 restores the 2-wave tiles
    static const int force = getenv("RA_K3C_COOP_MAX") ? atoi(getenv("RA_K3C_COOP_MAX")) : -1;
    if (force >= 0) coop_max = force;
    if (max_slots > coop_max && max_slots <= 256 * 32) { launch_c_nw<2>(net, sarena_c, barena, fr, io, max_slots, stream, grid_slots); return; }
#endif
    if (max_slots <= coop_max) { launch_mlp_sdf_coop(net, sarena_c, barena, fr, io, max_slots, stream, grid_slots); return; }
    if (k3c_waves(max_slots) == 4) launch_c_nw<4>(net, sarena_c, barena, fr, io, max_slots, stream, grid_slots);
    else launch_c_nw<8>(net, sarena_c, barena, fr, io, max_slots, stream, grid_slots);
}
