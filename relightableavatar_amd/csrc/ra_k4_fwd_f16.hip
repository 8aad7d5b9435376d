// K4 forward with tape (ra_k4.hpp) for IEEE half operands: the production type.
#include "ra_k4.hpp"
size_t mlp_full_rev_tape_bytes(int slots) { return k4_tape_bytes(slots); }
int mlp_full_rev_bwd_stages(int relight) { return relight ? BW_STAGES_RELIGHT : BW_STAGES_ANISDF; }
void launch_mlp_fwd_tape_f16(const GeoNet& net, const void* fwd_arena, const float* barena, const FrameState& fr, const FullIO& io, char* tape, hipStream_t stream) {
    launch_k4_fwd<f16>(net, fwd_arena, barena, fr, io, tape, stream);
}
