// K4 forward with tape (ra_k4.hpp) for bfloat16 operands (cfg.mlp_dtype = 'bf16').
#include "ra_k4.hpp"
void launch_mlp_fwd_tape_bf16(const GeoNet& net, const void* fwd_arena, const float* barena, const FrameState& fr, const FullIO& io, char* tape, hipStream_t stream) {
    launch_k4_fwd<bf16>(net, fwd_arena, barena, fr, io, tape, stream);
}
