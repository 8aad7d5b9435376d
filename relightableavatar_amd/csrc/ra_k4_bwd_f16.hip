// K4 reverse-mode backward + heads (ra_k4.hpp) for IEEE half operands: the production type.
#include "ra_k4.hpp"
void launch_mlp_bwd_heads_f16(const MatNet& mat, const ColNet& col, const void* bwd_arena, const float* barena, const float* shead_row, const FrameState& fr,
                               const FullIO& io, const char* tape, hipStream_t stream) {
    launch_k4_bwd<f16>(mat, col, bwd_arena, barena, shead_row, fr, io, tape, stream);
}
