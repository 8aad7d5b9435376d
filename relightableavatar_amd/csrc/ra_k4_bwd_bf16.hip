// K4 reverse-mode backward + heads (ra_k4.hpp) for bfloat16 operands (cfg.mlp_dtype = 'bf16').
#include "ra_k4.hpp"
void launch_mlp_bwd_heads_bf16(const MatNet& mat, const ColNet& col, const void* bwd_arena, const float* barena, const float* shead_row, const FrameState& fr,
                               const FullIO& io, const char* tape, hipStream_t stream) {
    launch_k4_bwd<bf16>(mat, col, bwd_arena, barena, shead_row, fr, io, tape, stream);
}
