// Shared declarations of the gfx950 render hot path (see include/relightableavatar.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/relightableavatar.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---------------------------------------------------------------------------------------------
// Network description — built by ra_pack.cpp.  The 16-bit weights live in the streams the kernels consume in order
// (ra_pack.cpp StreamBuilder: forward stream of K3 / K4, backward stream of K4); a layer only records where its fp32 bias
// row starts in the bias arena (256 floats for a wide layer, 32 for a head) and its K depth in 16-wide k-steps.
// ---------------------------------------------------------------------------------------------
struct WideLayer { uint32_t ks; uint32_t bias; };

struct GeoNet {             // residual deformation + signed distance networks
    // residual net (base_network.py:14-42): L0 K=64 (PE10 padded), L1-3, L4 = L4a (K=256) + L4b (K=64, PE skip), L5-7, head 3
    WideLayer r[8];         // r[0]..r[7] (r[4] = the 256-wide part of the skip layer)
    WideLayer r4b;          // PE part of the skip layer (bias unused)
    WideLayer rhead;        // 256 -> 3 (rows padded to 32)
    // sdf net (net_utils.py:1276-1352): l0 K=64 (PE8 padded), l1,l2, l3 (205 rows + 51 zero rows), l4..l7 (1/sqrt2 folded into l4), l8
    WideLayer s[8];         // s[4] = the 205-wide part of the skip layer (cols 205..255 zero)
    WideLayer s4b;          // [PE8 51 | lo 9 | pad 4] part of the skip layer, 1/sqrt2 folded (bias unused)
    WideLayer shead;        // row 0 = sdf
    WideLayer sfeat;        // rows = feat 0..255 (lin8 rows 1..256)
    // per-frame folded biases for r[0] and r[4] live in frame state (cond folded in)
};

struct MatNet {             // albedo + roughness heads fused block-diagonally (relight_network.py:45-47)
    WideLayer m0;           // 256 -> [128 albedo | 128 roughness]
    WideLayer m1;           // block diagonal 256 -> 256
    WideLayer mhead;        // rows 0..2 albedo (cols 0..127), row 3 roughness (cols 128..255)
};

struct ColNet {             // RenderNetwork (base_network.py:132-171)
    WideLayer c0a;          // feat part of l0 (K=256)
    WideLayer c0b;          // [PE4(view) 27, norm 3, pad 2] part of l0 (K=32)
    WideLayer c1, c2, c3;   // c3's cond slice folded into a per-frame bias
    WideLayer chead;        // 256 -> 3
};

struct FrameState {         // device pointers owned by the ctx
    float* R;        // 9
    float* Th;       // 3
    float* vertA;    // n_verts x 24: per-vertex blended (A | big_A) rows 0..2 (3x4 each)
    float4* pverts4; // n_verts (x,y,z,0)
    float* pnorm;    // n_verts x 3
    float* tverts;   // n_verts x 3
    float* bias_r0;  // 256: b_r0 + W_r0[:, 63:219] cond
    float* bias_r4;  // 256
    float* bias_c3;  // 256 (colour net, cond_fix)
    int n_verts;
    // exact 3-NN acceleration: Morton-sorted vertices, boxes of 32-point leaves and of 8-leaf groups
    const float* bvh_soa;     // Morton-sorted vertices per 32-point leaf as x[32] | y[32] | z[32] | id[32] (id = original vertex index, int bits)
    const float4* bvh_sbox;   // per super box: lo, hi
    const float* bvh_lpair;   // per super box: its 8 leaf boxes as 4 pair records of 12 floats (behind bvh_sbox in the same buffer)
    int bvh_leaves;           // 0 -> brute force
    int bvh_supers;
};

struct DevCounters {       // device-side work counters (ra_get_counters)
    unsigned long long n_coarse, n_fine_sdf, n_fine_full, n_shadow_rays, n_hit_pixels;
    unsigned long long n_fine_sdf_wide;      // the part of n_fine_sdf that went through the 8-wave K3 (launches that fill the chip)
    unsigned long long n_fine_sdf_comp;      // the part of n_fine_sdf answered in compensated arithmetic (K3C: the surface trace)
};

struct MlpIO {
    const float* bpts;      // n_slots x 3 big-pose points (compacted)
    const int* idx;         // n_slots: point index of each slot
    const int* count;       // device: number of valid slots
    float* sdf;             // per point: in = coarse smpl sdf, out = blended HDQ sdf
    float dist_th;
    int smooth;
    float resd_limit;
    DevCounters* counters;  // nullable
};

struct FullIO {             // the full query: geometry with normal + material / colour heads
    const float* bpts;      // n_slots x 3
    const float* mats;      // n_slots x 24 (A_bw rows | big_A_bw rows), nullable -> identity
    const float* view;      // n_points x 3 world view dirs (AniSDF), nullable
    const int* idx;
    const int* count;       // device: fine slots of the whole compacted list
    int slot0, slot_cap;    // this launch pair handles slots [slot0, min(slot0 + slot_cap, *count)) — the tape is sized for slot_cap
    float* raw;             // n_points x C, scattered by idx
    int C;
    float beta;             // clamp(_beta, 1e-9, 1e6)
    float resd_limit;
    float albedo_slope, albedo_bias, rough_slope, rough_bias;
    int relight;
    // debug (nullable)
    float* dbg_grad;        // n_slots x 3  d sdf / d bpts
    float* dbg_feat;        // n_slots x 256
    float* dbg_sdf;         // n_slots
    float* dbg_resd;        // n_slots x 3
    float* dbg_gc;          // n_slots x 3  d sdf / d cpts
    float* dbg_pe;          // n_slots x 128 gradients wrt the encoding slots [sdf h0 | sdf h1 | resd h0 | resd h1] (debugging aid, pre-zeroed)
    int dbg_layer;          // debugging aid: which delta (B fragments) of the backward pass goes to dbg_pe (n_slots x 256), -1 = encoding slots
    DevCounters* counters;  // nullable
};

// K3 (ra_k3.hpp): HDQ fine distance query; activations stay in registers, weights stream through LDS (sarena: ra_pack.cpp StreamBuilder).
// One translation unit per operand type: IEEE half (production) and bfloat16.
// sarena_pairs: the same fragments with the row blocks of a layer interleaved in pairs (the 2- / 4-wave latency variants)
// K3 workgroup width by the launch's upper bound of fine points (ra_k3.hpp launch_k3): 2 / 4 waves (one per SIMD) for launches that
// cannot fill the 256 CUs with 256-point tiles, 8 waves otherwise
inline int k3_waves(int max_slots) { return max_slots <= 256 * 64 ? 2 : (max_slots <= 256 * 256 ? 4 : 8); }
// grid_slots (all fused distance launches): the size the GRID is made for when it is larger than max_slots, which then only picks the
// workgroup width.  max_slots may come from an earlier frame's count (launch-variant hints): if this frame's count is many times larger —
// a camera cut — a narrow variant on a grid sized for the hint would be a cliff (a few workgroups for millions of points); on a grid
// sized for a fraction of the bound it is only the narrow variant's lower rate.  Workgroups beyond the real count find no tile and exit.
inline int mlp_grid(int max_slots, int grid_slots, int tile) {
    const int m = grid_slots > max_slots ? grid_slots : max_slots;
    const int tiles = (m + tile - 1) / tile;
    return tiles < 256 ? tiles : 256;     // one workgroup per CU (the weight ring fills its LDS), persistent over tiles
}
void launch_mlp_sdf_stream_f16(const GeoNet& net, const void* sarena, const void* sarena_pairs, const float* barena, const FrameState& fr, const MlpIO& io,
                               int max_slots, hipStream_t stream, int grid_slots = 0);
void launch_mlp_sdf_stream_bf16(const GeoNet& net, const void* sarena, const void* sarena_pairs, const float* barena, const FrameState& fr, const MlpIO& io,
                                int max_slots, hipStream_t stream, int grid_slots = 0);

// K3C (ra_k3c.hpp): the same query in compensated arithmetic (f16 hi + lo operand pairs, three MFMAs per k-step) on the split stream
// sarena_c, 16 points per wave: 4 waves per workgroup (64-point tiles, one wave per SIMD) for launches of at most 16 Ki points, else 8
// (128-point tiles, two waves per SIMD).
// K3CC (ra_k3cc.hpp): launches of at most k3c_coop_max points — an upper bound; the surface loop fills about half of it, one round of 256
// tiles — give every 16-point tile to FOUR cooperating waves (a quarter of each layer's row blocks per wave, private register-resident
// weight streams): 102 -> 56 us per launch.  All variants are bit-identical.
inline int k3c_waves(int max_slots) { return max_slots <= 256 * 64 ? 4 : 8; }
constexpr int k3c_coop_max = 256 * 32;
void launch_mlp_sdf_coop(const GeoNet& net, const void* sarena_c, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream,
                         int grid_slots = 0);
// allow_coop = false: launches of at most k3c_coop_max points take K3C's 4-wave tiles instead of K3CC (the context's self-test of K3CC
// failed at ra_finalize_weights: ra_ctx::k3cc_ok)
void launch_mlp_sdf_comp(const GeoNet& net, const void* sarena_c, const float* barena, const FrameState& fr, const MlpIO& io, int max_slots, hipStream_t stream,
                         bool allow_coop = true, int grid_slots = 0);

// K4 (ra_k4.hpp): forward with tape + reverse-mode backward + heads, on the sub-batch io.slot0 / io.slot_cap of the fine list
size_t mlp_full_rev_tape_bytes(int slots);
int mlp_full_rev_bwd_stages(int relight);       // 16-fragment stages of the backward stream the kernels are compiled for
void launch_mlp_fwd_tape_f16(const GeoNet& net, const void* fwd_arena, const float* barena, const FrameState& fr, const FullIO& io, char* tape, hipStream_t stream);
void launch_mlp_fwd_tape_bf16(const GeoNet& net, const void* fwd_arena, const float* barena, const FrameState& fr, const FullIO& io, char* tape, hipStream_t stream);
void launch_mlp_bwd_heads_f16(const MatNet& mat, const ColNet& col, const void* bwd_arena, const float* barena, const float* shead_row, const FrameState& fr,
                              const FullIO& io, const char* tape, hipStream_t stream);
void launch_mlp_bwd_heads_bf16(const MatNet& mat, const ColNet& col, const void* bwd_arena, const float* barena, const float* shead_row, const FrameState& fr,
                               const FullIO& io, const char* tape, hipStream_t stream);

// --- error plumbing ---------------------------------------------------------------------------
void ra_set_error(const std::string& msg);
#define RA_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ra_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
