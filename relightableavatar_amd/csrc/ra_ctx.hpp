// The opaque context behind the C ABI: packed weights, frame state, scratch arenas, counters.
#pragma once
#include "ra_kernels.hpp"

struct HostNets {
    GeoNet geo{};
    MatNet mat{};
    ColNet col{};
    bool has_color = false;
    float beta = 0.1f;
    std::vector<uint16_t> sarena;      // K3 weight stream (consumption order, 1952 fragments of 1 KB), every layer padded; host only (prefix of fwd_arena)
    std::vector<uint16_t> sarena_trim; // the 8-wave K3's stream: sarena without the fragments that only hold padding (1920)
    std::vector<uint16_t> sarena_pairs;  // the same with the row blocks of every layer interleaved in pairs (K3 latency variants)
    std::vector<uint16_t> sarena_c;    // K3C: every fragment as an IEEE-half [hi | lo] pair in the 16x16x32 layout (3872 fragments), then K3CC's four per-wave streams (4 x 992)
    std::vector<uint16_t> fwd_arena;   // K4 forward stream: sarena + the 256 feature rows (2080 fragments)
    std::vector<uint16_t> bwd_arena;   // K4 backward stream: transposed geometry layers, then the material / colour head
    int bwd_geo_frags = 0, bwd_frags = 0;
    std::vector<float> shead_row;      // lin8 row 0 (sdf) in fp32: seed of the backward pass
    std::vector<float> barena;
    std::vector<float> cond_r0, b_r0, cond_r4, b_r4, cond_c3, b_c3;   // fp32 cond slices [256][cond]
    std::vector<float> light_xyz, light_area, light_sharp;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);   // grow-only; returns non-zero on failure
    void release();
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// serialises the light-visibility stages of the contexts that share it (include/relightableavatar.h, "frames in flight")
struct ra_gate {
    int device = 0;
    hipEvent_t done = nullptr;      // end of the last stage submitted through the gate
    bool armed = false;
};

// Pinned host staging for small per-frame inputs (bone poses): a slot is filled by the host, copied H2D asynchronously and may be refilled
// once its event has passed — the host never waits for the stream unless it runs more than `n` frames ahead.
struct PinRing {
    static constexpr int n = 8;
    char* base = nullptr;
    size_t slot_bytes = 0;
    int next = 0;
    hipEvent_t ev[n] = {};
    bool used[n] = {};
};

// Launch-variant hints.  The fused MLP kernels are persistent over tiles and correct for ANY device-side count; which variant (workgroup
// width) a launch gets only decides its speed, and it is chosen from a host-side UPPER BOUND of the count.  Where the bound is far above
// the real count (the ground pass: pixels x 512 lights against a handful of fine points), the 8-wave kernel ran almost empty.  Each render
// call therefore copies its fine-count slots to pinned memory behind an event when it ends, and the SAME call of a later frame (calls are
// numbered from ra_set_frame) reads them — if the event has passed; the host never waits — as a hint for the variant choice only.
struct HintSlot {
    hipEvent_t ev = nullptr;
    int* host = nullptr;        // pinned, HINT_SLOTS ints
    bool pending = false;
    int n_pending = 0;          // fine-count slots the pending copy holds
    int n_valid = 0;
    std::vector<int> vals;
};

struct ra_ctx {
    int device = 0;
    std::vector<HintSlot> hints;    // per render call since ra_set_frame
    int call_no = 0;
    HintSlot* cur_hint = nullptr;
    PinRing pin;
    ra_gate* gate = nullptr;
    ra_config cfg{};
    bool have_cfg = false, have_weights = false, have_frame = false;
    bool k3cc_ok = true;        // K3CC's self-test against K3C at ra_finalize_weights (bit for bit); false -> launches of <= 8 Ki points use K3C's 4-wave tiles
    std::map<std::string, std::vector<float>> state_dict;
    HostNets host;
    // device copies
    DevBuf sarena, sarena_pairs, sarena_c, fwd_arena, bwd_arena, shead_row, barena, cond_r0, cond_r4, cond_c3, b_r0, b_r4, b_c3, light_xyz, light_area, light_sharp, light_dir;
    int n_lights = 0;
    // the key-light tier (ra_config.key_light_share): per-light flags of the current frame's probes, on the device
    DevBuf key_mask, key_share;   // (key_share: every light's largest share of a probe's power among the frame's probes)
    bool key_valid = false;       // the flags were computed for the probes this frame is shaded with
    bool key_external = false;    // ... by ra_set_key_probes (the novel-light renderer: every probe of the re-shade); else per render call
    // frame
    FrameState fr{};
    DevBuf fR, fTh, fvertA, fpverts4, fpnorm, ftverts, fbias_r0, fbias_r4, fbias_c3, fcond, fbvh_pts, fbvh_pairs, fbvh_order;
    bool use_bvh = true;
    // N3: vertex -> incident corners of the template mesh (built once per faces array)
    DevBuf adj_start, adj_list, adj_dfaces;
    unsigned long long adj_hash = 0;
    int adj_n_faces = 0, adj_n_verts = 0;
    // scratch (grow-only)
    std::map<std::string, DevBuf> scratch;
    DevBuf dcounters;       // DevCounters (64 B) + at byte 128: the int counters of a chunk (ra_api.cpp: CNT_*, fine-count slots)
    int fc_next = 0;        // next unused fine-count slot (each hdq pass takes a fresh, still-zero one)
    bool fc_wrapped = false;
    bool cnt_zero = false;  // the named counters (hit / ray / sample counts) were zeroed by the chunk's bulk memset
    // host-side counters
    uint64_t n_coarse = 0, n_shaded = 0;
    // timing of the fused MLP launches
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_kind;      // 0: 8-wave K3 launch, 2: narrow K3, 3: K3C, 1: full query (K4 pair)
    size_t ev_used = 0;

    template <typename T> T* buf(const std::string& name, size_t count, int* err) {
        DevBuf& b = scratch[name];
        if (b.ensure(count * sizeof(T))) { *err = 1; return nullptr; }
        return b.as<T>();
    }
};

int ra_pack_weights(ra_ctx* ctx, std::string& err);
