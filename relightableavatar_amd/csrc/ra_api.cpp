// C ABI + orchestration of the render hot path (include/relightableavatar.h).
// Every entry point only enqueues work on the caller's stream; counts that steer later passes
// (fine points, hit pixels, shadow rays) stay on the device and kernels size themselves from them.
#include "ra_ctx.hpp"
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include <cstring>
#include <cstdio>
#include <cstdlib>

static thread_local std::string g_err;
void ra_set_error(const std::string& msg) { g_err = msg; }

#define RA_CHECK(cond, msg)          \
    do {                             \
        if (!(cond)) {               \
            ra_set_error(msg);       \
            return 1;                \
        }                            \
    } while (0)

int DevBuf::ensure(size_t need) {
    if (need <= bytes && p) return 0;
    if (need == 0) need = 16;
    if (p) { hipDeviceSynchronize(); hipFree(p); p = nullptr; bytes = 0; }
    size_t want = need + need / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { ra_set_error(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); p = nullptr; return 1; }
    bytes = want;
    return 0;
}
void DevBuf::release() { if (p) hipFree(p); p = nullptr; bytes = 0; }

void launch_gather_rows(const int* hit_idx, const int* hit_count, int P, const float* src, int C, float* dst, hipStream_t s);

extern "C" {

const char* ra_last_error(void) { return g_err.c_str(); }
int ra_abi_version(void) { return RA_ABI_VERSION; }

int ra_ctx_create(ra_ctx** out, int device) {
    RA_CHECK(out, "ra_ctx_create: null out");
    int n = 0;
    RA_HIP(hipGetDeviceCount(&n));
    RA_CHECK(n > 0, "ra_ctx_create: no HIP device visible (the render path has no CPU fallback)");
    RA_CHECK(device >= 0 && device < n, "ra_ctx_create: bad device index");
    RA_HIP(hipSetDevice(device));
    ra_ctx* c = new ra_ctx();
    c->device = device;
    if (c->dcounters.ensure(1024)) { delete c; return 1; }
    RA_HIP(hipMemset(c->dcounters.p, 0, 1024));
    *out = c;
    return 0;
}

int ra_gate_create(ra_gate** out, int device) {
    RA_CHECK(out, "ra_gate_create: null out");
    RA_HIP(hipSetDevice(device));
    ra_gate* g = new ra_gate();
    g->device = device;
    if (hipEventCreateWithFlags(&g->done, hipEventDisableTiming) != hipSuccess) { delete g; ra_set_error("ra_gate_create: hipEventCreate failed"); return 1; }
    *out = g;
    return 0;
}
int ra_gate_destroy(ra_gate* g) {
    if (!g) return 0;
    hipSetDevice(g->device);
    hipDeviceSynchronize();
    hipEventDestroy(g->done);
    delete g;
    return 0;
}
int ra_set_gate(ra_ctx* c, ra_gate* g) {
    RA_CHECK(c, "ra_set_gate: null context");
    RA_CHECK(!g || g->device == c->device, "ra_set_gate: gate and context live on different devices");
    c->gate = g;
    return 0;
}

int ra_ctx_destroy(ra_ctx* c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    DevBuf* bufs[] = {&c->sarena, &c->sarena_pairs, &c->sarena_c, &c->fwd_arena, &c->bwd_arena, &c->shead_row, &c->barena, &c->cond_r0, &c->cond_r4, &c->cond_c3, &c->b_r0, &c->b_r4, &c->b_c3, &c->light_xyz,
                      &c->light_area, &c->light_sharp, &c->light_dir, &c->fR, &c->fTh, &c->fvertA, &c->fpverts4, &c->fpnorm, &c->ftverts,
                      &c->fbias_r0, &c->fbias_r4, &c->fbias_c3, &c->fcond, &c->dcounters, &c->fbvh_pts, &c->fbvh_pairs, &c->fbvh_order,
                      &c->adj_start, &c->adj_list, &c->adj_dfaces};
    for (DevBuf* b : bufs) b->release();
    c->key_mask.release(); c->key_share.release();
    for (auto& kv : c->scratch) kv.second.release();
    for (auto& e : c->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (int k = 0; k < PinRing::n; ++k) if (c->pin.ev[k]) hipEventDestroy(c->pin.ev[k]);
    for (HintSlot& h : c->hints) { if (h.ev) hipEventDestroy(h.ev); if (h.host) hipHostFree(h.host); }
    if (c->pin.base) hipHostFree(c->pin.base);
    delete c;
    return 0;
}

int ra_default_config(ra_config* o) {
    RA_CHECK(o, "ra_default_config: null argument");
    *o = ra_config{};
    o->xyz_res = 10; o->sdf_res = 8; o->view_res = 4; o->n_bones = 52; o->relight = 1;
    o->resd_limit = 0.05f; o->blend_radius = 0.075f;
    o->albedo_slope = 1.f; o->albedo_bias = 0.f; o->roughness_slope = 0.9f; o->roughness_bias = 0.09f;
    o->fresnel_f0 = 0.02f; o->shading_albedo = 0.8f; o->albedo_multiplier = 1.f;
    o->tonemapping = 1; o->bg_brightness = 0.f; o->mlp_f16 = 1; o->query_skip = 1; o->k4_batch_slots = 0;
    o->trace_precision = 1; o->clip_near = 0.02f; o->clip_far = 10.f;
    o->only_visibility = 0; o->vis_shade_map = 0; o->use_geodesic_filter = 1;
    o->key_light_share = 0.0078f;
    return 0;
}

int ra_set_config(ra_ctx* c, const ra_config* cfg) {
    RA_CHECK(c && cfg, "ra_set_config: null argument");
    RA_CHECK(cfg->n_bones > 0 && cfg->n_bones <= 256, "ra_set_config: bad n_bones");
    RA_CHECK(cfg->trace_precision >= 0 && cfg->trace_precision <= 2, "ra_set_config: trace_precision must be 0, 1 or 2 (a zero-initialised ra_config is not the default: ra_default_config)");
    RA_CHECK(cfg->clip_far > cfg->clip_near, "ra_set_config: clip_far must exceed clip_near (a zero-initialised ra_config is not the default: ra_default_config)");
    RA_CHECK(cfg->vis_shade_map >= 0 && cfg->vis_shade_map <= 2, "ra_set_config: vis_shade_map must be 0, 1 or 2");
    RA_CHECK(cfg->key_light_share >= 0.f && cfg->key_light_share <= 1.f, "ra_set_config: key_light_share must be a fraction in [0, 1] (0 = no key-light tier)");
    c->cfg = *cfg;
    c->have_cfg = true;
    return 0;
}

int ra_set_weight(ra_ctx* c, const char* name, const float* data, size_t numel) {
    RA_CHECK(c && name && (data || numel == 0), "ra_set_weight: null argument");
    c->state_dict[name] = std::vector<float>(data, data + numel);
    c->have_weights = false;
    return 0;
}

static int upload(DevBuf& b, const void* src, size_t bytes, hipStream_t s) {
    if (b.ensure(bytes)) return 1;
    if (bytes) RA_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

// K3CC (csrc/ra_k3cc.hpp) keeps 62 weight fragments in flight in AGPRs it addresses by name; that is safe only while the compiler writes no
// AGPR of its own in that kernel — checked on the shipped object's assembly at build time (csrc/Makefile, tools/check_k3cc_isa.py) and
// HERE, on the device: a few hundred points through K3CC and through K3C's 4-wave tiles (the same arithmetic, weights through LDS) must
// agree bit for bit.  On a mismatch the context never launches K3CC (launch_mlp_sdf_comp allow_coop = false) and says so once on stderr.
// The answer is a property of the kernel's code on this device, not of the weights: one test per process and device.
static std::mutex k3cc_mu;
static std::map<int, bool> k3cc_result;
static int k3cc_self_test(ra_ctx* c, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(k3cc_mu);
        auto it = k3cc_result.find(c->device);
        if (it != k3cc_result.end()) { c->k3cc_ok = it->second; return 0; }
    }
    constexpr int N = 400;                    // 25 tiles of 16 points, the last tile of K3C's 64-point tiles partly filled
    std::vector<float> x(3 * N);
    unsigned u = 12345u;
    for (float& v : x) { u = u * 1664525u + 1013904223u; v = ((u >> 8) * (1.f / 16777216.f) - 0.5f) * 0.9f; }
    std::vector<int> idx(N);
    for (int i = 0; i < N; ++i) idx[i] = i;
    DevBuf bx, bi, bc, ba, bb, bz;
    if (bx.ensure(x.size() * 4) || bi.ensure(N * 4) || bc.ensure(4) || ba.ensure(N * 4) || bb.ensure(N * 4) || bz.ensure(2048)) return 1;
    struct Free { DevBuf* b[6]; ~Free() { for (DevBuf* p : b) p->release(); } } fr_{{&bx, &bi, &bc, &ba, &bb, &bz}};
    RA_HIP(hipMemcpyAsync(bx.p, x.data(), x.size() * 4, hipMemcpyHostToDevice, s));
    RA_HIP(hipMemcpyAsync(bi.p, idx.data(), N * 4, hipMemcpyHostToDevice, s));
    const int n = N;
    RA_HIP(hipMemcpyAsync(bc.p, &n, 4, hipMemcpyHostToDevice, s));
    RA_HIP(hipMemsetAsync(ba.p, 0xff, N * 4, s));
    RA_HIP(hipMemsetAsync(bb.p, 0, N * 4, s));
    RA_HIP(hipMemsetAsync(bz.p, 0, 2048, s));
    FrameState f{};
    f.bias_r0 = bz.as<float>(); f.bias_r4 = bz.as<float>() + 256;          // no frame yet: zero pose biases
    MlpIO io{};
    io.bpts = bx.as<float>(); io.idx = bi.as<int>(); io.count = bc.as<int>(); io.dist_th = 1.f; io.smooth = 0; io.resd_limit = c->cfg.resd_limit;
    io.sdf = ba.as<float>();
    launch_mlp_sdf_comp(c->host.geo, c->sarena_c.p, c->barena.as<float>(), f, io, N, s, true);
    io.sdf = bb.as<float>();
    launch_mlp_sdf_comp(c->host.geo, c->sarena_c.p, c->barena.as<float>(), f, io, N, s, false);
    std::vector<unsigned> a(N), b(N);
    RA_HIP(hipMemcpyAsync(a.data(), ba.p, N * 4, hipMemcpyDeviceToHost, s));
    RA_HIP(hipMemcpyAsync(b.data(), bb.p, N * 4, hipMemcpyDeviceToHost, s));
    RA_HIP(hipStreamSynchronize(s));
    RA_HIP(hipGetLastError());
    c->k3cc_ok = a == b;
    if (!c->k3cc_ok)
        fprintf(stderr, "relightableavatar: K3CC self-test failed (its distances differ from K3C's): the cooperative small-launch kernel is disabled on device %d (ra_k3cc_enabled)\n", c->device);
    std::lock_guard<std::mutex> lk(k3cc_mu);
    k3cc_result[c->device] = c->k3cc_ok;
    return 0;
}

int ra_k3cc_enabled(const ra_ctx* c) { return c && c->have_weights && c->k3cc_ok ? 1 : 0; }

int ra_finalize_weights(ra_ctx* c, void* stream) {
    RA_CHECK(c && c->have_cfg, "ra_finalize_weights: call ra_set_config first");
    hipStream_t s = (hipStream_t)stream;
    RA_HIP(hipSetDevice(c->device));
    std::string err;
    if (ra_pack_weights(c, err)) { ra_set_error("ra_finalize_weights: " + err); return 1; }
    HostNets& H = c->host;
    if (upload(c->sarena, H.sarena_trim.data(), H.sarena_trim.size() * 2, s)) return 1;        // device copy: the trimmed stream (8-wave K3)
    if (upload(c->sarena_pairs, H.sarena_pairs.data(), H.sarena_pairs.size() * 2, s)) return 1;
    if (upload(c->sarena_c, H.sarena_c.data(), H.sarena_c.size() * 2, s)) return 1;
    if (upload(c->fwd_arena, H.fwd_arena.data(), H.fwd_arena.size() * 2, s)) return 1;
    if (upload(c->bwd_arena, H.bwd_arena.data(), H.bwd_arena.size() * 2, s)) return 1;
    if (upload(c->shead_row, H.shead_row.data(), H.shead_row.size() * 4, s)) return 1;
    if (upload(c->barena, H.barena.data(), H.barena.size() * 4, s)) return 1;
    if (upload(c->cond_r0, H.cond_r0.data(), H.cond_r0.size() * 4, s)) return 1;
    if (upload(c->cond_r4, H.cond_r4.data(), H.cond_r4.size() * 4, s)) return 1;
    if (upload(c->b_r0, H.b_r0.data(), H.b_r0.size() * 4, s)) return 1;
    if (upload(c->b_r4, H.b_r4.data(), H.b_r4.size() * 4, s)) return 1;
    if (H.has_color) {
        if (upload(c->cond_c3, H.cond_c3.data(), H.cond_c3.size() * 4, s)) return 1;
        if (upload(c->b_c3, H.b_c3.data(), H.b_c3.size() * 4, s)) return 1;
    }
    if (c->cfg.relight) {
        c->n_lights = (int)H.light_area.size();
        if (upload(c->light_xyz, H.light_xyz.data(), H.light_xyz.size() * 4, s)) return 1;
        if (upload(c->light_area, H.light_area.data(), H.light_area.size() * 4, s)) return 1;
        if (upload(c->light_sharp, H.light_sharp.data(), H.light_sharp.size() * 4, s)) return 1;
        if (c->light_dir.ensure(H.light_xyz.size() * 4)) return 1;
        launch_light_dirs(c->light_xyz.as<float>(), c->n_lights, c->light_dir.as<float>(), s);
    }
    RA_HIP(hipStreamSynchronize(s));     // host staging vectors may be reused
    if (k3cc_self_test(c, s)) return 1;
    c->have_weights = true;
    return 0;
}

int ra_set_frame(ra_ctx* c, const ra_frame* f, void* stream) {
    RA_CHECK(c && f, "ra_set_frame: null argument");
    RA_CHECK(c->have_weights, "ra_set_frame: weights not finalized");
    RA_CHECK(f->R && f->Th && f->poses && f->A && f->big_A && f->pverts && f->pnorm && f->tverts && f->weights, "ra_set_frame: null frame array");
    RA_CHECK(f->n_verts >= 3, "ra_set_frame: need at least 3 vertices (K=3 neighbours)");
    hipStream_t s = (hipStream_t)stream;
    RA_HIP(hipSetDevice(c->device));
    const int nv = f->n_verts, nb = c->cfg.n_bones, cond = nb * 3;
    if (c->fvertA.ensure((size_t)nv * 24 * 4) || c->fpverts4.ensure((size_t)nv * 16) || c->fbias_r0.ensure(1024) || c->fbias_r4.ensure(1024) ||
        c->fbias_c3.ensure(1024) || c->fcond.ensure((size_t)cond * 4))
        return 1;
    // R, Th, pnorm, tverts are read in place: the caller keeps the frame's arrays alive and unchanged until the next ra_set_frame
    // (include/relightableavatar.h) — four copy launches less per frame
    launch_pack_verts(f->pverts, nv, c->fpverts4.as<float4>(), s);
    launch_vert_blend(f->weights, f->A, f->big_A, nv, nb, c->fvertA.as<float>(), s);
    const int nleaf = c->use_bvh ? bvh_leaf_count(nv) : 0;
    const int nsuper = bvh_super_count(nleaf);
    if (nleaf > 0) {
        // leaves: 512 B each; boxes: super boxes (lo | hi), then per super box the four pair records of its leaf boxes
        if (c->fbvh_pts.ensure((size_t)nleaf * 32 * 16) || c->fbvh_pairs.ensure((size_t)nsuper * (32 + 192)) || c->fbvh_order.ensure((size_t)nv * 4)) return 1;
        launch_bvh_build(c->fpverts4.as<float4>(), nv, c->fbvh_order.as<int>(), c->fbvh_pts.as<float>(), c->fbvh_pairs.as<float4>(), nleaf, nsuper, s);
        RA_HIP(hipGetLastError());
    }
    launch_fold_bias(c->cond_r0.as<float>(), cond, 0, cond, f->poses, c->b_r0.as<float>(), c->fbias_r0.as<float>(), s);
    launch_fold_bias(c->cond_r4.as<float>(), cond, 0, cond, f->poses, c->b_r4.as<float>(), c->fbias_r4.as<float>(), s);
    if (c->host.has_color && f->cond_fix)
        launch_fold_bias(c->cond_c3.as<float>(), cond, 0, cond, f->cond_fix, c->b_c3.as<float>(), c->fbias_c3.as<float>(), s);
    FrameState& fr = c->fr;
    fr.R = (float*)f->R; fr.Th = (float*)f->Th; fr.vertA = c->fvertA.as<float>(); fr.pverts4 = c->fpverts4.as<float4>();
    fr.pnorm = (float*)f->pnorm; fr.tverts = (float*)f->tverts; fr.bias_r0 = c->fbias_r0.as<float>();
    fr.bias_r4 = c->fbias_r4.as<float>(); fr.bias_c3 = c->fbias_c3.as<float>(); fr.n_verts = nv;
    fr.bvh_soa = c->fbvh_pts.as<float>();
    fr.bvh_sbox = c->fbvh_pairs.as<float4>(); fr.bvh_lpair = reinterpret_cast<const float*>(fr.bvh_sbox + (size_t)2 * nsuper); fr.bvh_leaves = nleaf; fr.bvh_supers = nsuper;
    c->have_frame = true;
    c->call_no = 0;             // render calls are numbered from here (launch-variant hints, ra_ctx.hpp HintSlot)
    RA_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
namespace {

struct Timer {
    ra_ctx* c; hipStream_t s; int kind; hipEvent_t a = nullptr, b = nullptr;
    Timer(ra_ctx* c_, hipStream_t s_, int kind_) : c(c_), s(s_), kind(kind_) {
        if (!c->timing) return;
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t x, y;
            hipEventCreate(&x); hipEventCreate(&y);
            c->ev_pool.push_back({x, y});
            c->ev_kind.push_back(0);
        }
        c->ev_kind[c->ev_used] = kind;
        a = c->ev_pool[c->ev_used].first; b = c->ev_pool[c->ev_used].second;
        c->ev_used++;
        hipEventRecord(a, s);
    }
    ~Timer() { if (a) hipEventRecord(b, s); }
};

DevCounters* dcnt(ra_ctx* c) { return c->dcounters.as<DevCounters>(); }
int* icnt(ra_ctx* c, int k) { return reinterpret_cast<int*>(c->dcounters.as<char>() + 128) + k; }   // small int counters
enum { CNT_HIT = 1, CNT_RAYS = 2, CNT_SAMP = 3, CNT_FC0 = 8, CNT_FC_SLOTS = 96, CNT_ALL = CNT_FC0 + CNT_FC_SLOTS };

// Every hierarchical-distance pass compacts its fine points through a device counter that must start at zero.  Instead of one
// 4-byte memset launch per pass (21 per relit chunk), the counters are a set that ONE memset zeroes per chunk; each pass takes
// the next unused slot.  Stream order makes the refill safe: the memset runs after every earlier user.
void zero_chunk_counters(ra_ctx* c, hipStream_t s) {
    hipMemsetAsync(icnt(c, 0), 0, CNT_ALL * sizeof(int), s);
    c->fc_next = 0;
    c->fc_wrapped = false;
    c->cnt_zero = true;
}
int* next_fine_counter(ra_ctx* c, hipStream_t s) {
    if (c->fc_next >= CNT_FC_SLOTS) {
        hipMemsetAsync(icnt(c, CNT_FC0), 0, CNT_FC_SLOTS * sizeof(int), s);
        c->fc_next = 0;
        c->fc_wrapped = true;           // slots are being reused: no hints from or for this call
    }
    return icnt(c, CNT_FC0 + c->fc_next++);
}

// one render call's window on the hints (ra_ctx.hpp HintSlot): construction picks the slot of this call and harvests the counts an earlier
// frame left there; destruction queues the copy of this call's fine-count slots behind an event
struct HintScope {
    ra_ctx* c; hipStream_t s;
    HintScope(ra_ctx* c_, hipStream_t s_) : c(c_), s(s_) {
        const int k = c->call_no++;
        if (k >= 64) { c->cur_hint = nullptr; return; }
        if ((int)c->hints.size() <= k) c->hints.resize(k + 1);
        HintSlot& h = c->hints[k];
        if (!h.ev) {
            if (hipEventCreateWithFlags(&h.ev, hipEventDisableTiming) != hipSuccess || hipHostMalloc((void**)&h.host, CNT_FC_SLOTS * sizeof(int)) != hipSuccess) {
                h.ev = nullptr; h.host = nullptr; c->cur_hint = nullptr; (void)hipGetLastError(); return;
            }
        }
        if (h.pending && hipEventQuery(h.ev) == hipSuccess) {
            h.vals.assign(h.host, h.host + h.n_pending);
            h.n_valid = h.n_pending;
            h.pending = false;
        }
        (void)hipGetLastError();            // hipEventQuery's hipErrorNotReady is not an error of the call
        c->cur_hint = &h;
    }
    ~HintScope() {
        HintSlot* h = c->cur_hint;
        c->cur_hint = nullptr;
        if (!h || h->pending || c->fc_next <= 0 || c->fc_wrapped) return;      // an unread copy is still in flight: keep it
        if (hipMemcpyAsync(h->host, icnt(c, CNT_FC0), (size_t)c->fc_next * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return;
        if (hipEventRecord(h->ev, s) != hipSuccess) return;
        h->n_pending = c->fc_next;
        h->pending = true;
    }
};

// the fine count the pass that takes fine-count slot k found in an earlier frame (-1: unknown)
int fine_hint(const ra_ctx* c, int k) {
    const HintSlot* h = c->cur_hint;
    return (h && !c->fc_wrapped && k < h->n_valid) ? h->vals[k] : -1;
}
// the size the variant choice of a fused MLP launch sees: the bound, or — with a hint — a quarter more than the earlier count
// A hint comes from an EARLIER frame: after a camera cut the count can be many times larger.  Every variant is correct for every count
// (persistent over tiles), but a narrow variant on a grid sized for the hint would be a cliff: the hint picks the workgroup WIDTH only.
int variant_size(int n, int hint) {
    if (hint < 0) return n;
    const long long v = (long long)hint + hint / 4 + 1024;
    return v < n ? (int)v : n;
}
// ... and the size its grid is made for (mlp_grid, ra_common.hpp): never below an eighth of the bound, whatever the hint says
int grid_size(int n, int nv) { return nv > n / 8 ? nv : n / 8; }

void k3_launch(ra_ctx* c, const MlpIO& io, int n, hipStream_t s, int grid_slots = 0) {
    if (c->cfg.mlp_f16) launch_mlp_sdf_stream_f16(c->host.geo, c->sarena.p, c->sarena_pairs.p, c->barena.as<float>(), c->fr, io, n, s, grid_slots);
    else launch_mlp_sdf_stream_bf16(c->host.geo, c->sarena.p, c->sarena_pairs.p, c->barena.as<float>(), c->fr, io, n, s, grid_slots);
}

// which distance queries run in compensated arithmetic (ra_config.trace_precision): the surface trace from 1 on, everything at 2
// (the light-visibility rays towards the frame's key lights join them through hdq_pass's second fine list: ra_config.key_light_share)
enum { Q_OTHER = 0, Q_SURFACE = 1 };
bool precise(const ra_ctx* c, int what) { return c->cfg.trace_precision >= 2 || (c->cfg.trace_precision == 1 && what == Q_SURFACE); }
constexpr int KEY_LIGHTS_MAX = 48;       // per frame; bounds the second ray list of a light-visibility stage (rays <= pixels x this) and the tier's cost
bool key_tier(const ra_ctx* c) { return c->cfg.trace_precision == 1 && c->cfg.key_light_share > 0.f && c->n_lights > 0; }
// the frame's key-light flags from the probe a render call shades with — unless the caller named the frame's probes itself (ra_set_key_probes)
int key_mask_from(ra_ctx* c, const float* probe, int ph, int pw, hipStream_t s) {
    if (c->key_external) return 0;
    c->key_valid = false;
    if (!key_tier(c) || !probe) return 0;
    if (c->key_mask.ensure((size_t)c->n_lights) || c->key_share.ensure((size_t)c->n_lights * sizeof(float))) return 1;
    launch_key_lights(probe, 1, ph, pw, c->light_dir.as<float>(), c->light_area.as<float>(), c->n_lights, c->cfg.key_light_share, KEY_LIGHTS_MAX, 0,
                      c->key_share.as<float>(), c->key_mask.as<unsigned char>(), s);
    c->key_valid = true;
    return 0;
}

// the fine level of one query: K3, or K3C where the pass is in the precise tier
// n: upper bound of the device-side count; hint: the count this pass found in an earlier frame (-1: none).  Every variant is correct for
// every count (persistent over tiles); the size only picks the workgroup width and the grid.
void fine_level(ra_ctx* c, const MlpIO& io, int n, bool comp, hipStream_t s, int hint = -1) {
    const int nv = variant_size(n, hint);
    if (comp) {
        Timer t(c, s, 3);
        launch_mlp_sdf_comp(c->host.geo, c->sarena_c.p, c->barena.as<float>(), c->fr, io, nv, s, c->k3cc_ok, grid_size(n, nv));
    } else {
        Timer t(c, s, k3_waves(nv) == 8 ? 0 : 2);      // timed per kernel family: 0 = 8-wave K3, 2 = the narrow variants
        k3_launch(c, io, nv, s, grid_size(n, nv));
    }
}

// one hierarchical distance query over the points of rs; writes sdf[n]
// key / n_key (the key-light tier; shadow rays only): the fine points of rays towards lights with key[light] != 0 — at most n_key — form
// a second fine list that the compensated kernel answers; the pass then is ONE coarse launch + K3 on the first list + K3C / K3CC on the second
int hdq_pass(ra_ctx* c, const RaySet& rs, int n, float th, int smooth, float* sdf, hipStream_t s, int what = Q_OTHER,
             const unsigned char* key = nullptr, int n_key = 0) {
    if (n <= 0) return 0;
    int err = 0;
    int* fine_idx = c->buf<int>("fine_idx", n, &err);
    float* bpts = c->buf<float>("fine_bpts", (size_t)n * 3, &err);
    if (err) return 1;
    HdqOut out{};
    out.sdf = sdf; out.fine_count = next_fine_counter(c, s); out.fine_idx = fine_idx; out.bpts = bpts;
    const int hint = fine_hint(c, c->fc_next - 1);
    int hint2 = -1;
    if (key && n_key > 0) {
        out.key = key;
        out.fine_idx2 = c->buf<int>("fine_idx_k", n_key, &err);
        out.bpts2 = c->buf<float>("fine_bpts_k", (size_t)n_key * 3, &err);
        if (err) return 1;
        out.fine_count2 = next_fine_counter(c, s);
        hint2 = fine_hint(c, c->fc_next - 1);
    }
    out.counters = dcnt(c);
    launch_hdq_coarse(c->fr, rs, n, th, c->cfg.blend_radius, out, s, c->cfg.use_geodesic_filter != 0);
    MlpIO io{};
    io.bpts = bpts; io.idx = fine_idx; io.count = out.fine_count; io.sdf = sdf; io.dist_th = th; io.smooth = smooth;
    io.resd_limit = c->cfg.resd_limit; io.counters = dcnt(c);
    fine_level(c, io, n, precise(c, what), s, hint);
    if (out.key) {
        io.bpts = out.bpts2; io.idx = out.fine_idx2; io.count = out.fine_count2;
        fine_level(c, io, n_key, true, s, hint2);
    }
    return 0;
}

// the full query on the compacted fine list: forward with tape + reverse-mode backward + heads, in sub-batches of
// cfg.k4_batch_slots fine slots that share ONE tape (4.9 KB per slot).  n is only the upper bound of the device-side
// count: launch pairs beyond it find no tile and exit at once.
int full_query(ra_ctx* c, FullIO io, int n, hipStream_t s) {
    // frames in flight: a full query that fills the chip (the volume path's) takes its turn at the gate like a light-visibility stage
    const bool gated = c->gate && n > 65536;
    if (gated && c->gate->armed) RA_HIP(hipStreamWaitEvent(s, c->gate->done, 0));
    struct Release {
        ra_ctx* c; hipStream_t s; bool on;
        ~Release() { if (on) { hipEventRecord(c->gate->done, s); c->gate->armed = true; } }
    } release{c, s, gated};
    Timer t(c, s, 1);
    const int batch = c->cfg.k4_batch_slots > 0 ? c->cfg.k4_batch_slots : (1 << 20);
    const int cap = n < batch ? n : batch;
    int err = 0;
    char* tape = c->buf<char>("k4_tape", mlp_full_rev_tape_bytes(cap), &err);
    if (err) {
        ra_set_error("full query: no device memory for the activation tape (" + std::to_string(mlp_full_rev_tape_bytes(cap) >> 20) +
                     " MB): lower cfg.k4_batch_slots / cfg.volume_chunk_rays");
        return 1;
    }
    const bool f16w = c->cfg.mlp_f16 != 0;
    for (int s0 = 0; s0 < n; s0 += cap) {
        io.slot0 = s0;
        io.slot_cap = n - s0 < cap ? n - s0 : cap;
        if (f16w) {
            launch_mlp_fwd_tape_f16(c->host.geo, c->fwd_arena.p, c->barena.as<float>(), c->fr, io, tape, s);
            launch_mlp_bwd_heads_f16(c->host.mat, c->host.col, c->bwd_arena.p, c->barena.as<float>(), c->shead_row.as<float>(), c->fr, io, tape, s);
        } else {
            launch_mlp_fwd_tape_bf16(c->host.geo, c->fwd_arena.p, c->barena.as<float>(), c->fr, io, tape, s);
            launch_mlp_bwd_heads_bf16(c->host.mat, c->host.col, c->bwd_arena.p, c->barena.as<float>(), c->shead_row.as<float>(), c->fr, io, tape, s);
        }
    }
    return 0;
}

// Network.forward (eval) on n (or *n_dev) points: raw[n][C], zero for non-fine points
int forward_pass(ra_ctx* c, const float* x, const float* v, int n, const int* n_dev, float th, float* raw, hipStream_t s) {
    if (n <= 0) return 0;
    int err = 0;
    const int C = c->cfg.relight ? 17 : 16;
    int* fine_idx = c->buf<int>("fine_idx", n, &err);
    float* bpts = c->buf<float>("fine_bpts", (size_t)n * 3, &err);
    float* mats = c->buf<float>("fine_mats", (size_t)n * 24, &err);
    float* sdf = c->buf<float>("fwd_sdf", n, &err);
    if (err) return 1;
    RaySet rs{};
    rs.mode = 0; rs.x = x; rs.n_dev = n_dev;
    HdqOut out{};
    out.sdf = sdf; out.fine_count = next_fine_counter(c, s); out.fine_idx = fine_idx; out.bpts = bpts; out.mats = mats;
    out.raw_zero = raw; out.raw_C = C;          // points outside dist_th: zero rows, written by the coarse level itself
    out.counters = dcnt(c);
    launch_hdq_coarse(c->fr, rs, n, th, c->cfg.blend_radius, out, s, c->cfg.use_geodesic_filter != 0);
    FullIO io{};
    io.bpts = bpts; io.mats = mats; io.view = v; io.idx = fine_idx; io.count = out.fine_count; io.raw = raw; io.C = C;
    io.beta = c->host.beta; io.resd_limit = c->cfg.resd_limit;
    io.albedo_slope = c->cfg.albedo_slope; io.albedo_bias = c->cfg.albedo_bias;
    io.rough_slope = c->cfg.roughness_slope; io.rough_bias = c->cfg.roughness_bias;
    io.relight = c->cfg.relight;
    io.counters = dcnt(c);
    if (full_query(c, io, n, s)) return 1;
    return 0;
}

int check_ready(ra_ctx* c, const char* who) {
    if (!c) { ra_set_error(std::string(who) + ": null ctx"); return 1; }
    if (!c->have_weights) { ra_set_error(std::string(who) + ": weights not finalized"); return 1; }
    if (!c->have_frame) { ra_set_error(std::string(who) + ": no frame set (ra_set_frame)"); return 1; }
    if (hipSetDevice(c->device) != hipSuccess) { ra_set_error(std::string(who) + ": hipSetDevice failed"); return 1; }
    return 0;
}

TraceState alloc_trace(ra_ctx* c, const char* pfx, int n, bool soft, int* err) {
    TraceState ts{};
    std::string p(pfx);
    ts.t = c->buf<float>(p + "t", n, err);
    ts.d0 = c->buf<float>(p + "d0", n, err);
    ts.occ = c->buf<float>(p + "occ", n, err);
    ts.ot = c->buf<float>(p + "ot", n, err);
    ts.stuck = c->buf<unsigned char>(p + "stuck", n, err);
    if (!soft) {
        ts.dt = c->buf<float>(p + "dt", n, err);
        ts.st = c->buf<float>(p + "st", n, err);
        ts.cd = c->buf<float>(p + "cd", n, err);
        ts.off = c->buf<float>(p + "off", n, err);
        ts.rlx = c->buf<float>(p + "rlx", n, err);
    }
    return ts;
}

}  // namespace

extern "C" {

int ra_raw_channels(const ra_ctx* c) { return c && c->cfg.relight ? 17 : 16; }

int ra_hdq_sdf(ra_ctx* c, const float* x, int n, float dist_th, int smooth, float* sdf, void* stream) {
    if (check_ready(c, "ra_hdq_sdf")) return 1;
    RA_CHECK(n >= 0 && (n == 0 || (x && sdf)), "ra_hdq_sdf: bad arguments");
    RaySet rs{};
    rs.mode = 0; rs.x = x;
    if (hdq_pass(c, rs, n, dist_th, smooth, sdf, (hipStream_t)stream)) return 1;
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_observed_sdf(ra_ctx* c, const float* bpts, int n, float* sdf, void* stream) {
    if (check_ready(c, "ra_observed_sdf")) return 1;
    RA_CHECK(n >= 0 && (n == 0 || (bpts && sdf)), "ra_observed_sdf: bad arguments");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    int* idx = c->buf<int>("fine_idx", n, &err);
    if (err) return 1;
    int* cnt = next_fine_counter(c, s);
    launch_iota(idx, n, cnt, s);
    MlpIO io{};
    io.bpts = bpts; io.idx = idx; io.count = cnt; io.sdf = sdf; io.dist_th = 1.f; io.smooth = 0;
    io.resd_limit = c->cfg.resd_limit; io.counters = dcnt(c);
    fine_level(c, io, n, precise(c, Q_OTHER), s);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_bigpose_transform(ra_ctx* c, const float* x, int n, const float* R, const float* Th, int invert, float* out, void* stream) {
    if (check_ready(c, "ra_bigpose_transform")) return 1;
    RA_CHECK(n >= 0 && (n == 0 || (x && out && R && Th)), "ra_bigpose_transform: bad arguments");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    int* fine_idx = c->buf<int>("fine_idx", n, &err);
    float* fb = c->buf<float>("fine_bpts", (size_t)n * 3, &err);
    float* sdfc = c->buf<float>("bt_sdf", n, &err);
    float* sb = c->buf<float>("bt_sdf_batch", (size_t)n * 3, &err);
    int* nb = c->buf<int>("bt_nn", (size_t)n * 3, &err);
    float* d2 = c->buf<float>("bt_d2", (size_t)n * 3, &err);
    float* bp = c->buf<float>("bt_bpts", (size_t)n * 3, &err);
    float* tp = c->buf<float>("bt_tpts", (size_t)n * 3, &err);
    float* mats = c->buf<float>("bt_mats", (size_t)n * 24, &err);
    if (err) return 1;
    RaySet rs{};
    rs.mode = 0; rs.x = x;
    HdqOut o{};
    o.sdf = sdfc; o.fine_count = next_fine_counter(c, s); o.fine_idx = fine_idx; o.bpts = fb;
    o.dbg_sdf_batch = sb; o.dbg_nn_batch = nb; o.dbg_d2 = d2; o.dbg_bpts = bp; o.dbg_tpts = tp; o.dbg_mats = mats;
    o.counters = dcnt(c);
    launch_hdq_coarse(c->fr, rs, n, 1e9f, c->cfg.blend_radius, o, s, c->cfg.use_geodesic_filter != 0);      // transform=False -> filtering off: dist = 1e9 (:253-259)
    launch_bigpose_compose(mats, d2, n, c->cfg.blend_radius, R, Th, invert, out, s);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_forward(ra_ctx* c, const float* x, const float* v, int n, float dist_th, float* raw, void* stream) {
    if (check_ready(c, "ra_forward")) return 1;
    RA_CHECK(n >= 0 && (n == 0 || (x && raw)), "ra_forward: bad arguments");
    RA_CHECK(c->cfg.relight || v || n == 0, "ra_forward: the AniSDF colour net needs view directions");
    if (forward_pass(c, x, v, n, nullptr, dist_th, raw, (hipStream_t)stream)) return 1;
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_sphere_trace(ra_ctx* c, const float* ray_o, const float* ray_d, const float* near_, const float* far_, const float* tan_i,
                    int n, const ra_trace_params* p, float* surf, float* occ, float* st, float* ot, void* stream) {
    if (check_ready(c, "ra_sphere_trace")) return 1;
    RA_CHECK(p && n >= 0 && (n == 0 || (ray_o && ray_d && near_ && far_)), "ra_sphere_trace: bad arguments");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    TraceState ts = alloc_trace(c, "tr_", n, false, &err);   // full state for either mode
    float* sdf = c->buf<float>("tr_sdf", n, &err);
    if (err) return 1;
    ts.near_ = near_; ts.far_ = far_; ts.tan_i = tan_i; ts.light = nullptr;
    launch_trace_init(ts, n, nullptr, *p, s);
    RaySet rs{};
    rs.mode = 1; rs.o = ray_o; rs.d = ray_d; rs.t = ts.t;
    rs.nn_hint = c->buf<int>("tr_nn", (size_t)n * 3, &err);      // every iteration starts from the neighbours of the one before
    if (err) return 1;
    for (int it = 0; it < p->iters; ++it) {
        rs.hint_valid = it > 0;
        if (hdq_pass(c, rs, n, p->dist_th, 1, sdf, s, p->soft_shadow ? Q_OTHER : Q_SURFACE)) return 1;
        launch_trace_update(ts, sdf, n, nullptr, it, *p, s);
    }
    if (occ) RA_HIP(hipMemcpyAsync(occ, ts.occ, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    if (st) RA_HIP(hipMemcpyAsync(st, ts.st, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    if (ot) RA_HIP(hipMemcpyAsync(ot, ts.ot, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    if (surf) {
        float* depth = c->buf<float>("tr_depth", n, &err);
        float* acc = c->buf<float>("tr_acc", n, &err);
        int* hidx = c->buf<int>("tr_hidx", n, &err);
        if (err) return 1;
        launch_surface_finish(ray_o, ray_d, ts.st, ts.occ, n, surf, depth, acc, hidx, icnt(c, CNT_HIT), s);
    }
    RA_HIP(hipGetLastError());
    return 0;
}

// light_visibility (sphere_tracing_renderer.py:265-344) for the hit slots of one chunk: per (slot, light) cosine and
// visibility; rays that face the light and cross the box are sphere traced with the DFSS state machine (HOT LOOP B).
static int light_visibility_stage(ra_ctx* c, const float* surf, const float* norm_slots, const float* acc, const int* hit_idx,
                                  const int* hit_count, int P, const float* bbox, float near_offset, const ra_trace_params& shadow,
                                  int no_visibility, int local_visibility, float** lvis_out, float** ldot_out, hipStream_t s,
                                  int n_boxes = 0, const float* boxes = nullptr, const int* box_start = nullptr, const int* pix_nn = nullptr,
                                  const int* perm = nullptr, bool split_wide_groups = false) {
    int err = 0;
    const int L = c->n_lights;
    const size_t NR = (size_t)P * L;
    float* lvis = c->buf<float>("lv_lvis", NR, &err);
    float* ldot = c->buf<float>("lv_ldot", NR, &err);
    ShadowGen g{};
    g.surf = surf; g.norm = norm_slots; g.acc = acc; g.hit_idx = hit_idx; g.hit_count = hit_count; g.ldir = c->light_dir.as<float>();
    for (int k = 0; k < 6; ++k) g.bbox[k] = bbox[k];
    g.n_boxes = n_boxes > 1 ? n_boxes : 0;
    for (int j = 0; j < g.n_boxes; ++j) {
        for (int k = 0; k < 6; ++k) g.boxes[j][k] = boxes[6 * j + k];
        g.box_start[j] = box_start[j];
    }
    if (g.n_boxes) g.box_start[g.n_boxes] = box_start[g.n_boxes];
    g.perm = g.n_boxes ? perm : nullptr;
    g.near_offset = near_offset; g.L = L; g.no_visibility = no_visibility; g.local_visibility = local_visibility;
    g.split_wide_groups = split_wide_groups ? 1 : 0;
    g.lvis = lvis; g.ldot = ldot;
    const bool traced = !(no_visibility || local_visibility);
    if (traced) {
        g.ray_pix = c->buf<int>("lv_pix", NR, &err);
        g.ray_light = c->buf<int>("lv_light", NR, &err);
        g.ray_slot = c->buf<int>("lv_slot", NR, &err);
        g.near_ = c->buf<float>("lv_near", NR, &err);
        g.far_ = c->buf<float>("lv_far", NR, &err);
    }
    g.ray_count = icnt(c, CNT_RAYS);
    TraceState sh{};
    float* ssdf = nullptr;
    if (traced) {
        sh = alloc_trace(c, "sh_", (int)NR, shadow.soft_shadow != 0, &err);      // hard shadows (cfg.no_dfss) run the surface trace's state machine (:182-197)
        ssdf = c->buf<float>("sh_sdf", NR, &err);
    }
    // the key-light tier: the fine points of the rays towards the frame's key lights (at most KEY_LIGHTS_MAX; flags on the device) form a
    // second fine list in every pass of the loop below, answered by the compensated kernel
    const bool keyed = traced && key_tier(c) && c->key_valid;
    const size_t NK = keyed ? (size_t)P * (size_t)(L < KEY_LIGHTS_MAX ? L : KEY_LIGHTS_MAX) : 0;
    if (err) return 1;
    // frames in flight: this stage (the frame's large launches) starts when the stage submitted before it through the same gate has ended
    if (c->gate && c->gate->armed) RA_HIP(hipStreamWaitEvent(s, c->gate->done, 0));
    launch_shadow_gen(g, P, s, c->cnt_zero);     // the chunk's bulk memset covers the first shadow stage; a second one zeroes its counter itself
    c->cnt_zero = false;
    if (traced) {
        sh.near_ = g.near_; sh.far_ = g.far_; sh.tan_i = c->light_sharp.as<float>(); sh.light = g.ray_light;
        launch_trace_init(sh, (int)NR, g.ray_count, shadow, s);
        RaySet r2{};
        r2.mode = 2; r2.o = surf; r2.t = sh.t; r2.pix = g.ray_pix; r2.light = g.ray_light; r2.ldir = c->light_dir.as<float>();
        r2.n_dev = g.ray_count;
        r2.skip = c->cfg.query_skip ? sh.stuck : nullptr;
        r2.nn_hint = c->buf<int>("lv_nn", NR * 3, &err);       // every iteration starts from the neighbours of the one before
        if (err) return 1;
        for (int it = 0; it < shadow.iters; ++it) {
            r2.hint_valid = it > 0;
            // first pass: a shadow ray starts next to its pixel's surface point, whose neighbours the surface trace's last query found
            r2.hint_src = it == 0 ? pix_nn : nullptr;
            r2.hint_src_index = it == 0 ? g.ray_pix : nullptr;
            if (hdq_pass(c, r2, (int)NR, shadow.dist_th, 1, ssdf, s, Q_OTHER, keyed ? c->key_mask.as<unsigned char>() : nullptr, (int)NK)) return 1;
            launch_trace_update(sh, ssdf, (int)NR, g.ray_count, it, shadow, s);
        }
        launch_shadow_scatter(sh.occ, g.ray_slot, g.ray_count, (int)NR, lvis, s);
        launch_accumulate(g.ray_count, &dcnt(c)->n_shadow_rays, s);
    }
    if (c->gate) { RA_HIP(hipEventRecord(c->gate->done, s)); c->gate->armed = true; }
    *lvis_out = lvis;
    *ldot_out = ldot;
    return 0;
}

int ra_debug_key_lights(ra_ctx* c, unsigned char* key_dev, float* share_dev, void* stream) {
    RA_CHECK(c && key_dev && share_dev, "ra_debug_key_lights: null argument");
    RA_CHECK(c->key_valid && c->n_lights > 0, "ra_debug_key_lights: no key lights have been computed (ra_set_key_probes, or a render call with a probe)");
    RA_HIP(hipSetDevice(c->device));
    RA_HIP(hipMemcpyAsync(key_dev, c->key_mask.p, (size_t)c->n_lights, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    RA_HIP(hipMemcpyAsync(share_dev, c->key_share.p, (size_t)c->n_lights * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int ra_begin_render(ra_ctx* c) {
    RA_CHECK(c, "ra_begin_render: null ctx");
    c->call_no = 0;          // render calls are numbered from here (launch-variant hints: the k-th call of a frame reads the k-th call's counts of an earlier one)
    return 0;
}

int ra_set_key_probes(ra_ctx* c, const float* probes, int n, int ph, int pw, int accumulate, void* stream) {
    RA_CHECK(c && n >= 0 && (n == 0 || (probes && ph > 0 && pw > 0)), "ra_set_key_probes: bad arguments");
    if (n == 0) { c->key_external = false; c->key_valid = false; return 0; }
    RA_CHECK(c->have_weights && c->n_lights > 0, "ra_set_key_probes: needs the relight network's light set (ra_finalize_weights)");
    RA_HIP(hipSetDevice(c->device));
    const bool acc = accumulate && c->key_external && c->key_valid;
    c->key_external = true;
    if (!key_tier(c)) { c->key_valid = false; return 0; }
    if (c->key_mask.ensure((size_t)c->n_lights) || c->key_share.ensure((size_t)c->n_lights * sizeof(float))) return 1;
    launch_key_lights(probes, n, ph, pw, c->light_dir.as<float>(), c->light_area.as<float>(), c->n_lights, c->cfg.key_light_share, KEY_LIGHTS_MAX,
                      acc ? 1 : 0, c->key_share.as<float>(), c->key_mask.as<unsigned char>(), (hipStream_t)stream);
    c->key_valid = true;
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_render_sphere_chunk(ra_ctx* c, const float* ray_o, const float* ray_d, const float* near_, const float* far_, int P,
                           const float* bbox, const float* probe, int ph, int pw, const ra_sphere_params* p,
                           const ra_render_out* out, void* stream) {
    if (check_ready(c, "ra_render_sphere_chunk")) return 1;
    RA_CHECK(p && out && P >= 0, "ra_render_sphere_chunk: bad arguments");
    if (P == 0) return 0;
    RA_CHECK(ray_o && ray_d && near_ && far_, "ra_render_sphere_chunk: null ray arrays");
    const bool relit = p->relighting != 0;
    RA_CHECK(!relit || (c->cfg.relight && probe && bbox && c->n_lights > 0), "ra_render_sphere_chunk: relighting needs the relight network, a probe and a bbox");
    RA_CHECK(p->n_samples >= 1 && p->n_samples <= 16, "ra_render_sphere_chunk: n_samples out of range");
    RA_CHECK(!relit || (long long)P * c->n_lights < (1ll << 31), "ra_render_sphere_chunk: chunk too large (rays x lights must fit an int): lower cfg.render_chunk_size / cfg.sphere_chunk_rays");
    if (p->n_boxes > 1) {
        RA_CHECK(p->n_boxes <= RA_MAX_BOXES && p->boxes && p->box_start && bbox, "ra_render_sphere_chunk: at most 32 boxes per call, with their tables");
        RA_CHECK(p->box_start[0] == 0 && p->box_start[p->n_boxes] == P, "ra_render_sphere_chunk: box_start must run from 0 to P");
        for (int j = 0; j < p->n_boxes; ++j) RA_CHECK(p->box_start[j] <= p->box_start[j + 1], "ra_render_sphere_chunk: box_start must ascend");
    }
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    const int S = p->n_samples, C = c->cfg.relight ? 17 : 16, L = c->n_lights;
    zero_chunk_counters(c, s);                // ONE memset for every device counter this chunk uses
    HintScope hints(c, s);
    // ---- spatially coherent ray order (per-ray results are order-free; outputs go back through perm)
    const int* perm = nullptr;
    if (bbox) {
        const size_t tb = sort_hits_temp_bytes(P);
        unsigned* k0 = c->buf<unsigned>("hs_k0", P, &err);
        unsigned* k1 = c->buf<unsigned>("hs_k1", P, &err);
        int* v0 = c->buf<int>("hs_v0", P, &err);
        int* pm_ = c->buf<int>("rs_perm", P, &err);
        char* tmp = c->buf<char>("hs_tmp", tb + 16, &err);
        float* so = c->buf<float>("rs_o", (size_t)P * 3, &err);
        float* sd = c->buf<float>("rs_d", (size_t)P * 3, &err);
        float* sn = c->buf<float>("rs_n", P, &err);
        float* sf = c->buf<float>("rs_f", P, &err);
        if (err) return 1;
        const float bmin[3] = {bbox[0], bbox[1], bbox[2]};
        if (launch_sort_rays(ray_o, ray_d, near_, far_, P, bmin, k0, k1, v0, pm_, tmp, tb, so, sd, sn, sf, s)) { ra_set_error("ra_render_sphere_chunk: radix sort failed"); return 1; }
        ray_o = so; ray_d = sd; near_ = sn; far_ = sf; perm = pm_;
    }
    // ---- surface trace (HOT LOOP A)
    TraceState ts = alloc_trace(c, "sf_", P, false, &err);
    float* sdf = c->buf<float>("sf_sdf", P, &err);
    float* surf = c->buf<float>("sf_surf", (size_t)P * 3, &err);
    float* depth = c->buf<float>("sf_depth", P, &err);
    float* acc = c->buf<float>("sf_acc", P, &err);
    int* hit_idx = c->buf<int>("sf_hit", P, &err);
    int* slot_of_ray = c->buf<int>("sf_slot", P, &err);
    if (err) return 1;
    ts.near_ = near_; ts.far_ = far_;
    launch_trace_init(ts, P, nullptr, p->surface, s);
    RaySet rs{};
    rs.mode = 1; rs.o = ray_o; rs.d = ray_d; rs.t = ts.t; rs.skip = c->cfg.query_skip ? ts.stuck : nullptr;
    rs.nn_hint = c->buf<int>("sf_nn", (size_t)P * 3, &err);      // every iteration starts from the neighbours of the one before
    if (err) return 1;
    for (int it = 0; it < p->surface.iters; ++it) {
        rs.hint_valid = it > 0;
        if (hdq_pass(c, rs, P, p->surface.dist_th, 1, sdf, s, Q_SURFACE)) return 1;
        launch_trace_update(ts, sdf, P, nullptr, it, p->surface, s);
    }
    int* hit_count = icnt(c, CNT_HIT);
    launch_surface_finish(ray_o, ray_d, ts.st, ts.occ, P, surf, depth, acc, hit_idx, hit_count, s, slot_of_ray, true);
    launch_accumulate(hit_count, &dcnt(c)->n_hit_pixels, s);
    if (relit) {   // spatially coherent hit order for the shadow trace (results are scattered back, so order-free)
        const size_t tb = sort_hits_temp_bytes(P);
        unsigned* k0 = c->buf<unsigned>("hs_k0", P, &err);
        unsigned* k1 = c->buf<unsigned>("hs_k1", P, &err);
        int* v0 = c->buf<int>("hs_v0", P, &err);
        char* tmp = c->buf<char>("hs_tmp", tb + 16, &err);
        if (err) return 1;
        const float bmin[3] = {bbox[0], bbox[1], bbox[2]};
        if (launch_sort_hits(surf, acc, P, bmin, k0, k1, v0, hit_idx, tmp, tb, s)) { ra_set_error("ra_render_sphere_chunk: radix sort failed"); return 1; }
    }
    launch_slot_index(hit_idx, hit_count, P, slot_of_ray, s);          // ray -> hit slot in the final hit order (-1: miss)
    // ---- material query on S samples around each hit (render_human :602-620)
    float* xs = c->buf<float>("mt_x", (size_t)P * S * 3, &err);
    float* vs = c->buf<float>("mt_v", (size_t)P * S * 3, &err);
    float* raw = c->buf<float>("mt_raw", (size_t)P * S * C, &err);
    SurfaceMaps m{};
    m.cpts = c->buf<float>("mp_cpts", (size_t)P * 3, &err);
    m.bpts = c->buf<float>("mp_bpts", (size_t)P * 3, &err);
    m.resd = c->buf<float>("mp_resd", (size_t)P * 3, &err);
    m.norm = c->buf<float>("mp_norm", (size_t)P * 3, &err);
    m.albedo = c->buf<float>("mp_albedo", (size_t)P * 3, &err);
    m.rough = c->buf<float>("mp_rough", P, &err);
    m.rgb = c->buf<float>("mp_rgb", (size_t)P * 3, &err);
    m.valbedo = (out->volume_albedo && c->cfg.relight) ? c->buf<float>("mp_valbedo", (size_t)P * 3, &err) : nullptr;
    if (err) return 1;
    launch_surface_samples(surf, ray_d, hit_idx, hit_count, P, S, p->surf_sample_range, xs, vs, icnt(c, CNT_SAMP), s);
    if (forward_pass(c, xs, vs, P * S, icnt(c, CNT_SAMP), p->dist_th, raw, s)) return 1;
    launch_surface_composite(raw, C, S, hit_count, P, c->cfg.relight, c->cfg, m, s);
    // ---- light visibility + shading (HOT LOOP B)
    float *lvis = nullptr, *ldot = nullptr, *shade = nullptr, *spec = nullptr;
    if (relit) {
        if (key_mask_from(c, probe, ph, pw, s)) return 1;
        if (light_visibility_stage(c, surf, m.norm, acc, hit_idx, hit_count, P, bbox, p->shadow_near_offset, p->shadow,
                                   p->no_visibility, p->local_visibility, &lvis, &ldot, s, p->n_boxes, p->boxes, p->box_start, rs.nn_hint, perm, true)) return 1;
        m.rgb = c->buf<float>("mp_rgb", (size_t)P * 3, &err);
        shade = c->buf<float>("mp_shade", (size_t)P * 3, &err);
        spec = c->buf<float>("mp_spec", (size_t)P * 3, &err);
        if (err) return 1;
        ShadeIn in{};
        in.ray_o = ray_o; in.surf = surf; in.idx = hit_idx; in.count = hit_count; in.n = P;
        in.norm = m.norm; in.albedo = m.albedo; in.rough = m.rough; in.lvis = lvis; in.ldot = ldot;
        in.light_xyz = c->light_xyz.as<float>(); in.light_area = c->light_area.as<float>(); in.L = L;
        in.probes = probe; in.n_probes = 1; in.ph = ph; in.pw = pw; in.want_spec = out->spec != nullptr;
        in.rgb = m.rgb; in.shade = shade; in.spec = spec;
        launch_shade(in, c->cfg, s);
        c->n_shaded += 0;   // counted on device via hit pixels
    }
    // ---- every requested map to the full ray set (zeros elsewhere), premultiplied by acc (alpha_output_): one launch
    const int pm = p->premultiply;
    EmitMaps em{};
    em.slot_of_ray = slot_of_ray; em.acc = acc; em.perm = perm; em.P = P;
    long long total = 0;
    auto scat = [&](float* dst, const float* src, int Cc, bool premul, bool src_full) {
        if (!dst) return;
        if (em.n_jobs == RA_MAX_MAP_JOBS) { launch_emit_maps(em, s); em.n_jobs = 0; total = 0; }
        total += (long long)P * Cc;
        em.job[em.n_jobs] = MapJob{src, dst, Cc, premul ? 1 : 0, src_full ? 1 : 0};
        em.end[em.n_jobs++] = total;
    };
    scat(out->acc, acc, 1, false, true);
    scat(out->depth, depth, 1, pm, true);
    scat(out->surf, surf, 3, pm, true);
    scat(out->ray_o, ray_o, 3, false, true);
    scat(out->norm, m.norm, 3, pm, false);
    scat(out->cpts, m.cpts, 3, pm, false);
    scat(out->bpts, m.bpts, 3, pm, false);
    scat(out->resd, m.resd, 3, false, false);
    scat(out->rgb, m.rgb, 3, pm, false);
    if (c->cfg.relight) {
        scat(out->albedo, m.albedo, 3, pm, false);
        scat(out->roughness, m.rough, 1, pm, false);
        scat(out->volume_albedo, m.valbedo, 3, false, false);
        scat(out->volume_roughness, m.rough, 1, false, false);
    }
    scat(out->raw, raw, S * C, false, false);
    if (relit) {
        scat(out->shade, shade, 3, pm, false);
        scat(out->spec, spec, 3, pm, false);
        scat(out->lvis, lvis, L, pm, false);
        scat(out->ldot, ldot, L, pm, false);
    }
    launch_emit_maps(em, s);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_render_ground_chunk(ra_ctx* c, const float* ray_o, const float* ray_d, const float* acc, int P, const float* bbox,
                           const float* probe, int ph, int pw, const ra_ground_params* p, const ra_ground_out* out, void* stream) {
    if (check_ready(c, "ra_render_ground_chunk")) return 1;
    RA_CHECK(p && out && P >= 0, "ra_render_ground_chunk: bad arguments");
    if (P == 0) return 0;
    RA_CHECK(ray_o && ray_d && acc && bbox && probe, "ra_render_ground_chunk: null argument");
    RA_CHECK(c->cfg.relight && c->n_lights > 0, "ra_render_ground_chunk: needs the relight network's light set");
    RA_CHECK((long long)P * c->n_lights < (1ll << 31), "ra_render_ground_chunk: chunk too large (P x lights must fit an int)");
    if (p->n_boxes > 1) {
        RA_CHECK(p->n_boxes <= RA_MAX_BOXES && p->boxes && p->box_start, "ra_render_ground_chunk: at most 32 boxes per call, with their tables");
        RA_CHECK(p->box_start[0] == 0 && p->box_start[p->n_boxes] == P, "ra_render_ground_chunk: box_start must run from 0 to P");
        for (int j = 0; j < p->n_boxes; ++j) RA_CHECK(p->box_start[j] <= p->box_start[j + 1], "ra_render_ground_chunk: box_start must ascend");
    }
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    zero_chunk_counters(c, s);
    HintScope hints(c, s);
    c->cnt_zero = false;                      // launch_ground_hit zeroes its hit counter itself
    GroundIn g{};
    g.ray_o = ray_o; g.ray_d = ray_d; g.acc = acc; g.P = P;
    const float nn = std::sqrt(p->normal[0] * p->normal[0] + p->normal[1] * p->normal[1] + p->normal[2] * p->normal[2]) + 1e-8f;   // normalize(): x / (|x| + eps)
    for (int k = 0; k < 3; ++k) { g.n[k] = p->normal[k] / nn; g.orig[k] = p->origin[k]; g.albedo[k] = p->albedo[k]; }
    g.attach_envmap = p->attach_envmap; g.env_r = p->env_r; g.shading_multiplier = p->shading_multiplier;
    float* t = c->buf<float>("gd_t", P, &err);
    float* surf = c->buf<float>("gd_surf", (size_t)P * 3, &err);
    float* depth = c->buf<float>("gd_depth", P, &err);
    float* nslots = c->buf<float>("gd_norm", (size_t)P * 3, &err);
    int* hit_idx = c->buf<int>("gd_hit", P, &err);
    if (err) return 1;
    int* hit_count = icnt(c, CNT_HIT);
    launch_ground_hit(g, t, surf, depth, nslots, hit_idx, hit_count, s);
    {   // spatially coherent order of the traced pixels (results are written back per pixel, so order-free)
        const size_t tb = sort_hits_temp_bytes(P);
        unsigned* k0 = c->buf<unsigned>("hs_k0", P, &err);
        unsigned* k1 = c->buf<unsigned>("hs_k1", P, &err);
        int* v0 = c->buf<int>("hs_v0", P, &err);
        char* tmp = c->buf<char>("hs_tmp", tb + 16, &err);
        if (err) return 1;
        const float bmin[3] = {bbox[0], bbox[1], bbox[2]};
        if (launch_sort_hits(surf, acc, P, bmin, k0, k1, v0, hit_idx, tmp, tb, s)) { ra_set_error("ra_render_ground_chunk: radix sort failed"); return 1; }
    }
    float *lvis = nullptr, *ldot = nullptr;
    if (key_mask_from(c, probe, ph, pw, s)) return 1;
    if (light_visibility_stage(c, surf, nslots, acc, hit_idx, hit_count, P, bbox, p->shadow_near_offset, p->shadow, p->no_visibility,
                               p->local_visibility, &lvis, &ldot, s, p->n_boxes, p->boxes, p->box_start)) return 1;
    auto zero = [&](void* dst, int C) { if (dst) hipMemsetAsync(dst, 0, (size_t)P * C * sizeof(float), s); };
    zero(out->rgb, 3); zero(out->albedo, 3); zero(out->shade, 3); zero(out->spec, 3);
    zero(out->lvis, c->n_lights); zero(out->ldot, c->n_lights);
    GroundShade in{};
    in.g = g; in.t = t; in.surf = surf; in.hit_idx = hit_idx; in.hit_count = hit_count; in.lvis = lvis;
    in.ldir = c->light_dir.as<float>(); in.light_area = c->light_area.as<float>(); in.L = c->n_lights;
    in.probe = probe; in.ph = ph; in.pw = pw;
    in.rgb = (float*)out->rgb; in.albedo = (float*)out->albedo; in.shade = (float*)out->shade; in.spec = (float*)out->spec;
    in.lvis_out = (float*)out->lvis; in.ldot_out = (float*)out->ldot;
    launch_ground_shade(in, c->cfg, s);
    if (out->surf) RA_HIP(hipMemcpyAsync(out->surf, surf, (size_t)P * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out->depth) RA_HIP(hipMemcpyAsync(out->depth, depth, (size_t)P * sizeof(float), hipMemcpyDeviceToDevice, s));
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_gather_rays(int device, const long long* idx, int n, const float* ray_o, const float* ray_d, const float* near_, const float* far_,
                   float* out_o, float* out_d, float* out_near, float* out_far, void* stream) {
    RA_CHECK(n >= 0 && (n == 0 || (idx && ray_o && ray_d && near_ && far_ && out_o && out_d && out_near && out_far)), "ra_gather_rays: bad arguments");
    RA_HIP(hipSetDevice(device));
    launch_gather_shard_rays(idx, n, ray_o, ray_d, near_, far_, out_o, out_d, out_near, out_far, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_scatter_rows(int device, const float* src, const long long* src_idx, const long long* dst_idx, long long n, int C, float* dst, void* stream) {
    RA_CHECK(n >= 0 && C > 0 && (n == 0 || (src && src_idx && dst_idx && dst)) && n * C < (1ll << 40), "ra_scatter_rows: bad arguments");
    RA_HIP(hipSetDevice(device));
    launch_scatter_rows(src, src_idx, dst_idx, n, C, dst, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_blend_ground(ra_ctx* c, const float* ground, const float* human, const long long* inds, const float* acc, int F, int P, int C,
                    float* dst, void* stream) {
    RA_CHECK(c, "ra_blend_ground: null ctx");
    RA_CHECK(acc && dst && F >= 0 && P >= 0 && C > 0 && (!human || inds), "ra_blend_ground: bad arguments");
    RA_HIP(hipSetDevice(c->device));
    launch_blend_ground(ground, human, inds, acc, F, P, C, dst, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_render_volume_chunk(ra_ctx* c, const float* ray_o, const float* ray_d, const float* near_, const float* far_, int P,
                           int n_samples, float dist_th, const ra_render_out* out, void* stream) {
    if (check_ready(c, "ra_render_volume_chunk")) return 1;
    RA_CHECK(out && P >= 0 && n_samples >= 1, "ra_render_volume_chunk: bad arguments");
    if (P == 0) return 0;
    RA_CHECK(ray_o && ray_d && near_ && far_, "ra_render_volume_chunk: null ray arrays");
    RA_CHECK(!c->cfg.relight, "ra_render_volume_chunk: volume rendering is wired for the AniSDF network (base_renderer)");
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    zero_chunk_counters(c, s);
    HintScope hints(c, s);
    const int S = n_samples, C = 16;
    const size_t N = (size_t)((P + 63) & ~63) * S;          // samples are laid out per group of 64 rays (padded)
    RA_CHECK(N < (1u << 30), "ra_render_volume_chunk: chunk too large");
    // spatially coherent ray order (Morton code of the entry point); outputs go back through the permutation
    const size_t tb = sort_hits_temp_bytes(P);
    unsigned* k0 = c->buf<unsigned>("hs_k0", P, &err);
    unsigned* k1 = c->buf<unsigned>("hs_k1", P, &err);
    int* v0 = c->buf<int>("hs_v0", P, &err);
    int* perm = c->buf<int>("rs_perm", P, &err);
    char* tmp = c->buf<char>("hs_tmp", tb + 16, &err);
    float* so = c->buf<float>("rs_o", (size_t)P * 3, &err);
    float* sd = c->buf<float>("rs_d", (size_t)P * 3, &err);
    float* sn = c->buf<float>("rs_n", P, &err);
    float* sf = c->buf<float>("rs_f", P, &err);
    float* xs = c->buf<float>("vl_x", N * 3, &err);
    float* vs = c->buf<float>("vl_v", N * 3, &err);
    float* raw = c->buf<float>("vl_raw", N * C, &err);
    if (err) return 1;
    if (launch_sort_rays(ray_o, ray_d, near_, far_, P, nullptr, k0, k1, v0, perm, tmp, tb, so, sd, sn, sf, s, c->cfg.clip_near, c->cfg.clip_far)) { ra_set_error("ra_render_volume_chunk: radix sort failed"); return 1; }
    launch_volume_samples(so, sd, sn, sf, P, S, xs, vs, s);
    if (forward_pass(c, xs, vs, (int)N, nullptr, dist_th, raw, s)) return 1;
    launch_volume_composite(raw, C, sn, sf, P, S, c->cfg.bg_brightness, *out, perm, s);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_reshade(ra_ctx* c, const float* ray_o, const float* surf, const float* norm, const float* albedo, const float* roughness,
               const float* lvis, const float* ldot, int P, const float* probes, int n_probes, int ph, int pw, float* rgb,
               float* shade, float* spec, void* stream) {
    RA_CHECK(c && c->have_weights && c->cfg.relight, "ra_reshade: needs a relight ctx with weights");
    RA_CHECK(P >= 0 && n_probes >= 0, "ra_reshade: bad sizes");
    if (P == 0 || n_probes == 0) return 0;
    RA_CHECK(ray_o && surf && norm && albedo && roughness && lvis && ldot && probes, "ra_reshade: null input");
    RA_HIP(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    for (int q0 = 0; q0 < n_probes; q0 += 8) {
        const int nq = n_probes - q0 < 8 ? n_probes - q0 : 8;
        ShadeIn in{};
        in.ray_o = ray_o; in.surf = surf; in.idx = nullptr; in.count = nullptr; in.n = P;
        in.norm = norm; in.albedo = albedo; in.rough = roughness; in.lvis = lvis; in.ldot = ldot;
        in.light_xyz = c->light_xyz.as<float>(); in.light_area = c->light_area.as<float>(); in.L = c->n_lights;
        in.probes = probes + (size_t)q0 * ph * pw * 3; in.n_probes = nq; in.ph = ph; in.pw = pw; in.want_spec = spec != nullptr;
        in.rgb = rgb ? rgb + (size_t)q0 * P * 3 : nullptr;
        in.shade = shade ? shade + (size_t)q0 * P * 3 : nullptr;
        in.spec = spec ? spec + (size_t)q0 * P * 3 : nullptr;
        ra_config cfg = c->cfg;
        cfg.tonemapping = 1;      // novel_light_sphere_tracing.py:47 applies linear2srgb unconditionally
        cfg.only_visibility = 0;  // ... and knows none of render_human's debugging switches (:21-66): it shades with the cosines and probes it is given
        cfg.vis_shade_map = 0;
        launch_shade(in, cfg, s);
    }
    c->n_shaded += (uint64_t)P * n_probes;
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_reshade_ground(ra_ctx* c, const float* ray_d, const float* albedo_map, const float* lvis, const float* ldot, int P,
                      const float* probes, int n_probes, int ph, int pw, const float* images, int ih, int iw, int attach_envmap,
                      float* rgb, float* albedo, float* shade, float* spec, void* stream) {
    RA_CHECK(c && c->have_weights && c->cfg.relight, "ra_reshade_ground: needs a relight ctx with weights");
    RA_CHECK(P >= 0 && n_probes >= 0, "ra_reshade_ground: bad sizes");
    if (P == 0 || n_probes == 0) return 0;
    RA_CHECK(ray_d && lvis && ldot && probes && ph > 0 && pw > 0, "ra_reshade_ground: null input");
    RA_CHECK(attach_envmap || albedo_map, "ra_reshade_ground: albedo_map is needed when the probe is not attached to the ground");
    RA_CHECK(!images || (ih > 0 && iw > 0), "ra_reshade_ground: bad image size");
    RA_HIP(hipSetDevice(c->device));
    GroundReshade in{};
    in.ray_d = ray_d; in.albedo_map = albedo_map; in.lvis = lvis; in.ldot = ldot;
    in.ldir = c->light_dir.as<float>(); in.light_area = c->light_area.as<float>(); in.L = c->n_lights;
    in.probes = probes; in.n_probes = n_probes; in.ph = ph; in.pw = pw; in.images = images; in.ih = ih; in.iw = iw;
    in.attach_envmap = attach_envmap; in.P = P;
    in.rgb = rgb; in.albedo = albedo; in.shade = shade; in.spec = spec;
    launch_ground_reshade(in, (hipStream_t)stream);
    c->n_shaded += (uint64_t)P * n_probes;
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_get_counters(ra_ctx* c, ra_counters* out, void* stream) {
    RA_CHECK(c && out, "ra_get_counters: null argument");
    RA_HIP(hipSetDevice(c->device));
    RA_HIP(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long h[8];
    RA_HIP(hipMemcpy(h, c->dcounters.p, sizeof(h), hipMemcpyDeviceToHost));
    out->n_coarse = h[0];
    out->n_fine_sdf = h[1];
    out->n_fine_full = h[2];
    out->n_shadow_rays = h[3];
    out->n_hit_pixels = h[4];
    out->n_shaded = c->n_shaded + h[4];
    out->n_fine_sdf_wide = h[5];
    out->n_fine_sdf_comp = h[6];
    return 0;
}

int ra_reset_counters(ra_ctx* c, void* stream) {
    RA_CHECK(c, "ra_reset_counters: null ctx");
    RA_HIP(hipSetDevice(c->device));
    RA_HIP(hipStreamSynchronize((hipStream_t)stream));
    RA_HIP(hipMemset(c->dcounters.p, 0, 64));
    c->n_coarse = 0;
    c->n_shaded = 0;
    c->ev_used = 0;
    return 0;
}

int ra_set_knn_mode(ra_ctx* c, int use_bvh) {
    RA_CHECK(c, "ra_set_knn_mode: null ctx");
    c->use_bvh = use_bvh != 0;
    c->have_frame = false;      // takes effect at the next ra_set_frame
    return 0;
}

// pinned staging ring of the context (ra_ctx.hpp PinRing)
static char* pin_acquire(ra_ctx* c, size_t bytes, int* slot) {
    PinRing& r = c->pin;
    if (bytes > r.slot_bytes) {
        if (r.base) {
            for (int k = 0; k < PinRing::n; ++k) if (r.used[k]) { hipEventSynchronize(r.ev[k]); r.used[k] = false; }
            hipHostFree(r.base);
            r.base = nullptr;
        }
        const size_t sb = (bytes + 4095) & ~(size_t)4095;
        if (hipHostMalloc((void**)&r.base, sb * PinRing::n, hipHostMallocDefault) != hipSuccess) { r.base = nullptr; r.slot_bytes = 0; return nullptr; }
        r.slot_bytes = sb;
        for (int k = 0; k < PinRing::n; ++k) if (!r.ev[k]) hipEventCreateWithFlags(&r.ev[k], hipEventDisableTiming);
    }
    const int k = r.next;
    r.next = (k + 1) % PinRing::n;
    if (r.used[k]) hipEventSynchronize(r.ev[k]);        // only when the host is PinRing::n frames ahead of this stream
    *slot = k;
    return r.base + (size_t)k * r.slot_bytes;
}
static void pin_release(ra_ctx* c, int slot, hipStream_t s) {
    hipEventRecord(c->pin.ev[slot], s);
    c->pin.used[slot] = true;
}

int ra_pose_frame(ra_ctx* c, const ra_pose_in* in, const ra_pose_out* out, void* stream) {
    RA_CHECK(c && in && out, "ra_pose_frame: null argument");
    const int J = in->n_bones, N = in->n_verts, F = in->n_faces;
    RA_CHECK(J > 0 && J <= 256 && N > 0 && F >= 0, "ra_pose_frame: bad sizes");
    RA_CHECK(in->poses && in->tjoints && in->parents && in->big_A && in->Rh && in->Th && in->tverts && in->weights, "ra_pose_frame: null input");
    RA_CHECK(!out->pnorm || (in->faces && F > 0), "ra_pose_frame: vertex normals need faces");
    hipStream_t s = (hipStream_t)stream;
    RA_HIP(hipSetDevice(c->device));
    for (int j = 1; j < J; ++j) RA_CHECK(in->parents[j] >= 0 && in->parents[j] < j, "ra_pose_frame: parents must be in topological order");
    // ---- the frame's small host inputs: ONE pinned block, ONE asynchronous upload; the bone transforms themselves (52 Rodrigues
    // rotations + the chain of 4 x 4 products, float64) run on the device behind it.  Nothing here waits for the stream: with frames
    // in flight an animated sequence poses frame f + 1 while frame f renders (round 3 computed the chain on the host and ended in a
    // hipStreamSynchronize: a full host stall per animated frame).
    const size_t n_in = (size_t)J * (3 + 3 + 16 + 1) + 6;
    int err = 0;
    float* dIn = c->buf<float>("pf_in", n_in, &err);
    float* dA = c->buf<float>("pf_A", (size_t)J * 16, &err);
    float* dJ = c->buf<float>("pf_J", (size_t)J * 3, &err);
    float* dR = c->buf<float>("pf_R", 12, &err);
    float* dP = c->buf<float>("pf_p", (size_t)N * 3, &err);
    float* dW = c->buf<float>("pf_w", (size_t)N * 3, &err);
    RA_CHECK(!err, "ra_pose_frame: out of device memory");
    {
        int slot = 0;
        float* st = reinterpret_cast<float*>(pin_acquire(c, n_in * sizeof(float), &slot));
        RA_CHECK(st, "ra_pose_frame: no pinned host memory for the staging ring");
        std::memcpy(st, in->poses, (size_t)J * 12);
        std::memcpy(st + 3 * J, in->tjoints, (size_t)J * 12);
        std::memcpy(st + 6 * J, in->big_A, (size_t)J * 64);
        std::memcpy(st + 22 * J, in->Rh, 12);
        std::memcpy(st + 22 * J + 3, in->Th, 12);
        std::memcpy(st + 22 * J + 6, in->parents, (size_t)J * 4);
        RA_HIP(hipMemcpyAsync(dIn, st, n_in * sizeof(float), hipMemcpyHostToDevice, s));
        pin_release(c, slot, s);
    }
    launch_bone_transforms(dIn, J, dA, dJ, dR, s);
    const float* dB = dIn + 6 * J;             // big_A as uploaded
    float* pv = out->pverts ? (float*)out->pverts : dP;
    float* wv = out->wverts ? (float*)out->wverts : dW;
    launch_lbs_verts((const float*)in->tverts, (const float*)in->weights, dA, dB, dR, dR + 9, N, J, (float*)out->tpose, pv, wv, s);
    if (out->pnorm) {
        // incident corners per vertex in index_add order, cached per (faces pointer, count)
        // the cache key is the CONTENT of the face array (a multiply-xorshift mix over 8-byte words, four independent lanes: ~10 us for
        // SMPL's 13 776 faces; FNV-1a byte by byte took 40 us of host time per frame)
        unsigned long long fh = 1469598103934665603ull ^ (unsigned long long)F;
        {
            const size_t nw = (size_t)F * 3 / 2;
            unsigned long long lane[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
            size_t k = 0;
            for (; k + 4 <= nw; k += 4)
                for (int l = 0; l < 4; ++l) {
                    unsigned long long w;
                    std::memcpy(&w, reinterpret_cast<const char*>(in->faces) + (k + l) * 8, 8);
                    lane[l] = (lane[l] ^ w) * 0x100000001b3ull;
                    lane[l] ^= lane[l] >> 29;
                }
            for (; k < nw; ++k) {
                unsigned long long w;
                std::memcpy(&w, reinterpret_cast<const char*>(in->faces) + k * 8, 8);
                lane[0] = (lane[0] ^ w) * 0x100000001b3ull;
                lane[0] ^= lane[0] >> 29;
            }
            if ((size_t)F * 3 % 2) lane[1] = (lane[1] ^ (unsigned)in->faces[3 * F - 1]) * 0x100000001b3ull;
            for (int l = 0; l < 4; ++l) { fh = (fh ^ lane[l]) * 1099511628211ull; fh ^= fh >> 31; }
        }
        if (c->adj_hash != fh || c->adj_n_faces != F || c->adj_n_verts != N) {
            std::vector<int> start(N + 1, 0), adj((size_t)F * 3);
            const int order[3] = {1, 2, 0};
            for (int f = 0; f < F; ++f)
                for (int k = 0; k < 3; ++k) {
                    const int v = in->faces[3 * f + k];
                    RA_CHECK(v >= 0 && v < N, "ra_pose_frame: face index out of range");
                    ++start[v + 1];
                }
            for (int v = 0; v < N; ++v) start[v + 1] += start[v];
            std::vector<int> fill(start.begin(), start.end() - 1);
            for (int pass = 0; pass < 3; ++pass)
                for (int f = 0; f < F; ++f) { const int corner = order[pass]; adj[fill[in->faces[3 * f + corner]]++] = (f << 2) | corner; }
            // a new mesh (rare): a vert_normals launch of an earlier frame may still be queued on s (frames-in-flight streams are
            // non-blocking: not ordered against the null stream the copies below run on) and would read a half-overwritten list
            RA_HIP(hipStreamSynchronize(s));
            if (c->adj_start.ensure((size_t)(N + 1) * 4) || c->adj_list.ensure((size_t)F * 12 + 4) || c->adj_dfaces.ensure((size_t)F * 12 + 4)) return 1;
            RA_HIP(hipMemcpy(c->adj_start.p, start.data(), (size_t)(N + 1) * 4, hipMemcpyHostToDevice));
            RA_HIP(hipMemcpy(c->adj_list.p, adj.data(), (size_t)F * 12, hipMemcpyHostToDevice));
            RA_HIP(hipMemcpy(c->adj_dfaces.p, in->faces, (size_t)F * 12, hipMemcpyHostToDevice));
            c->adj_hash = fh; c->adj_n_faces = F; c->adj_n_verts = N;
        }
        launch_vert_normals(pv, c->adj_dfaces.as<int>(), c->adj_start.as<int>(), c->adj_list.as<int>(), N, (float*)out->pnorm, s);
    }
    if (out->pbounds) launch_bounds(pv, N, in->bounds_padding, (float*)out->pbounds, s);
    if (out->wbounds) launch_bounds(wv, N, in->bounds_padding, (float*)out->wbounds, s);
    if (out->A) RA_HIP(hipMemcpyAsync(out->A, dA, (size_t)J * 64, hipMemcpyDeviceToDevice, s));
    if (out->R) RA_HIP(hipMemcpyAsync(out->R, dR, 36, hipMemcpyDeviceToDevice, s));
    if (out->joints) RA_HIP(hipMemcpyAsync(out->joints, dJ, (size_t)J * 12, hipMemcpyDeviceToDevice, s));
    if (out->poses) RA_HIP(hipMemcpyAsync(out->poses, dIn, (size_t)J * 12, hipMemcpyDeviceToDevice, s));
    if (out->Th) RA_HIP(hipMemcpyAsync(out->Th, dIn + 22 * J + 3, 12, hipMemcpyDeviceToDevice, s));
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_grow_bounds(ra_ctx* c, float* wbounds, float margin, void* stream) {
    RA_CHECK(c && wbounds, "ra_grow_bounds: null argument");
    RA_HIP(hipSetDevice(c->device));
    launch_grow_bounds(wbounds, margin, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_shift_envmap(ra_ctx* c, const float* img, int H, int W, int C, float shift, float* out, void* stream) {
    RA_CHECK(c, "ra_shift_envmap: null ctx");
    RA_CHECK(img && out && H > 0 && W > 0 && C > 0 && img != out, "ra_shift_envmap: bad arguments");
    RA_HIP(hipSetDevice(c->device));
    launch_shift_envmap(img, H, W, C, shift, out, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_add_light_probe(ra_ctx* c, float* rgb, int H, int W, const float* probe, int ph, int pw, const float* cam_R, int uH, int uW,
                       void* stream) {
    RA_CHECK(c, "ra_add_light_probe: null ctx");
    RA_CHECK(rgb && probe && cam_R && H > 0 && W > 0 && ph > 0 && pw > 0, "ra_add_light_probe: bad arguments");
    RA_CHECK(uH >= 0 && uW >= 0 && uH <= H && uW <= W, "ra_add_light_probe: the inset does not fit the image");
    RA_HIP(hipSetDevice(c->device));
    // gen_light_dir (relight_utils.py:9-30): camera axes (columns of R^T) with only the horizontal heading kept
    const double front0[3] = {cam_R[6], cam_R[7], cam_R[8]};             // third row of the w2c rotation = camera z in the world
    const double downz = cam_R[5] > 0 ? 1.0 : (cam_R[5] < 0 ? -1.0 : 0.0);   // sign of (camera y).z
    const double down[3] = {0.0, 0.0, downz};
    auto cross = [](const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    auto norml = [](double* v) { const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-8; v[0] /= n; v[1] /= n; v[2] /= n; };
    double right[3], front[3];
    cross(down, front0, right); norml(right);
    cross(right, down, front); norml(front);
    ProbeInset p{};
    for (int r = 0; r < 3; ++r) { p.axes[3 * r] = (float)right[r]; p.axes[3 * r + 1] = (float)-front[r]; p.axes[3 * r + 2] = (float)-down[r]; }
    p.H = H; p.W = W; p.uH = uH; p.uW = uW; p.ph = ph; p.pw = pw;
    launch_light_probe(p, probe, rgb, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_map_to_image(ra_ctx* c, const ra_image_params* p, const float* a, const float* b, const float* acc, const long long* pix, int P,
                    float* image, float* alpha, void* stream) {
    RA_CHECK(c && p && image, "ra_map_to_image: null argument");
    RA_CHECK(p->H > 0 && p->W > 0 && P >= 0 && (long long)p->H * p->W < (1ll << 30), "ra_map_to_image: bad sizes");
    RA_CHECK(p->type >= RA_IMG_SURFACE && p->type <= RA_IMG_RENDERING, "ra_map_to_image: unknown output type");
    RA_CHECK(P == 0 || a || p->type == RA_IMG_ALPHA, "ra_map_to_image: the map is missing");
    RA_CHECK(pix || P == p->H * p->W || P == 0, "ra_map_to_image: without pixel indices the maps must be full-frame");
    RA_CHECK(p->type != RA_IMG_RESIDUAL || b, "ra_map_to_image: Residual needs cpts_map and bpts_map");
    RA_CHECK((p->type != RA_IMG_SURFACE && p->type != RA_IMG_NORMAL && p->type != RA_IMG_ALPHA && p->type != RA_IMG_DEPTH) || acc || P == 0,
             "ra_map_to_image: this type needs acc_map");
    hipStream_t s = (hipStream_t)stream;
    RA_HIP(hipSetDevice(c->device));
    int err = 0;
    float* stats = c->buf<float>("im_stats", 4, &err);
    const bool pct_all = p->type == RA_IMG_RESIDUAL || ((p->type == RA_IMG_SHADING || p->type == RA_IMG_SPECULAR) && p->normalize);
    if (P > 0 && (pct_all || p->type == RA_IMG_DEPTH)) {
        const long long n = pct_all ? 3ll * P : P;
        const int k = (int)((pct_all ? 0.005 : 0.01) * (double)n);                    // int(percentile * depth_map.numel())
        RA_CHECK(k >= 1, "ra_map_to_image: too few rays for the percentile (the reference's topk(0).max() fails too)");
        const size_t tb = image_sort_temp_bytes(n);
        float* sa = c->buf<float>("im_sa", n, &err);
        float* sb = c->buf<float>("im_sb", n, &err);
        unsigned char* flag = c->buf<unsigned char>("im_flag", n, &err);
        char* tmp = c->buf<char>("im_tmp", tb + 16, &err);
        RA_CHECK(!err, "ra_map_to_image: out of device memory");
        const float* vals = a;
        if (p->type == RA_IMG_RESIDUAL) { launch_diff(a, b, n, sa, s); vals = sa; sa = c->buf<float>("im_sc", n, &err); RA_CHECK(!err, "ra_map_to_image: out of device memory"); }
        RA_CHECK(launch_percentiles(vals, n, p->type == RA_IMG_DEPTH ? acc : nullptr, k, sa, sb, flag, icnt(c, CNT_SAMP), tmp, tb, stats, s) == 0,
                 "ra_map_to_image: device sort failed");
    }
    RA_CHECK(!err, "ra_map_to_image: out of device memory");
    ImageJob j{};
    j.type = p->type; j.P = P; j.a = a; j.b = b; j.acc = acc; j.pix = pix; j.stats = stats;
    for (int k = 0; k < 9; ++k) j.cam_R[k] = p->cam_R[k];
    for (int k = 0; k < 6; ++k) j.tbounds[k] = p->tbounds[k];
    j.min_clip = p->min_clip; j.bg = p->bg_brightness; j.normalize = p->normalize; j.tonemap = p->tonemap;
    j.image = image; j.alpha = alpha;
    launch_compose_image(j, (long long)p->H * p->W, s);
    RA_HIP(hipGetLastError());
    return 0;
}

static void inv3x3(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    o[0] = (e * i - f * h) / det; o[1] = (c * h - b * i) / det; o[2] = (b * f - c * e) / det;
    o[3] = (f * g - d * i) / det; o[4] = (a * i - c * g) / det; o[5] = (c * d - a * f) / det;
    o[6] = (d * h - e * g) / det; o[7] = (b * g - a * h) / det; o[8] = (a * e - b * d) / det;
}

int ra_gen_rays(ra_ctx* c, int H, int W, const double* K, const double* R, const double* T, const float* bounds, const float* bounds_dev,
                void* ray_o, void* ray_d, void* near, void* far, void* mask_at_box, int* n_rays, int* n_rays_dev, void* stream) {
    RA_CHECK(c, "ra_gen_rays: null ctx");
    RA_CHECK(H > 0 && W > 0 && (long long)H * W < (1ll << 30), "ra_gen_rays: bad image size");
    RA_CHECK(K && R && T && (bounds || bounds_dev) && ray_o && ray_d && near && far && mask_at_box, "ra_gen_rays: null argument");
    hipStream_t s = (hipStream_t)stream;
    RA_HIP(hipSetDevice(c->device));
    RayCam cam;
    inv3x3(K, cam.Kinv);
    for (int k = 0; k < 9; ++k) cam.R[k] = R[k];
    for (int k = 0; k < 3; ++k) {
        cam.T[k] = T[k];
        cam.o[k] = -(R[k] * T[0] + R[3 + k] * T[1] + R[6 + k] * T[2]);      // -R^T T
        cam.bmin[k] = bounds ? bounds[k] : 0.f;
        cam.bmax[k] = bounds ? bounds[3 + k] : 0.f;
    }
    cam.H = H; cam.W = W;
    cam.bdev = bounds_dev;
    const int n = H * W;
    int err = 0;
    const size_t tb = gen_rays_temp_bytes(n);
    int* pix = c->buf<int>("ray_pix", (size_t)n + 1, &err);
    void* temp = c->buf<char>("ray_tmp", tb ? tb : 16, &err);
    RA_CHECK(!err, "ra_gen_rays: out of device memory");
    int* count_dev = pix + n;
    RA_CHECK(launch_gen_rays(cam, (unsigned char*)mask_at_box, pix, count_dev, temp, tb, (float*)ray_o, (float*)ray_d, (float*)near,
                             (float*)far, s) == 0, "ra_gen_rays: device selection failed");
    if (n_rays_dev) RA_HIP(hipMemcpyAsync(n_rays_dev, count_dev, sizeof(int), hipMemcpyDeviceToDevice, s));     // for a caller that reads it back later
    if (n_rays) {           // the count on the host costs a synchronisation; a caller that knows it (an unbounded box: H * W) passes NULL
        RA_HIP(hipMemcpyAsync(n_rays, count_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        RA_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

int ra_enable_timing(ra_ctx* c, int on) {
    RA_CHECK(c, "ra_enable_timing: null ctx");
    c->timing = on != 0;
    return 0;
}

int ra_get_kernel_time(ra_ctx* c, int kind, float* ms, int* n_launches, void* stream) {
    RA_CHECK(c && ms && n_launches, "ra_get_kernel_time: null argument");
    RA_CHECK(kind >= 0 && kind <= 4, "ra_get_kernel_time: kind must be 0 (distance query), 1 (full query), 2 (8-wave distance query), 3 (narrow distance query) or 4 (compensated distance query)");
    RA_HIP(hipSetDevice(c->device));
    RA_HIP(hipStreamSynchronize((hipStream_t)stream));
    float tot = 0.f;
    int n = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        const int k = c->ev_kind[i];          // internal: 0 = 8-wave K3, 2 = narrow K3, 3 = K3C, 1 = K4
        if (!(kind == 0 ? (k == 0 || k == 2) : kind == 1 ? k == 1 : kind == 2 ? k == 0 : kind == 3 ? k == 2 : k == 3)) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, c->ev_pool[i].first, c->ev_pool[i].second) == hipSuccess) { tot += t; ++n; }
    }
    *ms = tot;
    *n_launches = n;
    return 0;
}

int ra_get_mlp_time(ra_ctx* c, float* ms, int* n_launches, void* stream) { return ra_get_kernel_time(c, 0, ms, n_launches, stream); }

// ---- test hooks: stage outputs for parity tests (not used by the renderers) -------------------
int ra_debug_mlp(ra_ctx* c, const float* bpts, int n, float* resd, float* sdf, float* feat, void* stream) {
    // stage outputs of the geometry networks from the PRODUCTION forward kernel of the full query (K4 forward with tape)
    if (check_ready(c, "ra_debug_mlp")) return 1;
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    int* idx = c->buf<int>("fine_idx", n, &err);
    char* tape = c->buf<char>("k4_tape", mlp_full_rev_tape_bytes(n), &err);
    if (err) return 1;
    int* cnt = next_fine_counter(c, s);
    launch_iota(idx, n, cnt, s);
    FullIO io{};
    io.bpts = bpts; io.idx = idx; io.count = cnt; io.slot0 = 0; io.slot_cap = n; io.C = c->cfg.relight ? 17 : 16;
    io.beta = c->host.beta; io.resd_limit = c->cfg.resd_limit; io.relight = c->cfg.relight;
    io.dbg_resd = resd; io.dbg_sdf = sdf; io.dbg_feat = feat; io.dbg_layer = -1;
    if (c->cfg.mlp_f16) launch_mlp_fwd_tape_f16(c->host.geo, c->fwd_arena.p, c->barena.as<float>(), c->fr, io, tape, s);
    else launch_mlp_fwd_tape_bf16(c->host.geo, c->fwd_arena.p, c->barena.as<float>(), c->fr, io, tape, s);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_debug_full(ra_ctx* c, const float* bpts, int n, float* grad, float* sdf, float* feat, float* raw, void* stream) {
    if (check_ready(c, "ra_debug_full")) return 1;
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    const int C = c->cfg.relight ? 17 : 16;
    int* cnt = next_fine_counter(c, s);
    int* idx = c->buf<int>("fine_idx", n, &err);
    float* view = c->buf<float>("dbg_view", (size_t)n * 3, &err);
    if (err) return 1;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    RA_HIP(hipMemcpyAsync(idx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    RA_HIP(hipMemcpyAsync(cnt, &n, sizeof(int), hipMemcpyHostToDevice, s));
    RA_HIP(hipMemsetAsync(view, 0, (size_t)n * 12, s));
    RA_HIP(hipStreamSynchronize(s));
    FullIO io{};
    io.bpts = bpts; io.mats = nullptr; io.view = view; io.idx = idx; io.count = cnt; io.raw = raw; io.C = C;
    io.beta = c->host.beta; io.resd_limit = c->cfg.resd_limit;
    io.albedo_slope = c->cfg.albedo_slope; io.albedo_bias = c->cfg.albedo_bias;
    io.rough_slope = c->cfg.roughness_slope; io.rough_bias = c->cfg.roughness_bias;
    io.relight = c->cfg.relight;
    io.dbg_grad = grad; io.dbg_sdf = sdf; io.dbg_feat = feat; io.counters = nullptr;
    io.dbg_layer = -1;
#ifdef RA_TESTING            // debugging aids of tools/dbg_grad.py (test builds only)
    if (getenv("RA_DBG_GC")) { io.dbg_gc = grad; io.dbg_grad = nullptr; }       // d sdf / d cpts instead
    if (getenv("RA_DBG_LAYER")) io.dbg_layer = atoi(getenv("RA_DBG_LAYER"));
    if (getenv("RA_DBG_PE")) { io.dbg_pe = feat; io.dbg_feat = nullptr; RA_HIP(hipMemsetAsync(feat, 0, (size_t)n * 256 * 4, s)); }   // encoding-slot gradients in feat[:, :128]
#endif
    if (full_query(c, io, n, s)) return 1;
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_debug_aabb(ra_ctx* c, const float* o, const float* d, int n, const float* bbox, float* nr, float* fr, void* stream) {
    RA_CHECK(c && n >= 0 && (n == 0 || (o && d && bbox && nr && fr)), "ra_debug_aabb: bad arguments");
    RA_HIP(hipSetDevice(c->device));
    launch_debug_aabb(o, d, n, bbox, nr, fr, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_debug_lvis(ra_ctx* c, const float* surf, const float* norm, const float* acc, int n, const float* bbox, const ra_trace_params* shadow,
                  float near_offset, float* lvis_out, float* ldot_out, void* stream) {
    if (check_ready(c, "ra_debug_lvis")) return 1;
    RA_CHECK(c->cfg.relight && c->n_lights > 0, "ra_debug_lvis: needs the relight network's light set");
    RA_CHECK(n >= 0 && shadow && (n == 0 || (surf && norm && acc && bbox && lvis_out && ldot_out)), "ra_debug_lvis: bad arguments");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    int* hit_idx = c->buf<int>("dl_hit", n, &err);
    if (err) return 1;
    launch_iota(hit_idx, n, icnt(c, CNT_HIT), s);          // every point is its own hit slot
    float *lvis = nullptr, *ldot = nullptr;
    if (key_mask_from(c, nullptr, 0, 0, s)) return 1;       // no probe here: every ray in the plain tier (unless ra_set_key_probes named the key lights)
    if (light_visibility_stage(c, surf, norm, acc, hit_idx, icnt(c, CNT_HIT), n, bbox, near_offset, *shadow, 0, 0, &lvis, &ldot, s)) return 1;
    RA_HIP(hipMemcpyAsync(lvis_out, lvis, (size_t)n * c->n_lights * 4, hipMemcpyDeviceToDevice, s));
    RA_HIP(hipMemcpyAsync(ldot_out, ldot, (size_t)n * c->n_lights * 4, hipMemcpyDeviceToDevice, s));
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_debug_brdf(ra_ctx* c, const float* p2l, const float* p2c, const float* normal, const float* albedo, const float* rough, int L, int N,
                  float* brdf, void* stream) {
    RA_CHECK(c && c->have_cfg, "ra_debug_brdf: call ra_set_config first");
    RA_CHECK(L >= 0 && N >= 0 && (L * N == 0 || (p2l && p2c && normal && albedo && rough && brdf)), "ra_debug_brdf: bad arguments");
    RA_HIP(hipSetDevice(c->device));
    launch_debug_brdf(p2l, p2c, normal, albedo, rough, L, N, c->cfg, brdf, (hipStream_t)stream);
    RA_HIP(hipGetLastError());
    return 0;
}

int ra_debug_bvh_ids(ra_ctx* c, int* ids_host, int capacity, int* n_out, void* stream) {
    if (check_ready(c, "ra_debug_bvh_ids")) return 1;
    RA_CHECK(ids_host && n_out && capacity >= 0, "ra_debug_bvh_ids: bad arguments");
    const int nleaf = c->fr.bvh_leaves;
    *n_out = nleaf * 32;
    if (nleaf == 0) return 0;
    RA_CHECK(capacity >= nleaf * 32, "ra_debug_bvh_ids: capacity too small");
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> leaves((size_t)nleaf * 128);
    RA_HIP(hipMemcpyAsync(leaves.data(), c->fr.bvh_soa, leaves.size() * 4, hipMemcpyDeviceToHost, s));
    RA_HIP(hipStreamSynchronize(s));
    for (int l = 0; l < nleaf; ++l) memcpy(ids_host + (size_t)l * 32, leaves.data() + (size_t)l * 128 + 96, 32 * 4);
    return 0;
}

int ra_debug_hdq(ra_ctx* c, const float* x, int n, float th, float* sdf_coarse, float* sdf_batch, int* nn_batch, float* d2,
                 float* bpts, float* tpts, float* mats, int* fine_count_host, void* stream) {
    if (check_ready(c, "ra_debug_hdq")) return 1;
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    int* fine_idx = c->buf<int>("fine_idx", n, &err);
    float* fb = c->buf<float>("fine_bpts", (size_t)n * 3, &err);
    if (err) return 1;
    RaySet rs{};
    rs.mode = 0; rs.x = x;
    HdqOut out{};
    out.sdf = sdf_coarse; out.fine_count = next_fine_counter(c, s); out.fine_idx = fine_idx; out.bpts = fb;
    out.dbg_sdf_batch = sdf_batch; out.dbg_nn_batch = nn_batch; out.dbg_d2 = d2; out.dbg_bpts = bpts; out.dbg_tpts = tpts; out.dbg_mats = mats;
    out.counters = dcnt(c);
    RA_HIP(hipMemsetAsync(bpts, 0, (size_t)n * 12, s));
    RA_HIP(hipMemsetAsync(tpts, 0, (size_t)n * 12, s));
    RA_HIP(hipMemsetAsync(mats, 0, (size_t)n * 96, s));
    launch_hdq_coarse(c->fr, rs, n, th, c->cfg.blend_radius, out, s, c->cfg.use_geodesic_filter != 0);
    RA_HIP(hipStreamSynchronize(s));
    RA_HIP(hipMemcpy(fine_count_host, out.fine_count, sizeof(int), hipMemcpyDeviceToHost));
    RA_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
