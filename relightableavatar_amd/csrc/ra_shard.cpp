// Host side of the ray sharding (relightableavatar_amd/shard.py make_plan): the per-frame deal of a frame's in-box rays to the ranks of
// one node, in ONE pass over the frame's mask instead of a dozen numpy passes (1.1-2.8 ms of Python per frame and rank were a quarter of
// a 4.5 ms frame at eight ranks).  No device work; the index vectors are written straight into the caller's (pinned) staging block.
// The reference has no multi-GPU inference (run.py is single-process): SURVEY.md 8e.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/relightableavatar.h"

// plain host C++ (no HIP header): tests/native/shard_fuzz.cpp compiles this file with g++ -fsanitize=address,undefined
void ra_set_error(const std::string& msg);

extern "C" int ra_shard_plan(const unsigned char* mask, int H, int W, int world, int ground, long long P, const long long* ground_pos,
                             const long long* edges, int n_edges, unsigned char* owner, long long* order, long long* src, long long* inds,
                             long long* counts, long long* chunk_pos, long long* n_max_out) {
    constexpr int TILE = 8;
    if (!mask || H <= 0 || W <= 0 || world <= 0 || world > 256 || P < 0 || !order || !src || !counts || !n_max_out || (n_edges > 0 && (!edges || !chunk_pos))) {
        ra_set_error("ra_shard_plan: bad arguments");
        return 1;
    }
    const int tx = (W + TILE - 1) / TILE, ty = (H + TILE - 1) / TILE;
    // 1. the rank of every 8 x 8 tile: with the ground pass fixed diagonal stripes over the whole frame (a rank's in-box pixels must be
    //    a subset of its ground pixels); without it the tiles that HOLD in-box pixels, dealt round robin in raster order
    std::vector<unsigned char> tile_rank((size_t)tx * ty);
    if (ground) {
        for (int j = 0; j < ty; ++j)
            for (int i = 0; i < tx; ++i) tile_rank[(size_t)j * tx + i] = (unsigned char)((j + i) % world);
    } else {
        std::vector<unsigned char> present((size_t)tx * ty, 0);
        for (int y = 0; y < H; ++y) {
            const unsigned char* row = mask + (size_t)y * W;
            unsigned char* pr = present.data() + (size_t)(y / TILE) * tx;
            int x = 0;
            for (int i = 0; i < W / TILE; ++i, x += TILE) {
                unsigned long long w;
                memcpy(&w, row + x, 8);
                pr[i] |= (unsigned char)(w != 0ull);
            }
            for (; x < W; ++x) pr[x / TILE] |= (unsigned char)(row[x] != 0);
        }
        long long seen = 0;
        for (size_t t = 0; t < present.size(); ++t) {
            if (present[t]) { tile_rank[t] = (unsigned char)(seen % world); ++seen; } else tile_rank[t] = 0;
        }
    }
    // 2. owners and counts (the in-box rays are the mask's pixels in row-major order), ranges of every rank's rays inside the chunks
    //    of the whole ray list (chunk_pos[r * n_edges + e] = how many of rank r's rays lie before ray edges[e])
    std::vector<long long> cnt(world, 0);
    std::vector<unsigned char> own_local;
    if ((long long)H * W > 0x7fffffffLL) { ra_set_error("ra_shard_plan: frame too large"); return 1; }
    unsigned char* own = owner;
    if (!own) { own_local.resize((size_t)(P > 0 ? P : 1)); own = own_local.data(); }
    // pass A: the frame pixel and the tile of every ray, eight pixels per test (most of a frame is outside the box)
    std::vector<int> pix((size_t)(P > 0 ? P : 1)), tile_of((size_t)(P > 0 ? P : 1));
    long long r = 0;
    {
        int* pp = pix.data();
        int* tt = tile_of.data();
        for (int y = 0; y < H; ++y) {
            const unsigned char* row = mask + (size_t)y * W;
            const int f0 = y * W, t0 = (y / TILE) * tx;
            int x = 0;
            for (; x + 8 <= W; x += 8) {
                unsigned long long w;
                memcpy(&w, row + x, 8);
                if (w == 0ull) continue;
                for (int k = 0; k < 8; ++k)
                    if (row[x + k]) {
                        if (r >= P) { ra_set_error("ra_shard_plan: the mask holds more pixels than P"); return 1; }
                        tt[r] = t0 + x / TILE;
                        pp[r++] = f0 + x + k;
                    }
            }
            for (; x < W; ++x)
                if (row[x]) {
                    if (r >= P) { ra_set_error("ra_shard_plan: the mask holds more pixels than P"); return 1; }
                    tt[r] = t0 + x / TILE;
                    pp[r++] = f0 + x;
                }
        }
    }
    if (r != P) { ra_set_error("ra_shard_plan: the mask holds fewer pixels than P"); return 1; }
    // pass B: owners and counts, chunk by chunk of the unsharded ray list
    {
        long long c8[256] = {0};
        long long i = 0;
        const unsigned char* tr = tile_rank.data();
        for (int e = 0; e <= n_edges; ++e) {
            const long long stop = e < n_edges ? (edges[e] < P ? edges[e] : P) : P;
            for (; i < stop; ++i) {
                const unsigned char o = tr[tile_of[i]];
                own[i] = o;
                ++c8[o];
            }
            if (e < n_edges) for (int k = 0; k < world; ++k) chunk_pos[(size_t)k * n_edges + e] = c8[k];
        }
        for (int k = 0; k < world; ++k) cnt[k] = c8[k];
    }
    // 3. the exchange's index vectors: order = rays grouped by owner (original order inside), src[j] = where item j of `order` sits
    //    in the all_gather's output (rank * n_max + position in the rank's shard)
    long long n_max = 0;
    std::vector<long long> offs(world + 1, 0);
    for (int k = 0; k < world; ++k) { counts[k] = cnt[k]; offs[k + 1] = offs[k] + cnt[k]; if (cnt[k] > n_max) n_max = cnt[k]; }
    *n_max_out = n_max;
    std::vector<long long> fill(offs.begin(), offs.end() - 1);
    if (inds && !ground_pos) { ra_set_error("ra_shard_plan: inds needs ground_pos"); return 1; }
    if (inds) for (long long i = 0; i < P; ++i) { const long long j = fill[own[i]]++; order[j] = i; inds[j] = ground_pos[pix[i]]; }
    else for (long long i = 0; i < P; ++i) order[fill[own[i]]++] = i;
    for (int k = 0; k < world; ++k)
        for (long long j = offs[k]; j < offs[k + 1]; ++j) src[j] = (long long)k * n_max + (j - offs[k]);
    // (with the ground pass, inds = where every rank's human rays sit in its list of ground pixels, was filled beside `order`: ground_pos
    // is per frame pixel its position in its owner's full-frame pixel list — a function of the frame size only, cached by the caller)
    return 0;
}
