// Sanitizer build of the library's host-side shard plan (relightableavatar_amd/csrc/ra_shard.cpp), fuzzed against its contract
// (include/relightableavatar.h ra_shard_plan).  Built and run by tests/test_host_logic.py::test_shard_plan_under_sanitizers:
//     g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tests/native/shard_fuzz.cpp -o shard_fuzz && ./shard_fuzz
// No GPU, no HIP: the file under test is plain C++.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

static std::string g_err;
void ra_set_error(const std::string& msg) { g_err = msg; }
#include "../../relightableavatar_amd/csrc/ra_shard.cpp"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "case %d (H %d W %d world %d ground %d P %lld): %s failed (%s)\n", it, H, W, world, ground, P, #c, g_err.c_str()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n_cases = argc > 1 ? std::atoi(argv[1]) : 3000;
    std::mt19937_64 rng(12345);
    auto uni = [&](int lo, int hi) { return (int)(lo + rng() % (unsigned long long)(hi - lo + 1)); };
    for (int it = 0; it < n_cases; ++it) {
        const int H = uni(1, 70), W = uni(1, 70), world = uni(1, 9), ground = uni(0, 1);
        const int dens = uni(0, 100);
        std::vector<unsigned char> mask((size_t)H * W);
        long long P = 0;
        // a box-shaped region plus noise, like a body's mask_at_box; sometimes empty, sometimes full
        const int y0 = uni(0, H - 1), y1 = uni(y0, H - 1), x0 = uni(0, W - 1), x1 = uni(x0, W - 1);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const bool in = y >= y0 && y <= y1 && x >= x0 && x <= x1;
                const bool v = dens == 0 ? false : (dens == 100 ? true : (in ? uni(0, 99) < 90 : uni(0, 99) < dens / 10));
                mask[(size_t)y * W + x] = v ? (unsigned char)uni(1, 255) : 0;       // any non-zero byte is "in the box"
                P += v;
            }
        // ground_pos: per frame pixel its position in its owner's full-frame pixel list (diagonal 8 x 8 stripes), as shard.py builds it
        const int tx = (W + 7) / 8;
        std::vector<long long> gpos, gcount(world, 0);
        if (ground) {
            gpos.resize((size_t)H * W);
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) gpos[(size_t)y * W + x] = gcount[((y / 8) + (x / 8)) % world]++;
        }
        (void)tx;
        const int n_edges = uni(0, 4);
        std::vector<long long> edges(n_edges);
        for (int e = 0; e < n_edges; ++e) edges[e] = P > 0 ? (long long)(rng() % (unsigned long long)(P + 1)) : 0;
        for (int a = 0; a < n_edges; ++a) for (int b = a + 1; b < n_edges; ++b) if (edges[b] < edges[a]) std::swap(edges[a], edges[b]);
        const size_t n = (size_t)(P > 0 ? P : 1);
        // exact-size heap blocks: ASan catches a write one element past any of them
        std::vector<unsigned char> owner(n);
        std::vector<long long> order(n), src(n), inds(n), counts(world), chunk_pos((size_t)world * (n_edges > 0 ? n_edges : 1));
        long long n_max = -1;
        const bool with_owner = uni(0, 1);
        g_err.clear();
        const int rc = ra_shard_plan(mask.data(), H, W, world, ground, P, ground ? gpos.data() : nullptr, n_edges ? edges.data() : nullptr, n_edges,
                                     with_owner ? owner.data() : nullptr, order.data(), src.data(), ground ? inds.data() : nullptr, counts.data(),
                                     n_edges ? chunk_pos.data() : nullptr, &n_max);
        CHECK(rc == 0);
        long long tot = 0, mx = 0;
        for (int k = 0; k < world; ++k) { CHECK(counts[k] >= 0); tot += counts[k]; if (counts[k] > mx) mx = counts[k]; }
        CHECK(tot == P && n_max == mx);
        // order: a permutation of the rays, grouped by owner, ascending inside a group; src: rank * n_max + position in the shard
        std::vector<char> seen(n, 0);
        std::vector<int> pix;                  // ray -> frame pixel
        for (size_t f = 0; f < mask.size(); ++f) if (mask[f]) pix.push_back((int)f);
        long long j = 0;
        for (int k = 0; k < world; ++k)
            for (long long q = 0; q < counts[k]; ++q, ++j) {
                const long long i = order[j];
                CHECK(i >= 0 && i < P && !seen[i]);
                seen[i] = 1;
                if (q) CHECK(order[j - 1] < i);
                if (with_owner) CHECK(owner[i] == k);
                CHECK(src[j] == (long long)k * n_max + q);
                const int f = pix[i], ty_ = (f / W) / 8, tx_ = (f % W) / 8;
                if (ground) {
                    CHECK((ty_ + tx_) % world == k);                       // a rank's in-box pixels are among its ground pixels
                    CHECK(inds[j] == gpos[f] && inds[j] < gcount[k]);
                }
            }
        CHECK(j == P);
        // rays of one 8 x 8 tile share an owner (the deal is by tile)
        if (with_owner)
            for (long long i = 1; i < P; ++i) {
                const int fa = pix[i - 1], fb = pix[i];
                if ((fa / W) / 8 == (fb / W) / 8 && (fa % W) / 8 == (fb % W) / 8) CHECK(owner[i - 1] == owner[i]);
            }
        // chunk_pos[r * n_edges + e] = how many of rank r's rays lie before ray edges[e]
        for (int e = 0; e < n_edges; ++e) {
            long long s = 0;
            for (int k = 0; k < world; ++k) {
                const long long v = chunk_pos[(size_t)k * n_edges + e];
                CHECK(v >= 0 && v <= counts[k] && (e == 0 || v >= chunk_pos[(size_t)k * n_edges + e - 1]));
                s += v;
            }
            CHECK(s == (edges[e] < P ? edges[e] : P));
        }
        // error paths: a wrong P is refused, nothing is written out of bounds on the way
        if (it % 7 == 0) {
            long long nm2;
            CHECK(ra_shard_plan(mask.data(), H, W, world, ground, P + 1, ground ? gpos.data() : nullptr, nullptr, 0, nullptr, std::vector<long long>(n + 1).data(),
                                std::vector<long long>(n + 1).data(), ground ? std::vector<long long>(n + 1).data() : nullptr, counts.data(), nullptr, &nm2) == 1);
            if (P > 0)
                CHECK(ra_shard_plan(mask.data(), H, W, world, ground, P - 1, ground ? gpos.data() : nullptr, nullptr, 0, nullptr, std::vector<long long>(n).data(),
                                    std::vector<long long>(n).data(), ground ? std::vector<long long>(n).data() : nullptr, counts.data(), nullptr, &nm2) == 1);
            CHECK(ra_shard_plan(nullptr, H, W, world, ground, P, nullptr, nullptr, 0, nullptr, order.data(), src.data(), nullptr, counts.data(), nullptr, &nm2) == 1);
            CHECK(ra_shard_plan(mask.data(), H, W, 0, ground, P, nullptr, nullptr, 0, nullptr, order.data(), src.data(), nullptr, counts.data(), nullptr, &nm2) == 1);
        }
    }
    std::printf("shard_fuzz: %d cases ok\n", n_cases);
    return 0;
}
