// Host-side weight packer: reference state_dict (fp32) -> f16/bf16 weight streams in MFMA fragment order.
//   weight_norm fold  W = g * v / |v|_row          lib/utils/net_utils.py:1326-1327, base_network.py:145-149
//   SDF skip          cat([x, inputs]) / sqrt(2)   lib/utils/net_utils.py:1345-1346 (folded into lin4)
//   resd skip         cat([x, input])              lib/utils/net_utils.py:1266-1267 (x first)
//   cond slices are kept in fp32 for the per-frame bias fold (ra_set_frame)
#include "ra_ctx.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {

uint16_t f2bf(float f) {   // round to nearest even
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    const uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}

struct Mat {
    int rows = 0, cols = 0;
    std::vector<float> v;
    Mat() {}
    Mat(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.f) {}
    float& at(int r, int c) { return v[(size_t)r * cols + c]; }
    float at(int r, int c) const { return v[(size_t)r * cols + c]; }
};

uint16_t f2h(float f) {    // IEEE half, round to nearest even (clang's _Float16 conversion)
    const _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

struct Packer {                 // layer table + fp32 bias arena (the 16-bit weights go to the streams below)
    std::vector<float> b;
    bool half = false;          // f16 (true) or bf16 (false)

    // M: [rows<=256 or <=32][K]; records the layer's K depth and appends its bias row (padded to n_rows_pad)
    WideLayer add(const Mat& M, const std::vector<float>& bias, int n_rows_pad, int k_min = 0) {
        int K = (M.cols + 15) / 16 * 16;
        if (K < k_min) K = k_min;
        WideLayer L;
        L.ks = (uint32_t)(K / 16);
        L.bias = (uint32_t)b.size();
        for (int r = 0; r < n_rows_pad; ++r) b.push_back(r < (int)bias.size() ? bias[r] : 0.f);
        while (b.size() % 4) b.push_back(0.f);
        return L;
    }
};

// ---- K3 weight stream (ra_k3.hpp): fragments in the exact order the kernel consumes them,
// [layer][row block rb][k-step][lane][8], K indices permuted so that the packed D fragment of one layer IS the
// B fragment of the next (no LDS round trip for activations):
//   hidden k-step ks, lane half h, slot j  <->  feature 32*(ks>>1) + 16*(ks&1) + 8*(j>>2) + 4*h + (j&3)
//   PE k-step (first layers / skip part), q = 8*ks + j  <->  channel chan(q, h) (see pe_chan_*)
struct StreamBuilder {
    std::vector<uint16_t> w;
    bool half = false;
    uint16_t cv(float v) const { return half ? f2h(v) : f2bf(v); }
    static int hidden_feature(int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
    // rows_pad/32 row blocks; per row block: 16 hidden k-steps of Mh (may be null) then 4 PE k-steps of Mp (may be null)
    bool pairs = false;         // interleave the fragments of row blocks (2p, 2p + 1) k-step by k-step (K3's latency variants, ra_stream.hpp row_blocks)
    template <typename ChanFn>
    void frag(const Mat* Mh, const Mat* Mp, ChanFn chan, int rb, int ks, float pe_scale) {       // ks < 16: hidden k-step, else encoding k-step ks - 16
        auto put = [&](float v) { w.push_back(cv(v)); };
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int row = rb * 32 + (lane & 31);
                if (ks < 16) {
                    const int col = hidden_feature(ks, lane >> 5, j);
                    put(row < Mh->rows && col < Mh->cols ? Mh->at(row, col) : 0.f);
                } else {
                    const int col = chan(8 * (ks - 16) + j, lane >> 5);
                    put(row < Mp->rows && col >= 0 && col < Mp->cols ? Mp->at(row, col) * pe_scale : 0.f);
                }
            }
    }
    // ---- fragments for v_mfma_f32_16x16x32 (K3C, ra_k3c.hpp): a row block is 16 output rows, a k-step 32 inputs.
    // lane = (row m = lane & 15, k group kg = lane >> 4), slot j < 8.  Hidden k-step ks: the D fragments of row blocks 2 ks and 2 ks + 1 of
    // the previous layer (lane group g holds rows 16 rb + 4 g + i) packed side by side ARE this B fragment, so
    //     slot (kg, j)  <->  feature 32 ks + 16 (j >> 2) + 4 kg + (j & 3);
    // encoding k-step p (0 | 1): slot (kg, j) is slot q = 8 (2 p + (kg >> 1)) + j of lane half h = kg & 1 of the 32x32 layout (pe_chan_*).
    static int hidden_feature16(int ks, int kg, int j) { return 32 * ks + 16 * (j >> 2) + 4 * kg + (j & 3); }
    template <typename ChanFn>
    // part 1 = hi: the value rounded to IEEE half; part 2 = lo: the rounded residual of that rounding (compensated products)
    void frag16(const Mat* Mh, const Mat* Mp, ChanFn chan, int rb, int ks, float pe_scale, int part) {      // ks < 8: hidden, else encoding k-step ks - 8
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int row = rb * 16 + (lane & 15), kg = lane >> 4;
                float v = 0.f;
                if (ks < 8) {
                    const int col = hidden_feature16(ks, kg, j);
                    if (row < Mh->rows && col < Mh->cols) v = Mh->at(row, col);
                } else {
                    const int col = chan(8 * (2 * (ks - 8) + (kg >> 1)) + j, kg & 1);
                    if (row < Mp->rows && col >= 0 && col < Mp->cols) v = Mp->at(row, col) * pe_scale;
                }
                const uint16_t hi = cv(v);
                if (part != 2) { w.push_back(hi); continue; }
                _Float16 hf;
                std::memcpy(&hf, &hi, 2);
                w.push_back(cv(v - (float)hf));
            }
    }
    template <typename ChanFn>
    void add16(const Mat* Mh, const Mat* Mp, ChanFn chan, int rows_pad, float pe_scale) {       // always [hi | lo] pairs
        for (int rb = 0; rb < rows_pad / 16; ++rb)
            for (int ks = Mh ? 0 : 8; ks < (Mp ? 10 : 8); ++ks) {
                frag16(Mh, Mp, chan, rb, ks, pe_scale, 1);
                frag16(Mh, Mp, chan, rb, ks, pe_scale, 2);
            }
    }
    template <typename ChanFn>
    // hks: hidden k-steps emitted (16 = all 256 inputs; the trimmed stream drops the k-steps of inputs that do not exist)
    void add(const Mat* Mh, const Mat* Mp, ChanFn chan, int rows_pad, float pe_scale, int hks = 16) {
        const int nb = rows_pad / 32, group = (pairs && nb % 2 == 0) ? 2 : 1;
        for (int rb0 = 0; rb0 < nb; rb0 += group)
            for (int ks = Mh ? 0 : 16; ks < (Mp ? 20 : 16); ++ks) {
                if (ks >= hks && ks < 16) continue;
                for (int g = 0; g < group; ++g) frag(Mh, Mp, chan, rb0 + g, ks, pe_scale);
            }
    }
};

// PE10 (63 channels: x | per frequency sin xyz, cos xyz): slot q < 30 -> frequency q/3, axis q%3, sin (h=0) / cos (h=1)
int pe_chan_resd(int q, int h) {
    if (q < 30) return 3 + 6 * (q / 3) + 3 * h + q % 3;
    if (q == 30) return h ? 1 : 0;
    return h ? -1 : 2;
}
// PE8 + residual ("lo") columns of with_lo(): 0..50 PE, 51..53 lo(x), 54..56 lo(sin f0), 57..59 lo(cos f0)
int pe_chan_sdf(int q, int h) {
    if (q < 24) return 3 + 6 * (q / 3) + 3 * h + q % 3;
    if (q < 27) return h ? 51 + (q - 24) : (q - 24);
    if (q < 30) return h ? 57 + (q - 27) : 54 + (q - 27);
    return -1;
}

// the 32 + 32 input slots of the colour net's first layer that are not features: PE4 of the big-pose view direction
// (27 channels: x | per frequency sin xyz, cos xyz) and the world normal (3): slot q < 12 -> frequency q/3, axis q%3, sin (h=0) / cos (h=1);
// q = 12..14 -> view direction (h=0) / normal (h=1)
int pe_chan_col(int q, int h) {
    if (q < 12) return 3 + 6 * (q / 3) + 3 * h + q % 3;
    if (q < 15) return h ? 27 + (q - 12) : (q - 12);
    return -1;
}
// the 3 seeds of the residual head's transposed layer live in slots 0..2 of lane half 0
int chan_head3(int q, int h) { return (h == 0 && q < 3) ? q : -1; }

Mat transpose(const Mat& M, int rows_pad = 0) {
    Mat o(M.cols > rows_pad ? M.cols : rows_pad, M.rows);
    for (int r = 0; r < M.rows; ++r)
        for (int c = 0; c < M.cols; ++c) o.at(c, r) = M.at(r, c);
    return o;
}

// rows of a transposed encoding block in D-fragment order: row 32*blk + 8q + 4h + i <-> encoding slot s = 16*blk + 4q + i of
// lane half h <-> channel chan(s, h); channels >= n_real (the duplicated "lo" columns) and empty slots give zero rows.
// W: forward matrix [out][channels]; result [64][out]
template <typename ChanFn>
Mat transpose_pe(const Mat& W, ChanFn chan, int n_real) {
    Mat o(64, W.rows);
    for (int r = 0; r < 64; ++r) {
        const int blk = r / 32, rr = r % 32, q = rr / 8, h = (rr % 8) / 4, i = rr % 4;
        const int c = chan(16 * blk + 4 * q + i, h);
        if (c < 0 || c >= n_real || c >= W.cols) continue;
        for (int j = 0; j < W.rows; ++j) o.at(r, j) = W.at(j, c);
    }
    return o;
}

Mat slice_cols(const Mat& M, int c0, int c1) {
    Mat o(M.rows, c1 - c0);
    for (int r = 0; r < M.rows; ++r)
        for (int c = c0; c < c1; ++c) o.at(r, c - c0) = M.at(r, c);
    return o;
}

Mat slice_rows(const Mat& M, int r0, int r1) {
    Mat o(r1 - r0, M.cols);
    for (int r = r0; r < r1; ++r)
        for (int c = 0; c < M.cols; ++c) o.at(r - r0, c) = M.at(r, c);
    return o;
}

}  // namespace

static bool get(const std::map<std::string, std::vector<float>>& sd, const std::string& k, std::vector<float>& out) {
    auto it = sd.find(k);
    if (it == sd.end()) return false;
    out = it->second;
    return true;
}

static bool get_mat(const std::map<std::string, std::vector<float>>& sd, const std::string& k, int rows, int cols, Mat& M, std::string& err) {
    std::vector<float> v;
    if (!get(sd, k, v)) { err = "missing weight " + k; return false; }
    if ((int)v.size() != rows * cols) { err = "bad shape for " + k + ": numel " + std::to_string(v.size()) + " expected " + std::to_string(rows) + "x" + std::to_string(cols); return false; }
    M = Mat(rows, cols);
    M.v = v;
    return true;
}

static bool get_wn(const std::map<std::string, std::vector<float>>& sd, const std::string& p, int rows, int cols, Mat& M, std::vector<float>& bias, std::string& err) {
    Mat v;
    std::vector<float> g;
    if (!get_mat(sd, p + ".weight_v", rows, cols, v, err)) return false;
    if (!get(sd, p + ".weight_g", g) || (int)g.size() != rows) { err = "missing/bad " + p + ".weight_g"; return false; }
    if (!get(sd, p + ".bias", bias) || (int)bias.size() != rows) { err = "missing/bad " + p + ".bias"; return false; }
    M = Mat(rows, cols);
    for (int r = 0; r < rows; ++r) {
        double n2 = 0;
        for (int c = 0; c < cols; ++c) n2 += (double)v.at(r, c) * v.at(r, c);
        const float s = g[r] / (float)std::sqrt(n2);
        for (int c = 0; c < cols; ++c) M.at(r, c) = v.at(r, c) * s;
    }
    return true;
}

int ra_pack_weights(ra_ctx* ctx, std::string& err) {
    const auto& sd = ctx->state_dict;
    const ra_config& c = ctx->cfg;
    const int cond = c.n_bones * 3;
    const int xyz_dim = 3 + 6 * c.xyz_res, sdf_dim = 3 + 6 * c.sdf_res, view_dim = 3 + 6 * c.view_res;
    if (c.xyz_res != 10 || c.sdf_res != 8 || c.view_res != 4) { err = "kernels are specialised for xyz_res=10, sdf_res=8, view_res=4 (configs/base.yaml:47-49)"; return 1; }
    Packer P;
    P.half = c.mlp_f16 != 0;
    HostNets& H = ctx->host;
    Mat Rm[8], Rpe4, Rhead, Sm[8], Spe4, Shead, Sfeat;     // kept for the weight streams
    Mat C0a, C0b, C1, C2, C3, Chead, M0, M1, Mhead;
    // ---- residual deformation
    const std::string rp = "residual_deformation_network.mlp.linears.";
    const int in_ch = xyz_dim + cond;
    for (int i = 0; i < 9; ++i) {
        const int I = i == 0 ? in_ch : (i == 4 ? 256 + in_ch : 256);
        const int O = i == 8 ? 3 : 256;
        Mat W;
        std::vector<float> b;
        if (!get_mat(sd, rp + std::to_string(i) + ".weight", O, I, W, err)) return 1;
        if (!get(sd, rp + std::to_string(i) + ".bias", b) || (int)b.size() != O) { err = "missing/bad " + rp + std::to_string(i) + ".bias"; return 1; }
        if (i == 0) {
            H.geo.r[0] = P.add(slice_cols(W, 0, xyz_dim), b, 256);
            Rm[0] = slice_cols(W, 0, xyz_dim);
            H.cond_r0 = slice_cols(W, xyz_dim, in_ch).v;
            H.b_r0 = b;
        } else if (i == 4) {
            H.geo.r[4] = P.add(slice_cols(W, 0, 256), b, 256);
            H.geo.r4b = P.add(slice_cols(W, 256, 256 + xyz_dim), std::vector<float>(), 256);
            Rm[4] = slice_cols(W, 0, 256);
            Rpe4 = slice_cols(W, 256, 256 + xyz_dim);
            H.cond_r4 = slice_cols(W, 256 + xyz_dim, 256 + in_ch).v;
            H.b_r4 = b;
        } else if (i == 8) {
            H.geo.rhead = P.add(W, b, 32);
            Rhead = W;
        } else {
            H.geo.r[i] = P.add(W, b, 256);
            Rm[i] = W;
        }
    }
    // ---- signed distance
    const std::string sp = "signed_distance_network.mlp.lin";
    for (int l = 0; l < 9; ++l) {
        const int I = l == 0 ? sdf_dim : 256;
        const int O = l == 3 ? 256 - sdf_dim : (l == 8 ? 257 : 256);
        Mat W;
        std::vector<float> b;
        if (!get_wn(sd, sp + std::to_string(l), O, I, W, b, err)) return 1;
        if (l == 4) for (auto& x : W.v) x *= (float)(1.0 / std::sqrt(2.0));
        // first-layer style inputs are [x 3 | PE8 48 | lo(x) 3 | lo(freq-0 sin,cos) 6 | pad]: duplicate the weights
        auto with_lo = [&](const Mat& Wp) {
            Mat o(Wp.rows, sdf_dim + 9);
            for (int r = 0; r < Wp.rows; ++r) {
                for (int k = 0; k < sdf_dim; ++k) o.at(r, k) = Wp.at(r, k);
                for (int k = 0; k < 9; ++k) o.at(r, sdf_dim + k) = Wp.at(r, k);
            }
            return o;
        };
        if (l == 0) {
            H.geo.s[0] = P.add(with_lo(W), b, 256);
            Sm[0] = with_lo(W);
        } else if (l == 4) {
            const int nx = 256 - sdf_dim;   // 205
            H.geo.s[4] = P.add(slice_cols(W, 0, nx), b, 256, 256);         // K zero-padded to 256 (cols 205..255 of the tile hold junk)
            H.geo.s4b = P.add(with_lo(slice_cols(W, nx, 256)), std::vector<float>(), 256);
            Sm[4] = slice_cols(W, 0, nx);
            Spe4 = with_lo(slice_cols(W, nx, 256));
        } else
        if (l == 8) {
            H.geo.shead = P.add(slice_rows(W, 0, 1), std::vector<float>(b.begin(), b.begin() + 1), 32);
            H.geo.sfeat = P.add(slice_rows(W, 1, 257), std::vector<float>(b.begin() + 1, b.end()), 256);
            Shead = slice_rows(W, 0, 1);
            Sfeat = slice_rows(W, 1, 257);
        } else {
            H.geo.s[l] = P.add(W, b, 256);
            Sm[l] = W;
        }
    }
    {
        std::vector<float> beta;
        if (!get(sd, "signed_distance_network._beta", beta) || beta.size() != 1) { err = "missing signed_distance_network._beta"; return 1; }
        H.beta = std::fmin(std::fmax(beta[0], 1e-9f), 1e6f);     // base_network.py:74-76
    }
    // ---- colour net (always present in both checkpoints; frozen for relight)
    H.has_color = sd.count("render_network.l0.weight_v") > 0;
    if (H.has_color) {
        const int in0 = view_dim + 3 + 256;
        Mat W;
        std::vector<float> b;
        if (!get_wn(sd, "render_network.l0", 256, in0, W, b, err)) return 1;
        H.col.c0a = P.add(slice_cols(W, view_dim + 3, in0), b, 256);
        H.col.c0b = P.add(slice_cols(W, 0, view_dim + 3), std::vector<float>(), 256);
        C0a = slice_cols(W, view_dim + 3, in0); C0b = slice_cols(W, 0, view_dim + 3);
        if (!get_wn(sd, "render_network.l1", 256, 256, W, b, err)) return 1;
        H.col.c1 = P.add(W, b, 256);
        C1 = W;
        if (!get_wn(sd, "render_network.l2", 256, 256, W, b, err)) return 1;
        H.col.c2 = P.add(W, b, 256);
        C2 = W;
        if (!get_wn(sd, "render_network.l3", 256, 256 + cond, W, b, err)) return 1;
        H.col.c3 = P.add(slice_cols(W, 0, 256), b, 256);
        C3 = slice_cols(W, 0, 256);
        H.cond_c3 = slice_cols(W, 256, 256 + cond).v;
        H.b_c3 = b;
        if (!get_wn(sd, "render_network.l4", 3, 256, W, b, err)) return 1;
        H.col.chead = P.add(W, b, 32);
        Chead = W;
    }
    // ---- material heads (relight)
    if (c.relight) {
        Mat a0, a1, a2, r0, r1, r2;
        std::vector<float> ba0, ba1, ba2, br0, br1, br2;
        const std::string an = "albedo_network.linears.", rn = "roughness_network.linears.";
        if (!get_mat(sd, an + "0.weight", 128, 256, a0, err) || !get_mat(sd, an + "1.weight", 128, 128, a1, err) || !get_mat(sd, an + "2.weight", 3, 128, a2, err)) return 1;
        if (!get_mat(sd, rn + "0.weight", 128, 256, r0, err) || !get_mat(sd, rn + "1.weight", 128, 128, r1, err) || !get_mat(sd, rn + "2.weight", 1, 128, r2, err)) return 1;
        if (!get(sd, an + "0.bias", ba0) || !get(sd, an + "1.bias", ba1) || !get(sd, an + "2.bias", ba2) || !get(sd, rn + "0.bias", br0) || !get(sd, rn + "1.bias", br1) || !get(sd, rn + "2.bias", br2)) { err = "missing material bias"; return 1; }
        Mat m0(256, 256), m1(256, 256), mh(4, 256);
        std::vector<float> b0(256), b1(256), bh(4);
        for (int r = 0; r < 128; ++r) {
            for (int k = 0; k < 256; ++k) { m0.at(r, k) = a0.at(r, k); m0.at(128 + r, k) = r0.at(r, k); }
            for (int k = 0; k < 128; ++k) { m1.at(r, k) = a1.at(r, k); m1.at(128 + r, 128 + k) = r1.at(r, k); }
            b0[r] = ba0[r]; b0[128 + r] = br0[r];
            b1[r] = ba1[r]; b1[128 + r] = br1[r];
        }
        for (int k = 0; k < 128; ++k) {
            for (int r = 0; r < 3; ++r) mh.at(r, k) = a2.at(r, k);
            mh.at(3, 128 + k) = r2.at(0, k);
        }
        for (int r = 0; r < 3; ++r) bh[r] = ba2[r];
        bh[3] = br2[0];
        H.mat.m0 = P.add(m0, b0, 256);
        H.mat.m1 = P.add(m1, b1, 256);
        H.mat.mhead = P.add(mh, bh, 32);
        M0 = m0; M1 = m1; Mhead = mh;
        if (!get(sd, "light_xyz_", H.light_xyz) || !get(sd, "light_area", H.light_area) || !get(sd, "light_sharp", H.light_sharp)) { err = "missing light_xyz_/light_area/light_sharp"; return 1; }
        if (H.light_xyz.size() != H.light_area.size() * 3 || H.light_area.size() != H.light_sharp.size() || H.light_area.size() > RA_N_LIGHTS_MAX) { err = "bad light buffer shapes"; return 1; }
    }
    {   // K3 stream: resd L0..L7, head, sdf L0..L7, head.  The softplus layers work in the scaled domain
        // y' = y * beta*log2(e) (beta = 100): only the layers fed by the unscaled encoding carry the factor.
        StreamBuilder S;
        S.half = P.half;
        const float sp = 144.26950408889634f;
        for (int i = 0; i < 8; ++i) S.add(i == 0 ? nullptr : &Rm[i], i == 0 ? &Rm[0] : (i == 4 ? &Rpe4 : nullptr), pe_chan_resd, 256, 1.f);
        S.add(&Rhead, nullptr, pe_chan_resd, 32, 1.f);
        for (int l = 0; l < 8; ++l) S.add(l == 0 ? nullptr : &Sm[l], l == 0 ? &Sm[0] : (l == 4 ? &Spe4 : nullptr), pe_chan_sdf, 256, sp);
        S.add(&Shead, nullptr, pe_chan_sdf, 32, 1.f);
        if (S.w.size() != (size_t)1952 * 512) { err = "internal: weight stream has " + std::to_string(S.w.size() / 512) + " fragments, expected 1952"; return 1; }
        H.sarena = S.w;
        // the 8-wave K3's stream without the MFMAs that only multiply padding: lin3 of the SDF net has 256 - sdf_dim outputs (205: 7 row
        // blocks), lin4 reads them in 14 k-steps (ra_k3.hpp run_net).  The kernel is compiled for that shape.
        if (Sm[3].rows > 224 || Sm[4].cols > 224) { err = "SDF skip layer has " + std::to_string(Sm[3].rows) + " outputs, the distance kernel is built for <= 224 (multires differs from the compiled variant)"; return 1; }
        StreamBuilder St;
        St.half = P.half;
        for (int i = 0; i < 8; ++i) St.add(i == 0 ? nullptr : &Rm[i], i == 0 ? &Rm[0] : (i == 4 ? &Rpe4 : nullptr), pe_chan_resd, 256, 1.f);
        St.add(&Rhead, nullptr, pe_chan_resd, 32, 1.f);
        for (int l = 0; l < 8; ++l)
            St.add(l == 0 ? nullptr : &Sm[l], l == 0 ? &Sm[0] : (l == 4 ? &Spe4 : nullptr), pe_chan_sdf, l == 3 ? 224 : 256, sp, l == 4 ? 14 : 16);
        St.add(&Shead, nullptr, pe_chan_sdf, 32, 1.f);
        if (St.w.size() != (size_t)1920 * 512) { err = "internal: trimmed weight stream has " + std::to_string(St.w.size() / 512) + " fragments, expected 1920"; return 1; }
        H.sarena_trim = St.w;
        StreamBuilder Sp;           // the same stream in pair order
        Sp.half = P.half;
        Sp.pairs = true;
        for (int i = 0; i < 8; ++i) Sp.add(i == 0 ? nullptr : &Rm[i], i == 0 ? &Rm[0] : (i == 4 ? &Rpe4 : nullptr), pe_chan_resd, 256, 1.f);
        Sp.add(&Rhead, nullptr, pe_chan_resd, 32, 1.f);
        for (int l = 0; l < 8; ++l) Sp.add(l == 0 ? nullptr : &Sm[l], l == 0 ? &Sm[0] : (l == 4 ? &Spe4 : nullptr), pe_chan_sdf, 256, sp);
        Sp.add(&Shead, nullptr, pe_chan_sdf, 32, 1.f);
        H.sarena_pairs = Sp.w;
        // K3C: every fragment as an IEEE-half [hi | lo] pair, whatever the operand type of the plain kernels, in the 16x16x32 layout
        // (16 row blocks of 16 rows per layer, 8 hidden + 2 encoding k-steps of 32).  The SDF net's own hi + lo input columns (with_lo)
        // are fed zeros by K3C: its whole encoding is split.
        StreamBuilder Sc;
        Sc.half = true;
        for (int i = 0; i < 8; ++i) Sc.add16(i == 0 ? nullptr : &Rm[i], i == 0 ? &Rm[0] : (i == 4 ? &Rpe4 : nullptr), pe_chan_resd, 256, 1.f);
        Sc.add16(&Rhead, nullptr, pe_chan_resd, 16, 1.f);
        for (int l = 0; l < 8; ++l) Sc.add16(l == 0 ? nullptr : &Sm[l], l == 0 ? &Sm[0] : (l == 4 ? &Spe4 : nullptr), pe_chan_sdf, 256, sp);
        Sc.add16(&Shead, nullptr, pe_chan_sdf, 16, 1.f);
        if (Sc.w.size() != (size_t)3872 * 512) { err = "internal: split weight stream has " + std::to_string(Sc.w.size() / 512) + " fragments, expected 3872"; return 1; }
        // K3CC (ra_k3cc.hpp): four waves share a 16-point tile, wave w owns the row blocks 4 i + w of every layer and walks a PRIVATE
        // stream: its 4 row blocks per layer in order, both heads in every wave's stream (each wave computes them itself).  Same
        // fragments, permuted: 4 x 992 after the 3872 of the per-wave kernel.
        {
            static const int lay_off[8] = {0, 64, 320, 576, 832, 1152, 1408, 1664}, lay_f[8] = {4, 16, 16, 16, 20, 16, 16, 16};
            std::vector<uint16_t>& A = Sc.w;
            for (int w = 0; w < 4; ++w)
                for (int netb = 0; netb < 3872; netb += 1936) {
                    for (int l = 0; l < 8; ++l)
                        for (int i = 0; i < 4; ++i) {
                            const size_t src = (size_t)(netb + lay_off[l] + (4 * i + w) * lay_f[l]) * 512;
                            const std::vector<uint16_t> rbk(A.begin() + src, A.begin() + src + (size_t)lay_f[l] * 512);
                            A.insert(A.end(), rbk.begin(), rbk.end());
                        }
                    const size_t hsrc = (size_t)(netb + 1920) * 512;
                    const std::vector<uint16_t> head(A.begin() + hsrc, A.begin() + hsrc + (size_t)16 * 512);
                    A.insert(A.end(), head.begin(), head.end());
                }
            if (A.size() != (size_t)(3872 + 4 * 992) * 512) { err = "internal: cooperative weight stream has the wrong size"; return 1; }
        }
        H.sarena_c = Sc.w;
    }
    {   // K4 (reverse mode) streams, same fragment order / K permutation as the K3 stream.
        // forward: the K3 stream + the 256 feature rows of lin8 (head output, no activation; unscaled weights on scaled inputs)
        StreamBuilder F;
        F.half = P.half;
        F.w = H.sarena;
        F.add(&Sfeat, nullptr, pe_chan_sdf, 256, 1.f);
        if (F.w.size() != (size_t)2080 * 512) { err = "internal: forward stream of the full query has " + std::to_string(F.w.size() / 512) + " fragments"; return 1; }
        H.fwd_arena = F.w;
        // backward (unscaled weights; gradients are carried times GRAD_SCALE in the kernel): u = W_l^T delta_l, rows = input
        // features of layer l; the encoding-fed layers end in 64 rows of encoding-channel gradients (D-fragment order)
        StreamBuilder B;
        B.half = P.half;
        const float inv_sp = 1.f;
        for (int l = 7; l >= 1; --l) {
            Mat T = transpose(Sm[l], 256);                       // l == 4: 205 real rows, the rest zero
            B.add(&T, nullptr, pe_chan_sdf, 256, inv_sp);
            if (l == 4) { Mat Tp = transpose_pe(Spe4, pe_chan_sdf, sdf_dim); B.add(&Tp, nullptr, pe_chan_sdf, 64, inv_sp); }
        }
        { Mat Tp = transpose_pe(Sm[0], pe_chan_sdf, sdf_dim);
          B.add(&Tp, nullptr, pe_chan_sdf, 64, inv_sp); }
        {   // residual head: 3 seeds -> 256 rows, as a 4-k-step "encoding" layer
            Mat T = transpose(Rhead, 256);                       // [256][3]
            B.add(nullptr, &T, chan_head3, 256, 1.f);
        }
        for (int l = 7; l >= 1; --l) {
            Mat T = transpose(Rm[l], 256);
            B.add(&T, nullptr, pe_chan_resd, 256, 1.f);
            if (l == 4) { Mat Tp = transpose_pe(Rpe4, pe_chan_resd, xyz_dim); B.add(&Tp, nullptr, pe_chan_resd, 64, 1.f); }
        }
        { Mat Tp = transpose_pe(Rm[0], pe_chan_resd, xyz_dim); B.add(&Tp, nullptr, pe_chan_resd, 64, 1.f); }
        H.bwd_geo_frags = (int)(B.w.size() / 512);
        // heads on the features: material (softplus, scaled domain: the layer fed by the unscaled features carries the factor)
        // or colour net (ReLU; first layer = 16 feature k-steps + 4 k-steps of [PE4(view) | normal])
        const float sp = 144.26950408889634f;
        if (c.relight) {
            Mat M0s = M0;
            for (auto& x : M0s.v) x *= sp;
            B.add(&M0s, nullptr, pe_chan_col, 256, 1.f);
            B.add(&M1, nullptr, pe_chan_col, 256, 1.f);
            B.add(&Mhead, nullptr, pe_chan_col, 32, 1.f);
        } else if (H.has_color) {
            // interleave per row block: 16 hidden k-steps (features) then 4 encoding k-steps (view / normal)
            B.add(&C0a, &C0b, pe_chan_col, 256, 1.f);
            B.add(&C1, nullptr, pe_chan_col, 256, 1.f);
            B.add(&C2, nullptr, pe_chan_col, 256, 1.f);
            B.add(&C3, nullptr, pe_chan_col, 256, 1.f);
            B.add(&Chead, nullptr, pe_chan_col, 32, 1.f);
        }
        H.bwd_arena = B.w;
        H.bwd_frags = (int)(B.w.size() / 512);
        // the backward kernels walk a stream of compile-time length: a state_dict / cfg that changes it must fail here, not read a misaligned stream
        if ((c.relight || H.has_color) && H.bwd_frags != 16 * mlp_full_rev_bwd_stages(c.relight)) {
            err = "backward stream of the full query has " + std::to_string(H.bwd_frags) + " fragments, the kernels are built for " +
                  std::to_string(16 * mlp_full_rev_bwd_stages(c.relight)) + " (cond / view / xyz dimensions differ from the compiled variant)";
            return 1;
        }
        H.shead_row.assign(256, 0.f);
        for (int k = 0; k < 256; ++k) H.shead_row[k] = Shead.at(0, k);
    }
    H.barena = P.b;
    return 0;
}
