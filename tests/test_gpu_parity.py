"""GPU parity: the HIP path (through the C ABI) vs the golden fixtures generated from the reference
(tests/golden/*.npz) and vs the oracle on the same seeded inputs.  Run with `-m gpu` on an MI355X.

Tolerances (f16 MFMA operands, fp32 accumulate; everything outside the MLPs is fp32):
  coarse level / LBS warp ............ 1e-6 absolute (fp32, different summation order only), exact 3-NN indices
  resd, sdf, feat (MLP outputs) ...... 3e-4 abs (sdf), 2e-4 (feat); measured mean 4.5e-5 / 1.6e-5
  HDQ sdf ............................ 3e-4 abs
  normals (reverse-mode, K4 backward) .. 8e-3 abs per component, mean < 6e-4
  albedo / roughness / occ ........... 1e-4 abs
  traced surfaces .................... median |st err| < 2e-4; rays whose 16-iteration trace has not converged
                                       amplify sdf noise (occ = 500 d / t), so frame maps are judged by the
                                       fraction of pixels within tolerance + PSNR, not by max error
  frames ............................. SURVEY.md:409 contract for the 16-bit path: rgb PSNR >= 50 dB and max |err| <= 1e-2.
      What assert_contract() asserts, exactly: PSNR >= 50 dB AND max |err| <= 1e-2 over the rays whose reference value fp32 itself pins,
      and PSNR >= 40 dB over ALL rays (a sanity bound; where the all-ray figure reaches 50 dB it is asserted too: `all_rays=True`).
      The rays fp32 does not pin are coin tosses of the reference's own closest-approach rule (tests/golden/fp32_unstable_rays.json,
      tools/fp32_stability.py: listed when 3e-7 noise on the traced distances moves the surface point by > 0.1 mm in any of 32 runs;
      the file also holds every listed ray's flip probability at fp32's own noise level, 1.2e-7 — 30-50 % for the rays listed on
      frame_relight / frame_relight_smooth / frame_novel / frame_ground / smoke).  One such ray at 0.05 rgb takes a 256-ray frame
      from 64 to 51 dB, which is why the all-ray PSNR is only a sanity bound there.
      * through assert_contract: frame_relight, frame_relight_smooth, frame_novel (three probes), frame_ground, the multi-chunk and
        other-pose cases, the full-size sample, the volume frames (> 80 dB);
      * the switch matrix (round 5; tests/golden/switches.npz: the reference under 42 configuration overrides + round 6's 8 hard cases, one process per variant):
        25 relit windows, 8 ground-pass frames, 3 volume frames, 2 rotating-light sequences; rgb through assert_contract over ALL rays
        (no fp32-unstable ray on these windows: 60-96 dB, max 1.5e-4 .. 7.8e-3), the other maps to their tolerances;
        test_box_structure_is_morton_sorted pins the per-frame vertex order of the box structure against numpy;
      * round 6: frame_novel_ground (80.5-84.7 dB, max 1.9e-3 over all pixels; up to round 5: one pixel at 1.1e-2) and the full 512 x 512 frame
        against its all-compensated twin (test_full_frame_shadow_tier_is_harmless: max 9.3e-3, no pixel over 1e-2; round 5: 1.3e-2, 2 pixels)
        now hold the contract outright: the shadow rays towards the frame's key lights are traced in compensated arithmetic
        (cfg.key_light_share, csrc/ra_trace.hip key_lights_kernel).  The reference-made hard cases (test_hard_case_switch_matrix: a body
        part shadowing the body at distance under a key light, trained-like weights) are what showed the need: profiles/r06_hard_cases.txt.
      History: on the SURVEY 8d body (white noise in the skinning logits -> the world -> big-pose warp jumps by ~1 cm between neighbouring
      query points, the REFERENCE's trace ends in a limit cycle on ~9 % of the hit rays) plain 16-bit operands reach 50.6 dB / max 4.7e-2
      (tests/golden/precision_floor.json, tools/precision_floor.py); fp32 itself IS stable on all but ~0.5 % of those rays
      (tools/precision_tiers.py: a float64-accumulated oracle agrees with the fp32 one to 89-110 dB), and compensating ONLY the surface
      trace's distance queries (K3C, csrc/ra_k3c.hpp: f16 hi + lo operand pairs, 2 % of a frame's fine queries) reaches the contract.
      Measured: frame_relight 68.5 dB / 2.9e-3 on the fp32-stable rays (plain f16: 50.7 / 4.7e-2), frame_novel 66.7-73.5 dB (46-52).
  stage bisect ....................... test_mlp_stage_matches_the_operand_rounding_emulation: HIP sdf vs the kernel-like
        emulation is several times closer than the emulation is to fp32, i.e. the in-kernel loss IS the operand rounding
        (v_sin/v_cos encodings, scaled-domain softplus through v_exp/v_log and the f16 re-pack add nothing measurable).
bf16 operands are available (cfg.mlp_dtype='bf16'); they are ~10x noisier (sdf mean err 5e-4) and tested loosely.
"""
import json
import os

import numpy as np
import pytest
import torch

from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def build(mode, dtype='f16', **kw):
    from relightableavatar_amd.networks import make_network
    dev = _dev()
    cfg = make_cfg(mode, mlp_dtype=dtype, **kw)
    relight = mode in ('relight', 'novel_light')
    net = make_network(cfg)
    net.load_state_dict(synthetic.make_state_dict(0, relight=relight, cfg=cfg))
    return cfg, net.to(dev).eval(), dev


def err(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    same = (a == b) | (a.isnan() & b.isnan())
    return torch.where(same, torch.zeros_like(a), (a - b).abs())


def psnr(a, b):
    e = err(a, b)
    return float(-10 * torch.log10(torch.mean(e ** 2)))


@pytest.fixture(scope='module')
def ops(golden):
    return {k: T(v) for k, v in golden('ops.npz').items()}


@pytest.fixture(scope='module')
def relight():
    cfg, net, dev = build('relight')
    body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
    eng = net.set_frame(body)
    return cfg, net, dev, body, eng


def test_native_library_is_loaded(relight):
    """the path under test is the in-tree HIP extension, not a torch fallback"""
    maps = open('/proc/self/maps').read()
    assert 'librelightableavatar_hip.so' in maps
    # ... and the cooperative small-launch distance kernel (K3CC) passed its on-device self-test: a miscompiled K3CC would silently degrade
    # every small launch to K3C's 4-wave tiles (same results, slower)
    assert relight[4].k3cc_enabled()


def test_mlp_stage(ops, relight):
    _, _, dev, _, eng = relight
    resd, sdf, feat = eng.debug_mlp(ops['mlp_bpts'].to(dev))          # stage outputs of the full query's forward kernel (production K4)
    assert float(err(resd, ops['mlp_resd']).max()) < 5e-6
    e = err(sdf[:, None], ops['mlp_sdf'])
    assert float(e.max()) < 6e-4 and float(e.mean()) < 8e-5          # the f16 operand-rounding emulation itself: max 3.4e-4, rms 8e-5
    e = err(feat, ops['mlp_feat'])
    assert float(e.max()) < 2e-4 and float(e.mean()) < 3e-5


def test_mlp_stage_production_kernel(ops, relight):
    """the PRODUCTION distance-query kernel (streamed weights, register-resident activations) against the reference's own
    MLP outputs, through Network.inference_observed_distance_field (base_network.py:447-449)"""
    _, net, dev, body, eng = relight
    sdf = net.inference_observed_distance_field(ops['mlp_bpts'][None].to(dev), body)
    assert sdf.shape == (1, ops['mlp_bpts'].shape[0], 1)
    e = err(sdf[0], ops['mlp_sdf'])
    assert float(e.max()) < 6e-4 and float(e.mean()) < 8e-5      # the f16 operand-rounding emulation itself: max 3.4e-4, rms 8e-5


def test_mlp_stage_matches_the_operand_rounding_emulation(relight):
    """bisect of the precision gap: on 20 000 near-surface points the HIP kernel agrees with the oracle's
    kernel-like f16 operand-rounding emulation much better than that emulation agrees with fp32 — the kernel loses nothing
    beyond the rounding of MFMA operands."""
    from oracle import ra_oracle as O
    cfg, net, dev, body, eng = relight
    g = torch.Generator().manual_seed(11)
    d = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
    bpts = d * (0.38 + 0.12 * torch.rand(20000, 1, generator=g))
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    fr = O._frame(synthetic.make_body(0, posed=True))
    f32 = O.observed_sdf(O.OracleNet(sd, cfg), bpts, fr)[:, 0]
    emu = O.observed_sdf(O.OracleNet(sd, cfg, emulate='f16', kernel_like=True), bpts, fr)[:, 0]
    hip = eng.observed_sdf(bpts.to(dev)).cpu()
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt())
    floor, gap, total = rms(emu, f32), rms(hip, emu), rms(hip, f32)
    print(f'sdf rms: emulation-vs-fp32 {floor:.2e}, HIP-vs-emulation {gap:.2e}, HIP-vs-fp32 {total:.2e}')
    assert total < 1.25 * floor, (total, floor)          # HIP is as close to fp32 as the emulation is
    assert gap < 0.6 * floor, (gap, floor)               # and what separates HIP from the emulation is smaller than the rounding itself


def test_mlp_stage_bf16(ops):
    cfg, net, dev = build('relight', dtype='bf16')
    eng = net.set_frame(synthetic.to_device(synthetic.make_body(0, posed=True), dev))
    resd, sdf, feat = eng.debug_mlp(ops['mlp_bpts'].to(dev))
    e = err(sdf[:, None], ops['mlp_sdf'])
    assert float(e.max()) < 5e-3 and float(e.mean()) < 1e-3


def test_compensated_distance_query_is_fp32_accurate(ops):
    """K3C (csrc/ra_k3c.hpp): the distance query with f16 hi + lo operand pairs (three MFMAs per k-step) — the tier the surface trace
    runs in (cfg.trace_precision >= 1).  Against the reference's own MLP outputs and against the fp32 oracle on 20 000 near-surface points it
    must be as accurate as fp32 arithmetic itself (the fp32 oracle is 1.2e-7 rms from a float64 evaluation; plain f16 operands: 5.9e-5),
    and its workgroup widths and the cooperative small-launch variant must agree bit for bit."""
    from oracle import ra_oracle as O
    cfg, net, dev = build('relight', trace_precision=2)         # 2: every distance query in the compensated tier (validation setting)
    body = synthetic.make_body(0, posed=True)
    eng = net.set_frame(synthetic.to_device(body, dev))
    sdf = eng.observed_sdf(ops['mlp_bpts'].to(dev)).cpu()
    e = err(sdf[:, None], ops['mlp_sdf'])
    assert float(e.max()) < 3e-6 and float(e.mean()) < 5e-7, (float(e.max()), float(e.mean()))      # reference outputs (ops.npz); plain f16: 6e-4 / 8e-5
    g = torch.Generator().manual_seed(11)
    d = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
    bpts = d * (0.38 + 0.12 * torch.rand(20000, 1, generator=g))
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    fr = O._frame(body)
    f32 = O.observed_sdf(O.OracleNet(sd, cfg), bpts, fr)[:, 0]
    f64 = O.observed_sdf(O.OracleNet(sd, cfg, emulate='f64acc'), bpts, fr)[:, 0]
    hip = eng.observed_sdf(bpts.to(dev)).cpu()                  # 20 000 points: the 4-wave workgroups
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt())
    print(f'compensated tier: HIP vs float64-accumulated oracle rms {rms(hip, f64):.2e} (fp32 oracle vs the same: {rms(f32, f64):.2e}), max {float((hip - f64).abs().max()):.2e}')
    assert rms(hip, f64) < 4e-7 and float((hip - f64).abs().max()) < 3e-6
    narrow = eng.observed_sdf(bpts[:9000].to(dev)).cpu()        # 9 000 points: the 4-wave workgroups
    assert torch.equal(narrow, hip[:9000])
    # the same points behind a larger upper bound of the fine count (the launcher picks the kernel from the bound, the kernels read the
    # count on the device): 13 k real fine points in a 40 k-point query -> 8-wave tiles, alone -> the 4-wave kernel; 2 k in 23 k / alone -> K3CC
    big = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1) * (0.38 + 0.12 * torch.rand(20000, 1, generator=g)) * 1.02
    far = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1) * 5.0
    for n_near in (20000, 3000):
        eng.reset_counters()
        alone = eng.hdq_sdf(big[:n_near].to(dev), 0.125, True).cpu()
        n_fine = eng.counters().n_fine_sdf_comp
        eng.reset_counters()
        mixed = eng.hdq_sdf(torch.cat([big[:n_near], far]).to(dev), 0.125, True).cpu()
        assert eng.counters().n_fine_sdf_comp == n_fine > 0.5 * n_near       # the far points stop at the coarse level
        assert torch.equal(mixed[:n_near], alone), n_near
    # at most 8 Ki points: K3CC (csrc/ra_k3cc.hpp), four waves sharing a 16-point tile, a quarter of every layer's row blocks each —
    # the same operations on the same operands: bit-identical.  Ragged sizes; 5 000 and 8 192 points need a second round of tiles
    for n in (1, 15, 16, 17, 300, 4096, 4097, 5000, 8192):
        coop = eng.observed_sdf(bpts[:n].to(dev)).cpu()
        assert torch.equal(coop, hip[:n]), n
    # the hierarchical query (coarse level + blend) on world points, ragged size
    x = (bpts * 1.02)[:12345]
    ref = O.hdq_sdf(O.OracleNet(sd, cfg), x, fr, 0.125, True)[:, 0]
    h = eng.hdq_sdf(x.to(dev), 0.125, True).cpu()
    assert float((h - ref).abs().max()) < 5e-6
    c = eng.counters()
    assert c.n_fine_sdf_comp == c.n_fine_sdf > 0


def test_trace_precision_tiers():
    """cfg.trace_precision: 0 = plain 16-bit operands everywhere, 1 (default) = the surface trace compensated, the shadow rays plain"""
    from relightableavatar_amd.renderer import make_renderer
    comp = {}
    for tp in (0, 1):
        cfg, net, dev = build('relight', trace_precision=tp)
        out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(128, 128, seed=0, posed=True, crop=8), dev))
        c = net.engine().counters()
        comp[tp] = (c.n_fine_sdf_comp, c.n_fine_sdf, out.acc_map.clone())
    assert comp[0][0] == 0 and 0 < comp[1][0] < 0.05 * comp[1][1]
    assert float(((comp[0][2] > 0) == (comp[1][2] > 0)).float().mean()) > 0.98


def test_coarse_level_and_warp(ops, relight):
    _, _, dev, body, eng = relight
    o = eng.debug_hdq(ops['hdq_x'].to(dev), 0.125)
    m = ops['knn_fine']
    assert o.fine_count == int(m.sum())
    assert int((o.nn_batch.cpu().long() != ops['knn_nn_batch']).sum()) == 0        # exact 3-NN + geodesic rule
    assert float(err(o.sdf_batch, ops['knn_sdf_batch']).max()) < 1e-6
    assert float(err(o.bpts.cpu()[m], ops['warp_bpts'][m]).max()) < 2e-6
    assert float(err(o.tpts.cpu()[m], ops['warp_tpts'][m]).max()) < 2e-6
    assert float(err(o.mats.cpu()[m][:, :12], ops['warp_A_bw'][m][:, :3].reshape(-1, 12)).max()) < 2e-6
    assert float(err(o.mats.cpu()[m][:, 12:], ops['warp_big_A_bw'][m][:, :3].reshape(-1, 12)).max()) < 2e-6


def test_bvh_equals_brute_force(relight):
    """the per-frame vertex BVH must return the same neighbours as the O(N) scan, also far from the body"""
    _, _, dev, body, eng = relight
    g = torch.Generator().manual_seed(7)
    x = ((torch.rand(100000, 3, generator=g) - 0.5) * 3.0).to(dev)
    # 100 000 queries: one wave per 64 queries; 40 000: the small-launch variant (4 waves share 64 queries and merge)
    for n in (100000, 40000):
        a = eng.debug_hdq(x[:n].contiguous(), 0.125)
        eng.set_knn_mode(False)
        eng.set_frame(body, force=True)
        b = eng.debug_hdq(x[:n].contiguous(), 0.125)
        eng.set_knn_mode(True)
        eng.set_frame(body, force=True)
        assert int((a.nn_batch != b.nn_batch).sum()) == 0
        assert float((a.sdf_coarse - b.sdf_coarse).abs().max()) == 0.0
        assert a.fine_count == b.fine_count


def test_bvh_equals_brute_force_on_other_mesh_sizes(relight):
    """the box structure on meshes that are not SMPL-sized: a handful of vertices (one super box, padded leaves), counts that are
    not multiples of the 32-point leaf or the 8-leaf super box (missing leaves are inverted boxes), a larger mesh (9 500 vertices) and one
    beyond the builder's limit (16 384: brute force) — large and small launches (one wave per 64 queries / the split-wave variants),
    always the same neighbours and distances as the O(N) scan"""
    from relightableavatar_amd.base_utils import dotdict
    _, _, dev, body, eng = relight
    g = torch.Generator().manual_seed(11)
    x = ((torch.rand(70000, 3, generator=g) - 0.5) * 2.0).to(dev)
    n0 = body.pverts.shape[1]
    try:
        for n in (37, 1000, 6887, 9500, 16500):
            rep = (n + n0 - 1) // n0
            b = dotdict(body)
            for k in ('pverts', 'pnorm', 'tverts', 'weights'):
                v = body[k][0]
                v = torch.cat([v + (0.003 * j if k in ('pverts', 'tverts') else 0.0) for j in range(rep)])[:n]      # copies shifted by 3 mm
                b[k] = v[None].contiguous()
            for q in (70000, 20000, 3000):          # one wave per 64 queries / 8 waves / 16 waves per 64 queries
                eng.set_knn_mode(True)
                eng.set_frame(b, force=True)
                a = eng.debug_hdq(x[:q].contiguous(), 0.125)
                eng.set_knn_mode(False)
                eng.set_frame(b, force=True)
                c = eng.debug_hdq(x[:q].contiguous(), 0.125)
                assert int((a.nn_batch != c.nn_batch).sum()) == 0, (n, q)
                assert float((a.sdf_coarse - c.sdf_coarse).abs().max()) == 0.0, (n, q)
                assert a.fine_count == c.fine_count, (n, q)
    finally:
        eng.set_knn_mode(True)
        eng.set_frame(body, force=True)


def test_box_structure_is_morton_sorted(relight):
    """the per-frame vertex order of the box structure IS the ascending (30-bit Morton code of the posed vertex, vertex index) order, for
    every mesh size class (a handful of vertices, counts around the 512 / 4096 / 16384 boundaries of earlier sort variants, the builder's
    limit): bvh_rank_kernel ranks every vertex against all keys — recomputed here in numpy with the kernel's float32 arithmetic"""
    from relightableavatar_amd.base_utils import dotdict
    _, _, dev, body, eng = relight
    n0 = body.pverts.shape[1]

    def expand10(v):
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    try:
        for n in (3, 37, 512, 513, 1000, 1500, 3000, 4096, 6890, 9500, 16384):
            rep = (n + n0 - 1) // n0
            b = dotdict(body)
            for k in ('pverts', 'pnorm', 'tverts', 'weights'):
                v = body[k][0]
                v = torch.cat([v + (0.003 * j if k in ('pverts', 'tverts') else 0.0) for j in range(rep)])[:n]
                b[k] = v[None].contiguous()
            eng.set_knn_mode(True)
            eng.set_frame(b, force=True)
            ids = eng.debug_bvh_ids()
            nl = (n + 31) // 32
            assert ids.shape[0] == 32 * nl and bool((ids[n:] == 0x7fffffff).all()), n
            pv = b.pverts[0].cpu().numpy().astype(np.float32)
            lo, hi = pv.min(0), pv.max(0)
            e = np.maximum(hi - lo, np.float32(1e-12))
            q = np.minimum(np.maximum((pv - lo) / e * np.float32(1023.0), np.float32(0.0)), np.float32(1023.0)).astype(np.uint32)
            code = (expand10(q[:, 0]) << np.uint32(2)) | (expand10(q[:, 1]) << np.uint32(1)) | expand10(q[:, 2])
            key = (code.astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
            assert np.array_equal(ids[:n].astype(np.int64), np.argsort(key, kind='stable')), n
    finally:
        eng.set_frame(body, force=True)


def test_hinted_search_equals_brute_force_in_whole_frames():
    """advisor (round 3): the HINT variants of the coarse kernel (a tracing loop's queries start from the neighbours of the iteration before;
    dedup insert; split-wave merge with hints) are claimed exact but no test compared them bit for bit — debug_hdq never hints.  The O(N)
    scan (`set_knn_mode(False)`, kernel <false, 1>) is never hinted: a relit frame with the ground pass and a bare sphere trace must be
    IDENTICAL in every map with the box structure + hints and with the brute-force scan."""
    from relightableavatar_amd.renderer import make_renderer
    kw = dict(vis_ground_shading=True, ground_normal=[0.0, -1.0, 0.0], ground_origin=[0.0, 0.45, 0.0], vis_specular_map=True)
    outs, traces = [], []
    for bvh in (True, False):
        cfg, net, dev = build('relight', **kw)
        eng = net.engine()
        eng.set_knn_mode(bvh)
        batch = synthetic.to_device(synthetic.make_batch(96, 96, seed=0, posed=True), dev)
        out = make_renderer(cfg, net).render(batch)
        outs.append({k: out[k].clone() for k in ('rgb_map', 'acc_map', 'surf_map', 'norm_map', 'shade_map', 'spec_map', 'albedo_map', 'depth_map')})
        b2 = synthetic.to_device(synthetic.make_batch(128, 128, seed=0, posed=True), dev)
        eng.set_frame(b2, force=True)
        p = eng.trace_params(cfg.sphere_tracing, cfg.dist_th, False)
        traces.append(eng.sphere_trace(b2.ray_o[0], b2.ray_d[0], b2.near[0], b2.far[0], p))      # ra_sphere_trace: hinted from its second iteration on
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(), b.nan_to_num())     # depth = (surf_x - o_x) / d_x is NaN where d_x = 0 (quirk 4)
    for k in outs[0]:
        assert same(outs[0][k], outs[1][k]), k
    for a, b in zip(traces[0], traces[1]):
        assert same(a, b)


def test_hdq_sdf(ops, relight):
    _, net, dev, body, _ = relight
    x = ops['hdq_x'].to(dev)
    s = net.inference_world_distance_field(x[None], body, smooth_transition=True, dist_th=0.125)
    assert s.shape == (1, x.shape[0], 1)
    assert float(err(s[0], ops['hdq_sdf']).max()) < 3e-4
    s = net.inference_world_distance_field(x[None], body, smooth_transition=False, dist_th=0.125)
    assert float(err(s[0], ops['hdq_sdf_nosmooth']).max()) < 3e-4
    coarse = ~ops['knn_fine']
    assert float(err(s[0], ops['hdq_sdf_nosmooth'])[coarse].max()) < 1e-6         # coarse-only points are pure fp32


def test_forward_raw(ops, relight):
    _, net, dev, body, _ = relight
    raw = net(ops['fwd_x'][None].to(dev), None, 0.005, body).raw[0]
    ref = ops['fwd_raw']
    assert raw.shape == ref.shape == (300, 17)
    assert float(err(raw[:, 0:9], ref[:, 0:9]).max()) < 5e-6                       # cpts, bpts, resd
    assert float(err(raw[:, 9:13], ref[:, 9:13]).max()) < 1e-4                     # albedo, roughness
    assert float(err(raw[:, 16], ref[:, 16]).max()) < 1e-4                         # occ
    e = err(raw[:, 13:16], ref[:, 13:16])
    assert float(e.max()) < 8e-3 and float(e.mean()) < 6e-4                        # world normals


def test_forward_zero_outside_dist_th(relight):
    _, net, dev, body, _ = relight
    x = torch.tensor([[[5.0, 5.0, 5.0], [0.0, 0.0, 3.0]]], device=dev)
    assert float(net(x, None, 0.005, body).raw.abs().max()) == 0.0


def test_sphere_tracing(ops, relight):
    cfg, _, dev, _, eng = relight
    p = eng.trace_params(cfg.sphere_tracing, cfg.dist_th, False)
    surf, occ, st, ot = eng.sphere_trace(ops['st_o'].to(dev), ops['st_d'].to(dev), ops['st_near'].to(dev), ops['st_far'].to(dev), p)
    assert int(((occ.cpu() < 1) != (ops['st_occ'][:, 0] < 1)).sum()) <= 2           # hit mask
    e = err(st[:, None], ops['st_st'])
    assert float(e.median()) < 2e-4 and float((e < 3e-3).float().mean()) > 0.97
    p = eng.trace_params(cfg.obj_lvis, 0.125, True)
    n = ops['sh_o'].shape[0]
    _, occ, _, _ = eng.sphere_trace(ops['sh_o'].to(dev), ops['sh_d'].to(dev), torch.full((n,), 0.02, device=dev),
                                    torch.full((n,), 0.8, device=dev), p, tan_i=ops['sh_tan_i'].to(dev))
    e = err(occ[:, None], ops['sh_occ'])
    assert float(e.mean()) < 5e-3 and float((e < 2e-2).float().mean()) > 0.97


def test_sphere_tracing_stage_in_compensated_arithmetic(ops):
    """the same two stage fixtures (surface trace, DFSS shadow trace through the real HDQ; tests/golden/ops.npz, made by the reference's
    sphere_tracing :103-216) with EVERY distance query compensated (cfg.trace_precision 2): what remains of the percentile tolerances of
    test_sphere_tracing once the arithmetic is as good as fp32 — i.e. how much of them is the operands' rounding and how much the state
    machine's own discontinuities (limit cycles of the surface trace, accept conditions of the claybook estimate) on the noisy body"""
    cfg, net, dev = build('relight', trace_precision=2)
    body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
    eng = net.set_frame(body)
    p = eng.trace_params(cfg.sphere_tracing, cfg.dist_th, False)
    surf, occ, st, ot = eng.sphere_trace(ops['st_o'].to(dev), ops['st_d'].to(dev), ops['st_near'].to(dev), ops['st_far'].to(dev), p)
    e = err(st[:, None], ops['st_st'])
    hit_diff = int(((occ.cpu() < 1) != (ops['st_occ'][:, 0] < 1)).sum())
    print(f'surface trace, all compensated: hit-mask differences {hit_diff}, |st err| median {float(e.median()):.2e}, within 1e-4: {float((e < 1e-4).float().mean()) * 100:.1f} %, '
          f'within 3e-3: {float((e < 3e-3).float().mean()) * 100:.1f} %, max {float(e.max()):.2e}')
    # measured: 0 hit-mask differences, median 2.4e-7, 99.8 % within 1e-4 (one of 400 rays is one the reference's own fp32 does not pin)
    assert hit_diff <= 1 and float(e.median()) < 2e-6 and float((e < 1e-4).float().mean()) > 0.99
    p = eng.trace_params(cfg.obj_lvis, 0.125, True)
    n = ops['sh_o'].shape[0]
    _, occ, _, _ = eng.sphere_trace(ops['sh_o'].to(dev), ops['sh_d'].to(dev), torch.full((n,), 0.02, device=dev),
                                    torch.full((n,), 0.8, device=dev), p, tan_i=ops['sh_tan_i'].to(dev))
    e = err(occ[:, None], ops['sh_occ'])
    print(f'shadow trace, all compensated: |occ err| mean {float(e.mean()):.2e}, within 1e-3: {float((e < 1e-3).float().mean()) * 100:.1f} %, '
          f'within 2e-2: {float((e < 2e-2).float().mean()) * 100:.1f} %, max {float(e.max()):.2e}')
    # measured: mean 5.0e-6, max 2.2e-4 (plain f16 operands, test_sphere_tracing: mean < 5e-3, 97 % within 2e-2): the stage-level percentile
    # tolerances of the plain tier are the operands' rounding amplified by d * sharp / (2 t), not a defect of the state machine
    assert float(e.mean()) < 5e-5 and float(e.max()) < 1e-3


def _frame(mode, fname, golden, **kw):
    from relightableavatar_amd.renderer import make_renderer
    ref = golden(fname)
    if mode == 'anisdf':
        kw['n_samples'] = int(ref['n_samples'])
    cfg, net, dev = build(mode, **kw)
    batch = synthetic.to_device(synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']),
                                                     n_novel_lights=3 if mode == 'novel_light' else 0), dev)
    return make_renderer(cfg, net).render(batch), ref, batch, net


def within(out, ref, key, tol, frac):
    e = err(out[key], ref[key])
    ok = float((e <= tol).float().mean())
    assert ok >= frac, f'{key}: only {ok * 100:.1f}% of elements within {tol}'


def test_frame_anisdf_volume(golden):
    out, ref, _, _ = _frame('anisdf', 'frame_anisdf.npz', golden)
    for k, tol in (('acc_map', 5e-4), ('depth_map', 1e-3), ('cpts_map', 2e-4), ('resd_map', 1e-5), ('norm_map', 2e-3), ('rgb_map', 3e-4)):
        within(out, ref, k, tol, 1.0)
    assert psnr(out.rgb_map, ref['rgb_map']) > 80


def test_volume_padding_slots_are_not_queried():
    """the volume path lays its samples out in groups of 64 rays; the padding slots of the last group must neither cost nor count
    as full queries (round-1 advisor note: they used to repeat the last ray).  P identical rays through the body: the number of
    full queries is exactly proportional to P, and every ray gets the same pixel"""
    cfg, net, dev = build('anisdf')
    body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
    eng = net.set_frame(body)
    ctr = (0.5 * (body.wbounds[0, 0] + body.wbounds[0, 1])).float()
    o1 = ctr + torch.tensor([0.0, 0.0, 2.0], device=dev)
    d1 = torch.nn.functional.normalize(ctr - o1, dim=0)
    counts, pix = {}, {}
    for P in (64, 65, 127):
        ro, rd = o1.repeat(P, 1).contiguous(), d1.repeat(P, 1).contiguous()
        near, far = torch.full((P,), 1.0, device=dev), torch.full((P,), 3.0, device=dev)
        outs = dict(rgb=torch.zeros(P, 3, device=dev), acc=torch.zeros(P, device=dev))
        c0 = eng.counters().n_fine_full
        eng.render_volume_chunk(ro, rd, near, far, cfg.n_samples, cfg.dist_th, outs)
        counts[P] = eng.counters().n_fine_full - c0
        pix[P] = (outs['rgb'].clone(), outs['acc'].clone())
        assert float(outs['acc'].min()) > 0.05 and bool((outs['rgb'] == outs['rgb'][0]).all()) and bool((outs['acc'] == outs['acc'][0]).all())   # not a miss; all rays alike
    per_ray = counts[64] // 64
    assert per_ray > 4 and counts[64] == 64 * per_ray and counts[65] == 65 * per_ray and counts[127] == 127 * per_ray, counts
    assert torch.equal(pix[64][0][0], pix[65][0][64]) and torch.equal(pix[64][0][0], pix[127][0][126])


def test_frame_sphere_tracing(golden):
    out, ref, _, _ = _frame('sphere_tracing', 'frame_sphere.npz', golden)
    hit, hit_ref = out.acc_map.cpu() > 0, T(ref['acc_map']) > 0
    assert float((hit == hit_ref).float().mean()) > 0.99
    within(out, ref, 'acc_map', 2e-2, 0.98)
    within(out, ref, 'surf_map', 2e-3, 0.98)
    within(out, ref, 'norm_map', 2e-2, 0.97)
    within(out, ref, 'rgb_map', 5e-3, 0.98)
    assert psnr(out.rgb_map, ref['rgb_map']) > 50


def trimmed_psnr(a, b, keep=0.98):
    a, b = a.detach().float().cpu().reshape(-1, a.shape[-1]), torch.as_tensor(b).float().reshape(-1, a.shape[-1])
    pp = (a - b).abs().amax(-1)
    m = pp <= pp.kthvalue(max(1, int(round(keep * pp.numel())))).values
    return float(-10 * torch.log10(((a - b)[m] ** 2).mean()))


def unstable_rays(case, n_rays):
    """bool mask of the rays of a parity ray set whose traced surface the reference's own fp32 arithmetic does not pin
    (tests/golden/fp32_unstable_rays.json, tools/fp32_stability.py: 3e-7 noise on the distances flips them)"""
    here = os.path.dirname(os.path.abspath(__file__))
    d = json.load(open(os.path.join(here, 'golden', 'fp32_unstable_rays.json')))[case]
    assert d['n_rays'] == n_rays, (case, d['n_rays'], n_rays)
    m = torch.zeros(n_rays, dtype=torch.bool)
    m[d['unstable']] = True
    return m


def unstable_info(case):
    here = os.path.dirname(os.path.abspath(__file__))
    return json.load(open(os.path.join(here, 'golden', 'fp32_unstable_rays.json'))).get(case)


def assert_contract(rgb, rgb_ref, case, label=None, bad=None, all_rays=False):
    """SURVEY.md:409 for the 16-bit path, asserted outright on every ray whose reference value fp32 itself pins: rgb PSNR >= 50 dB and
    max |err| <= 1e-2.  About 0.5 % of the rays are coin tosses of the reference's own arithmetic (unstable_rays): a change of the last
    bit of a distance moves their surface point by millimetres, so which side of the toss an implementation lands on changes with any
    re-association (the 32x32 and the 16x16 tile of the compensated kernel land differently on `frame_relight` and on
    `frame_relight_smooth`); one such ray at 0.05 rgb takes a 256-ray frame from 64 to 51 dB.  The figures over ALL rays are printed."""
    e = err(rgb, rgb_ref)
    e = e.reshape(-1, e.shape[-1])
    bad = unstable_rays(case, e.shape[0]) if bad is None else bad
    p_all = float(-10 * torch.log10(torch.mean(e ** 2)))
    p = float(-10 * torch.log10(torch.mean(e[~bad] ** 2)))
    mx, mx_all = float(e[~bad].max()), float(e.max())
    print(f'{label or case}: rgb PSNR {p:.1f} dB, max |err| {mx:.2e} over the {int((~bad).sum())} fp32-stable rays '
          f'({int(bad.sum())} unstable; over all rays {p_all:.1f} dB, max {mx_all:.2e}), stable rays over 1e-2: {int((e[~bad].amax(-1) > 1e-2).sum())}')
    assert p >= 50.0, (label or case, p)
    assert mx <= 1e-2, (label or case, mx)
    assert p_all >= 40.0, (label or case, p_all)          # a sanity bound only: the unstable rays are a handful
    if all_rays:                                          # where every ray is pinned well enough, the contract's PSNR half holds over all of them
        assert p_all >= 50.0, (label or case, p_all)
    listed = unstable_info(case)
    if listed and int(bad.sum()):
        print(f'   fp32-unstable rays of {case}: {listed["unstable"]}, flip probability at noise 1.2e-7 (fp32\'s own level): '
              f'{listed.get("flip_probability", {}).get("noise_1.2e-7")}, at 3e-7: {listed.get("flip_probability", {}).get("noise_3e-7")}')
    return p, mx


def floor_of(golden_dir_name, dtype='f16'):
    here = os.path.dirname(os.path.abspath(__file__))
    return json.load(open(os.path.join(here, 'golden', 'precision_floor.json')))[f'{golden_dir_name}:{dtype}']


def test_frame_relight_smooth_meets_the_contract(golden):
    """SURVEY.md:409: rgb PSNR >= 50 dB and max |err| <= 1e-2 against the reference (smooth skinning field); with the surface trace in
    compensated arithmetic the traced surface agrees to 1e-4 on every ray fp32 pins (plain f16 operands: 3e-4 .. 4e-3)"""
    ref = golden('frame_relight_smooth.npz')
    cfg, net, dev = build('relight', vis_specular_map=True)
    from relightableavatar_amd.renderer import make_renderer
    batch = synthetic.to_device(synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']),
                                                     skin_noise=float(ref['skin_noise'])), dev)
    out = make_renderer(cfg, net).render(batch)
    assert bool(((out.acc_map.cpu() > 0) == (T(ref['acc_map']) > 0)).all())
    assert_contract(out.rgb_map, ref['rgb_map'], 'frame_relight_smooth.npz')
    ok = ~unstable_rays('frame_relight_smooth.npz', out.rgb_map.shape[1])
    assert float(err(out.albedo_map, ref['albedo_map'])[0][ok].max()) < 1e-3 and float(err(out.surf_map, ref['surf_map'])[0][ok].max()) < 1e-4
    assert psnr(out.shade_map, ref['shade_map']) >= 50.0
    # row H's full key set: render_human's per-hit leftovers (sphere_tracing_renderer.py:616-650) are in the output (lazily: reading
    # them costs the hit count's read-back) and match the reference's as sets of rows (its hit order is topk's, ours ascending)
    assert {'raw', 'volume_albedo', 'volume_roughness'} <= set(out.keys())
    for k, tol in (('volume_albedo', 1e-3), ('volume_roughness', 1e-3), ('raw', 3e-2)):      # raw carries the f16 normals
        a, b = out[k][0].double().cpu(), T(ref[k])[0].double()
        assert a.shape == b.shape, (k, a.shape, b.shape)
        d = torch.cdist(a, b, p=float('inf'))
        if k == 'raw':      # the samples' f16 normals: single samples off a crease reach 0.09 (the composited norm_map is what is shaded)
            m1, m0 = d.min(1).values, d.min(0).values
            assert float((m1 < tol).float().mean()) > 0.99 and float((m0 < tol).float().mean()) > 0.99 and float(m1.median()) < 3e-3, (k, float(m1.max()))
        else:
            assert float(d.min(1).values.max()) < tol and float(d.min(0).values.max()) < tol, (k, float(d.min(1).values.max()))


# ---- the hot path's configuration switches: the reference under each override (tests/golden/switches.npz, one process per variant;
# tests/test_oracle_frames.py pins the oracle on the same file)
from test_oracle_frames import (GROUND_SWITCH_NAMES, HARD_NOVEL_NAMES, HARD_SWITCH_NAMES, NOVEL_SWITCH_NAMES, SPHERE_SWITCH_NAMES, SWITCH_NAMES, VOLUME_SWITCH_NAMES,      # noqa: E402
                                hard_novel_case, novel_switch_case,
                                switch_batch, switch_batch_kw, switch_cfg, switch_state_dict, switch_variants, volume_switch_cfg)


@pytest.mark.parametrize('name', SWITCH_NAMES)
def test_switch_matrix(golden, name):
    """every switch of render_human / light_visibility / the microfacet model / the K-NN rule the reference's configs can flip
    (sphere_tracing_renderer.py:36,275,295-300,720-757; relight_utils.py:563-566; relight_network.py:63-66; sample_utils.py:116),
    one relit frame each against the reference's own output: SURVEY.md:409's contract on rgb, the other maps to their tolerances"""
    from relightableavatar_amd.networks import make_network
    from relightableavatar_amd.renderer import make_renderer
    ref = golden('switches.npz')
    dev = _dev()
    cfg = switch_cfg(switch_variants(ref)[name], mlp_dtype='f16')
    bkw = switch_batch_kw(switch_variants(ref)[name])
    net = make_network(cfg)
    net.load_state_dict(switch_state_dict(bkw, cfg))
    net = net.to(dev).eval()
    batch = synthetic.to_device(switch_batch(ref, bkw), dev)
    out = make_renderer(cfg, net).render(batch)
    sub = {k[len(name) + 1:]: v for k, v in ref.items() if k.startswith(name + '.')}
    for k in ('rgb_map', 'shade_map', 'spec_map'):      # maps_only: render_human's early return (:702-705) leaves none of them
        assert (k in sub) == (k in out), (name, k)
    assert bool(((out.acc_map.cpu() > 0) == (T(sub['acc_map']) > 0)).all())
    case = 'switches.npz:' + (name if name in ('trace_params', 'no_geodesic_filter', 'smpl24', 'other_weights', 'all_shadowed') else 'base')
    assert float(err(out.surf_map, sub['surf_map']).max()) < 1e-4
    assert float(err(out.albedo_map, sub['albedo_map']).max()) < 1e-3 and float(err(out.roughness_map, sub['roughness_map']).max()) < 1e-3
    assert float((err(out.norm_map, sub['norm_map']) < 2e-2).float().mean()) > 0.97
    if 'rgb_map' in sub:
        assert_contract(out.rgb_map, sub['rgb_map'], case, f'switches.npz / {name}', all_rays=True)
        assert out.shade_map.shape == T(sub['shade_map']).shape, (name, out.shade_map.shape)
        assert psnr(out.shade_map, sub['shade_map']) >= 50.0 and float(err(out.shade_map, sub['shade_map']).max()) < 2e-2
    else:
        assert net.engine().counters().n_shadow_rays == 0          # and no light visibility was traced for it
    if 'spec_map' in sub:
        within(out, sub, 'spec_map', 5e-3, 0.97)
    if name + '.hdq_x' in ref:       # the distance field all around the body (base / no_geodesic_filter: the neighbour rule)
        x = T(ref[name + '.hdq_x']).to(dev)
        s = net.engine().hdq_sdf(x, cfg.dist_th, True).cpu()
        e = err(s.reshape(-1), ref[name + '.hdq_sdf'].reshape(-1))
        assert float(e.max()) < 3e-4, float(e.max())
        # the coarse level's own answer (mean of the per-neighbour signed distances, or knn_with_filter's one value) is pure fp32
        dbg = net.engine().debug_hdq(x, cfg.dist_th)
        assert float(err(dbg.sdf_batch.mean(-1), ref[name + '.hdq_sdf_coarse']).max()) < 2e-6


@pytest.mark.parametrize('name', GROUND_SWITCH_NAMES)
def test_ground_switch_matrix(golden, name):
    """the switches of the ground-plane pass (render_ground :463-548: hard shadows, the visibility / cosine maps, linear output, local
    visibility, a plain ground colour + shading multiplier, the env_lvis trace settings incl. the box margin) against the reference's frames"""
    from relightableavatar_amd.networks import make_network
    from relightableavatar_amd.renderer import make_renderer
    ref = golden('switches.npz')
    dev = _dev()
    cfg = switch_cfg(switch_variants(ref)[name], mlp_dtype='f16')
    bkw = switch_batch_kw(switch_variants(ref)[name])
    net = make_network(cfg)
    net.load_state_dict(switch_state_dict(bkw, cfg))
    net = net.to(dev).eval()
    H = int(ref['ground_H'])
    batch = synthetic.to_device(switch_batch(ref, bkw, ground=True), dev)
    rend = make_renderer(cfg, net)
    m = batch.mask_at_box.reshape(1, -1).cpu()
    rend.ground_inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]   # the scatter order of the CPU reference run
    out = rend.render(batch)
    sub = {k[len(name) + 1:]: v for k, v in ref.items() if k.startswith(name + '.')}
    np.testing.assert_allclose(batch.wbounds.cpu().numpy(), sub['wbounds_after'], atol=1e-6)
    assert_contract(out.rgb_map, sub['rgb_map'], 'switches.npz:ground', f'switches.npz / {name}', bad=torch.zeros(H * H, dtype=torch.bool))
    # hard shadows (visibility = clip(500 d / t)): one penumbra-less edge pixel of 576 lands at 7.8e-3
    # (g_split_body: one of 576 pixels at the edge of the horn's shadow on the ground: 6.3e-3)
    assert float((err(out.rgb_map, sub['rgb_map']) < 5e-3).float().mean()) > (0.995 if name == 'g_no_dfss' else (0.998 if name == 'g_split_body' else 0.999))
    assert float(err(out.albedo_map, sub['albedo_map']).max()) < 1e-2
    assert float((err(out.shade_map, sub['shade_map']) < 5e-3).float().mean()) > 0.99
    assert float((err(out.spec_map, sub['spec_map']) < 5e-3).float().mean()) > 0.99
    assert float(err(out.acc_map, sub['acc_map']).max()) < 3e-2


def _hard_case(ref, name, trace_precision, key_light_share):
    from relightableavatar_amd.networks import make_network
    from relightableavatar_amd.renderer import make_renderer
    dev = _dev()
    cfg = switch_cfg(switch_variants(ref)[name], mlp_dtype='f16', trace_precision=trace_precision, key_light_share=key_light_share)
    bkw = switch_batch_kw(switch_variants(ref)[name])
    net = make_network(cfg)
    net.load_state_dict(switch_state_dict(bkw, cfg))
    net = net.to(dev).eval()
    out = make_renderer(cfg, net).render(synthetic.to_device(switch_batch(ref, bkw), dev))
    return cfg, net, dev, out


# the tiers a hard case is rendered with: (label, cfg.trace_precision, cfg.key_light_share)
HARD_CASE_TIERS = (('round-5 tiers (surface trace compensated, every shadow ray plain f16)', 1, 0.0),
                   ('shipped tiers (+ the shadow rays towards the key lights compensated)', 1, 0.0078),
                   ('every distance query compensated', 2, 0.0))
# round-5 tiers on the hard cases: (max |err| bound, rays over 1e-2) where SURVEY.md:409's max half is NOT met — measured on the GPU
# (profiles/r06_hard_cases.txt): what the key-light tier is for
HARD_CASE_ROUND5_MAX = {'split_body': (2e-2, 2), 'sharp_split': (3e-2, 2)}


@pytest.mark.parametrize('name', HARD_SWITCH_NAMES)
def test_hard_case_switch_matrix(golden, name):
    """The reference's own hard cases (round 6; tests/golden/make_golden.py SPLIT_BODY / synthetic.SHARP_BANDS), made by the reference:
    a body part that shadows another at distance under a key light (sphere_tracing_renderer.py:157-179, 265-344) with 12 and with the
    default 4 shadow iterations, trained-like weights with live high-frequency encoding columns (net_utils.py:1303-1352), both on the
    noisy skinning field.  Each is rendered in three tiers (HARD_CASE_TIERS) and the figures of all are printed.  SURVEY.md:409's contract
    (PSNR >= 50 dB, max |err| <= 1e-2 on the rays fp32 pins) is asserted for the SHIPPED tiers and for the all-compensated tier on every
    case; round 5's tiers (no key lights) keep the PSNR half and miss the max half on two cases by the documented amounts."""
    ref = golden('switches.npz')
    sub = {k[len(name) + 1:]: v for k, v in ref.items() if k.startswith(name + '.')}
    case = 'switches.npz:' + name
    res = []
    for label, tp, share in HARD_CASE_TIERS:
        cfg, net, dev, out = _hard_case(ref, name, tp, share)
        assert bool(((out.acc_map.cpu() > 0) == (T(sub['acc_map']) > 0)).all()), (name, label)
        ok = ~unstable_rays(case, out.rgb_map.shape[1])
        e = err(out.rgb_map, sub['rgb_map'])[0]
        pe = e.amax(-1)
        cnt = net.engine().counters()
        r = dict(psnr=float(-10 * torch.log10((e[ok] ** 2).mean())), max=float(e[ok].max()), over=int((pe[ok] > 1e-2).sum()),
                 surf=float(err(out.surf_map, sub['surf_map'])[0][ok].max()), norm=float(err(out.norm_map, sub['norm_map'])[0][ok].max()),
                 albedo=float(err(out.albedo_map, sub['albedo_map'])[0][ok].max()), shade=psnr(out.shade_map, sub['shade_map']),
                 comp_share=cnt.n_fine_sdf_comp / max(cnt.n_fine_sdf, 1))
        res.append(r)
        print(f'switches.npz / {name}, {label}: ' + ', '.join(f'{k} {v:.3g}' for k, v in r.items()) + f' ({int((~ok).sum())} fp32-unstable rays)')
        assert r['surf'] < (5e-4 if 'sharp' in name else 1e-4), r          # the surface trace is compensated in every tier
        assert r['albedo'] < 2e-3
        assert r['psnr'] >= 50.0, (name, label, r)
        if name + '.hdq_x' in ref and tp == 1 and share == 0.0:       # sharp_weights: the distance field all around the body, plain f16 operands
            s = net.engine().hdq_sdf(T(ref[name + '.hdq_x']).to(dev), cfg.dist_th, True).cpu()
            es = err(s.reshape(-1), ref[name + '.hdq_sdf'].reshape(-1))
            print(f'   hdq_sdf on 3 000 points, sharp weights: max {float(es.max()):.2e}, mean {float(es.mean()):.2e} (near-initialisation weights: 3e-4 bound)')
            assert float(es.max()) < 1e-3
        if tp == 1 and share > 0:      # the shipped tiers: the contract, on every case
            assert_contract(out.rgb_map, sub['rgb_map'], case, f'switches.npz / {name} ({label})')
            # ... at a price of a few per cent: the key lights' rays are a small part of the frame's distance queries
            assert r['comp_share'] < 0.25, r
    assert_contract(out.rgb_map, sub['rgb_map'], case, f'switches.npz / {name} (every distance query compensated)')
    lim = HARD_CASE_ROUND5_MAX.get(name, (1e-2, 0))
    assert res[0]['max'] <= lim[0] and res[0]['over'] <= lim[1], (name, res[0], lim)
    assert res[1]['max'] <= res[0]['max'] + 1e-3          # the key-light tier never makes a case worse


@pytest.mark.parametrize('name', HARD_NOVEL_NAMES)
def test_hard_case_novel_light(golden, name):
    """The hard-case body through the novel-light renderer, against the reference's own frames: ONE trace under the learned map, re-shaded
    under a lognormal probe and an OLAT-style one (a light of 100 over an ambient 0.25).  The frame's key lights are those of ALL its probes
    (ra_set_key_probes, called by the renderer's mirror before the trace): with round 5's tiers the OLAT probe's frame is where plain-f16
    shadow rays show; SURVEY.md:409's contract is asserted for the shipped tiers on every output."""
    from relightableavatar_amd.networks import make_network
    from relightableavatar_amd.renderer import make_renderer
    ref = golden('switches.npz')
    dev = _dev()
    case = 'switches.npz:split_body'          # the same window and surface trace as split_body
    res = {}
    for label, tp, share in HARD_CASE_TIERS:
        cfg, env, mk, want, names = hard_novel_case(ref, name, mlp_dtype='f16', trace_precision=tp, key_light_share=share)
        net = make_network(cfg)
        net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg, env=env))
        net = net.to(dev).eval()
        out = make_renderer(cfg, net).render(synthetic.to_device(mk(), dev))
        assert [k for k in out if k != 'diff'] == names
        for out_name, maps in want.items():
            ok = ~unstable_rays(case, out[out_name].rgb_map.shape[1])
            e = err(out[out_name].rgb_map, maps['rgb_map'])[0]
            r = dict(psnr=float(-10 * torch.log10((e[ok] ** 2).mean())), max=float(e[ok].max()), over=int((e[ok].amax(-1) > 1e-2).sum()))
            res[(label, out_name)] = r
            print(f'switches.npz / {name} / {out_name}, {label}: ' + ', '.join(f'{k} {v:.3g}' for k, v in r.items()))
            assert r['psnr'] >= 50.0, (label, out_name, r)
            if share > 0 or tp == 2:
                assert_contract(out[out_name].rgb_map, maps['rgb_map'], case, f'switches.npz / {name} / {out_name} ({label})')
    shipped, round5 = HARD_CASE_TIERS[1][0], HARD_CASE_TIERS[0][0]
    # measured (profiles/r06_hard_cases.txt): round 5's tiers main 58.2 dB / 1.16e-2, the lognormal probe 57.1 / 1.01e-2, OLAT 65.9 / 3.0e-3;
    # shipped 71.0 / 2.5e-3, 63.7 / 4.5e-3, 68.1 / 1.8e-3.  (With at most 24 key lights per frame the lognormal probe, whose power is spread
    # over ~60 lights at 2-6 x the mean, kept one ray at 1.05e-2: the learned map's key light had taken 14 of the 24; hence 48.)
    assert res[(shipped, 'main')]['max'] < 0.5 * res[(round5, 'main')]['max']


@pytest.mark.parametrize('name', SPHERE_SWITCH_NAMES)
def test_sphere_switch_matrix(golden, name):
    """config 3's path (sphere tracing of the AniSDF network: surface trace, full query, colour net on the traced normals; no relighting)
    with the trained-like weights, against the reference's frame: SURVEY.md:409's contract over all rays"""
    from relightableavatar_amd.networks import make_network
    from relightableavatar_amd.renderer import make_renderer
    ref = golden('switches.npz')
    dev = _dev()
    ov = switch_variants(ref)[name]
    cfg = make_cfg('sphere_tracing', mlp_dtype='f16')
    net = make_network(cfg)
    net.load_state_dict(synthetic.make_state_dict(0, relight=False, cfg=cfg, kind=ov.get('@weights_kind', 'init')))
    net = net.to(dev).eval()
    out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']), skin_noise=0.0), dev))
    sub = {k[len(name) + 1:]: v for k, v in ref.items() if k.startswith(name + '.')}
    assert bool(((out.acc_map.cpu() > 0) == (T(sub['acc_map']) > 0)).all())
    p, mx = psnr(out.rgb_map, sub['rgb_map']), float(err(out.rgb_map, sub['rgb_map']).max())
    print(f'switches.npz / {name}: rgb PSNR {p:.1f} dB, max {mx:.2e}; surf max {float(err(out.surf_map, sub["surf_map"]).max()):.2e}, normals max {float(err(out.norm_map, sub["norm_map"]).max()):.2e}')
    assert p >= 50.0 and mx <= 1e-2
    assert float(err(out.surf_map, sub['surf_map']).max()) < 1e-4 and float((err(out.norm_map, sub['norm_map']) < 2e-2).float().mean()) > 0.99


@pytest.mark.parametrize('name', VOLUME_SWITCH_NAMES)
def test_volume_switch_matrix(golden, name):
    """the volume renderer's switches (base_renderer.py:17,72,120-121) against the reference: background brightness, an active near / far
    clip (ra_config.clip_near / clip_far), another sample count over several render chunks"""
    from relightableavatar_amd.networks import make_network
    from relightableavatar_amd.renderer import make_renderer
    ref = golden('switches.npz')
    dev = _dev()
    cfg = volume_switch_cfg(switch_variants(ref)[name], mlp_dtype='f16')
    net = make_network(cfg)
    net.load_state_dict(synthetic.make_state_dict(0, relight=False, cfg=cfg, kind=switch_variants(ref)[name].get('@weights_kind', 'init')))
    net = net.to(dev).eval()
    H = int(ref['volume_H'])
    batch = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True, crop=int(ref['volume_crop']), skin_noise=0.0), dev)
    out = make_renderer(cfg, net).render(batch)
    sub = {k[len(name) + 1:]: v for k, v in ref.items() if k.startswith(name + '.')}
    if name == 'v_sharp_weights':      # round 6: trained-like weights (synthetic.SHARP_BANDS) — every sample is a full query on plain f16 operands
        p = psnr(out.rgb_map, sub['rgb_map'])
        print(f'switches.npz / {name}: rgb PSNR {p:.1f} dB, max {float(err(out.rgb_map, sub["rgb_map"]).max()):.2e}; normals max {float(err(out.norm_map, sub["norm_map"]).max()):.2e}, '
              f'depth max {float(err(out.depth_map, sub["depth_map"]).max()):.2e}, acc max {float(err(out.acc_map, sub["acc_map"]).max()):.2e}')
        assert p >= 50.0 and float(err(out.rgb_map, sub['rgb_map']).max()) <= 1e-2          # SURVEY.md:409
        within(out, sub, 'acc_map', 2e-3, 1.0)
        within(out, sub, 'norm_map', 2e-2, 0.99)
        return
    # v_bg: volume_rendering adds (1 - acc) * bg_brightness to EVERY composited channel (net_utils.py:970-999: cpts, resd, norm too), so
    # each map carries the f16 path's alpha error (<= 5e-4) times 0.5
    bg = name == 'v_bg'
    for k, tol in (('acc_map', 5e-4), ('depth_map', 1e-3), ('cpts_map', 5e-4 if bg else 2e-4), ('resd_map', 5e-4 if bg else 1e-5), ('norm_map', 2e-3),
                   ('rgb_map', 5e-4 if bg else 3e-4)):
        within(out, sub, k, tol, 1.0)
    assert psnr(out.rgb_map, sub['rgb_map']) > (70 if bg else 80)


@pytest.mark.parametrize('name', NOVEL_SWITCH_NAMES)
def test_novel_switch_matrix(golden, name):
    """cfg.vis_rotate_light (novel_light_sphere_tracing.py:163-171, relight_utils.py:55-110) against the reference: the sequence of names,
    the rotated probe of every stored heading and its re-shaded frame; with the ground pass the rotated IMAGE colours the ground and
    every heading is blended on its own"""
    from relightableavatar_amd.networks import make_network
    from relightableavatar_amd.renderer import make_renderer
    ref = golden('switches.npz')
    cfg, mk, want, names = novel_switch_case(ref, name)
    cfg.mlp_dtype = 'f16'
    dev = _dev()
    net = make_network(cfg)
    net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg))
    net = net.to(dev).eval()
    batch = synthetic.to_device(mk(), dev)
    rend = make_renderer(cfg, net)
    m = batch.mask_at_box.reshape(1, -1).cpu()
    if 'ground' in name:
        rend.ground_inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]
    out = rend.render(batch)
    assert [k for k in out if k != 'diff'] == names
    for out_name, maps in want.items():
        if out_name != 'main':
            assert float(err(out[out_name].envmap.probe.reshape(maps['probe'].shape), maps['probe']).max()) < 1e-5
        e = err(out[out_name].rgb_map, maps['rgb_map'])
        p = float(-10 * torch.log10(torch.mean(e ** 2)))
        print(f'{name} / {out_name}: rgb PSNR {p:.1f} dB, max {float(e.max()):.2e}')
        assert p >= 50.0 and float(e.max()) <= 1e-2, (out_name, p, float(e.max()))
        assert float((err(out[out_name].shade_map, maps['shade_map']) < 5e-3).float().mean()) > 0.99
        assert float((err(out[out_name].spec_map, maps['spec_map']) < 5e-3).float().mean()) > 0.97
        assert float(err(out[out_name].albedo_map, maps['albedo_map']).max()) < 1e-2


def test_frame_relight(golden):
    out, ref, batch, net = _frame('relight', 'frame_relight.npz', golden, vis_specular_map=True)
    # the SURVEY 8d body (white noise in the skinning logits): the contract itself, no emulation-derived floor (round 3 asserted floor - 3 dB;
    # plain f16 operands in the surface trace reach 50.7 dB / max 4.7e-2 here, the compensated tier 63.8 dB / 8.4e-3)
    assert_contract(out.rgb_map, ref['rgb_map'], 'frame_relight.npz', 'frame_relight (SURVEY 8d body)')
    np.testing.assert_allclose(batch.wbounds.cpu().numpy(), ref['wbounds_after'], atol=1e-6)     # in-place bbox growth quirk
    assert bool(((out.acc_map.cpu() > 0) == (T(ref['acc_map']) > 0)).all())
    within(out, ref, 'albedo_map', 5e-4, 0.99)
    within(out, ref, 'roughness_map', 5e-4, 0.99)
    within(out, ref, 'surf_map', 1e-4, 0.98)
    within(out, ref, 'norm_map', 2e-2, 0.97)
    within(out, ref, 'shade_map', 2e-2, 0.97)
    within(out, ref, 'spec_map', 5e-3, 0.97)
    within(out, ref, 'rgb_map', 1e-2, 0.99)
    c = net.engine().counters()
    assert c.n_hit_pixels == 256 and c.n_shadow_rays > 0 and c.n_fine_sdf > c.n_shadow_rays
    assert 0 < c.n_fine_sdf_comp < 0.05 * c.n_fine_sdf          # the surface trace's queries ran in the compensated tier: 2 % of the frame's


def test_frame_novel_light(golden):
    out, ref, batch, _ = _frame('novel_light', 'frame_novel.npz', golden)
    assert set(out.keys()) == {'main', 'diff', *batch.novel_lights.keys()}
    within(out.main, {k[5:]: v for k, v in ref.items() if k.startswith('main.')}, 'rgb_map', 1e-2, 0.97)
    for n in batch.novel_lights:
        sub = {k[len(n) + 1:]: v for k, v in ref.items() if k.startswith(n + '.')}
        within(out[n], sub, 'rgb_map', 1e-2, 0.97)
        within(out[n], sub, 'shade_map', 2e-2, 0.97)
        within(out[n], sub, 'spec_map', 5e-3, 0.97)
        assert_contract(out[n].rgb_map, sub['rgb_map'], 'frame_novel.npz', f'frame_novel {n}')
    # the cached per-light visibility / cosine of the main pass (what every probe is re-shaded from) against the reference's
    sub = {k[len('probe00.'):]: v for k, v in ref.items() if k.startswith('probe00.')}
    within(out['probe00'], sub, 'ldot_map', 2e-2, 0.97)          # n . l with the f16 normals
    within(out['probe00'], sub, 'lvis_map', 3e-2, 0.97)          # DFSS penumbra values: sdf noise x sharp / (2 t)
    assert float(err(out['probe00'].lvis_map, sub['lvis_map']).mean()) < 4e-3


def test_reshade_is_linear_in_the_probe(relight):
    """size-independent property: shade (linear radiance sum) is linear in the probe; batched == one by one"""
    cfg, net, dev, body, eng = relight
    g = torch.Generator().manual_seed(3)
    P = 5000
    ro = torch.randn(P, 3, generator=g).to(dev) + torch.tensor([0.0, 0.0, -2.0], device=dev)
    surf = (torch.rand(P, 3, generator=g) - 0.5).to(dev)
    nrm = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    alb, rgh = torch.rand(P, 3, generator=g).to(dev), (torch.rand(P, generator=g) * 0.9 + 0.09).to(dev)
    lvis, ldot = torch.rand(P, 512, generator=g).to(dev), (torch.rand(P, 512, generator=g) * 2 - 1).to(dev)
    p1, p2 = torch.rand(16, 32, 3, generator=g).to(dev), torch.rand(16, 32, 3, generator=g).to(dev) * 3
    probes = torch.stack([p1, p2, p1 + 2 * p2] + [p1 * k for k in range(2, 9)])         # 10 probes -> two launches
    rgb, shade, spec = eng.reshade(ro, surf, nrm, alb, rgh, lvis, ldot, probes)
    assert rgb.shape == (10, P, 3)
    assert float((shade[2] - (shade[0] + 2 * shade[1])).abs().max()) < 2e-4 * float(shade[2].abs().max())
    assert float((spec[5] - 4 * spec[0]).abs().max()) <= 1e-5 * float(spec[5].abs().max()) + 1e-7
    r1, s1, _ = eng.reshade(ro, surf, nrm, alb, rgh, lvis, ldot, probes[1:2])
    assert float((r1[0] - rgb[1]).abs().max()) == 0.0 and float((s1[0] - shade[1]).abs().max()) == 0.0


def test_full_size_properties():
    """BASELINE size (512x512 relight): chunking / sharding invariance, determinism, bounds"""
    from relightableavatar_amd import shard
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('relight')
    rend = make_renderer(cfg, net)
    H = 512
    base = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True), dev)
    wb0 = base.wbounds.clone()
    out = rend.render(base)
    rgb, acc = out.rgb_map.clone(), out.acc_map.clone()
    assert torch.isfinite(rgb).all() and float(rgb.min()) >= 0 and float(rgb.max()) <= 1.0 + 1e-6
    assert float(acc.min()) >= 0 and float(acc.max()) <= 1.0
    hit = acc > 0
    assert 0.2 < float(hit.float().mean()) < 0.9
    assert float(rgb[~hit].abs().max()) == 0.0                                 # zeros outside the silhouette
    n = out.norm_map[hit] / acc[hit][:, None]
    assert float((n.norm(dim=-1) - 1).abs().max()) < 1e-3                      # premultiplied unit normals
    # determinism: same frame again -> bit identical
    base.wbounds.copy_(wb0)
    out2 = rend.render(base)
    assert float((out2.rgb_map - rgb).abs().max()) == 0.0
    # rays are independent: rendering the two tile-interleaved shards of a 2-rank job and merging gives the same image
    parts = []
    for r in range(2):
        base.wbounds.copy_(wb0)
        parts.append(rend.render(shard.shard_batch(base, r, 2)).rgb_map[0])
    P = rgb.shape[1]
    merged = torch.zeros_like(rgb[0])
    for r in range(2):
        merged[shard.shard_indices(P, r, 2, base, merged.device)] = parts[r]
    assert float((merged - rgb[0]).abs().max()) == 0.0
    c = net.engine().counters()
    assert c.n_fine_sdf > 100 * c.n_hit_pixels                                 # ~1000 fine queries per hit pixel


def test_full_frame_shadow_tier_is_harmless():
    """The tier choice at FULL size, on every pixel: BASELINE's 512 x 512 frame with the shipped tiers (surface trace and the shadow rays
    towards the key lights compensated, the other 5 M shadow rays on plain f16 operands: cfg.trace_precision 1) against the same frame with
    EVERY distance query compensated
    (trace_precision 2: fp32-accurate shadows, 3 x their MFMA work).  The surface trace is the same arithmetic in both, so the hit masks
    and surface points must be identical, and what plain f16 shadows cost must stay far inside the contract."""
    from relightableavatar_amd.renderer import make_renderer
    outs = []
    for tp in (1, 2):
        cfg, net, dev = build('relight', trace_precision=tp)
        out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(512, 512, seed=0, posed=True), dev))
        outs.append({k: out[k].clone() for k in ('rgb_map', 'acc_map', 'surf_map', 'shade_map')})
        c = net.engine().counters()
        assert (c.n_fine_sdf_comp == c.n_fine_sdf) == (tp == 2)
    a, b = outs
    assert torch.equal(a['acc_map'], b['acc_map']) and torch.equal(a['surf_map'], b['surf_map'])
    e = (a['rgb_map'] - b['rgb_map']).abs()
    p = float(-10 * torch.log10((e ** 2).mean()))
    hit = a['acc_map'][0] > 0
    ph = float(-10 * torch.log10((e[0][hit] ** 2).mean()))
    pe = e[0].amax(-1)
    n2, n3 = int((pe > 1e-2).sum()), int((pe > 5e-3).sum())
    print(f'512 x 512, shadows plain f16 vs compensated: rgb PSNR {p:.1f} dB over all {e.shape[1]} in-box rays, {ph:.1f} dB over the {int(hit.sum())} hit pixels, '
          f'max |diff| {float(e.max()):.2e}, pixels over 1e-2: {n2}, over 5e-3: {n3}')
    # Round 6, with the key-light tier (cfg.key_light_share: the learned map's lobes make 5 of its 512 lights key lights): 64.5 dB over the
    # 19 929 hit pixels, max 9.3e-3, NO pixel over 1e-2 — SURVEY.md:409's max half holds on every pixel of BASELINE's frame against the
    # all-compensated frame.  Round 5 (every shadow ray plain f16): 64.3 dB, max 1.3e-2, 2 pixels over 1e-2 (a DFSS penumbra value is
    # d * sharp / (2 t): near the surface, t ~ 5 cm and sharp <= 29 amplify the 6e-5 distance error of plain f16 operands ~300 x per light).
    assert ph >= 60.0 and float(e.max()) <= 1e-2 and n2 == 0 and n3 <= 0.005 * int(hit.sum())


def test_full_size_sample_meets_the_contract():
    """BASELINE.json's frame (512 x 512 full relight) at FULL size, not only through properties: every ~40th in-box ray of the frame
    rendered by the HIP path and by the oracle (rays are independent units) on the body where the reference's own trace converges
    (skin_noise 0) is held to SURVEY.md:409's contract for the 16-bit path (rgb PSNR >= 50 dB, max |err| <= 1e-2).
    bench.py reports the same comparison for the benchmarked body in its `psnr_vs_oracle` object."""
    from oracle import ra_oracle as O
    from relightableavatar_amd.renderer import make_renderer
    torch.set_num_threads(16)
    cfg, net, dev = build('relight')
    mk = lambda: synthetic.sample_rays(synthetic.make_batch(512, 512, seed=0, posed=True, skin_noise=0.0), 1024)[0]
    out = make_renderer(cfg, net).render(synthetic.to_device(mk(), dev))
    ref = O.render_sphere_tracing(O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg), cfg), mk())
    n = ref.rgb_map.shape[1]
    hit = ref.acc_map > 0
    assert n >= 990 and 0.3 < float(hit.float().mean()) < 0.9
    assert float(((out.acc_map.cpu() > 0) == hit).float().mean()) > 0.998
    # every sampled ray that fp32 pins (1022 of 1028) is held to the contract; round 3 (plain f16 operands in the surface trace) could only
    # assert it on the 99 % best rays: 50.1 dB, max 9.7e-2, 4 rays over 1e-2
    assert_contract(out.rgb_map, ref.rgb_map, 'full_size_sample', f'512 x 512 relight, {n} sampled rays ({int(hit.sum())} hit), skin_noise 0')


def test_sharded_ground_pass_matches_the_whole_frame():
    """the README command's frame (relight + ground-plane pass) rendered as 2 and 3 shards: each rank's full-frame tiles with its human
    rays blended in locally, merged through the plan's index vectors — bit-identical to the unsharded frame, also across the ground
    pass's chunk boundaries (the box grows per chunk of the WHOLE frame)"""
    from relightableavatar_amd import shard
    from relightableavatar_amd.renderer import make_renderer
    kw = dict(vis_ground_shading=True, ground_normal=[0.0, -1.0, 0.0], ground_origin=[0.0, 0.45, 0.0], render_chunk_size=30000)
    cfg, net, dev = build('relight', **kw)
    rend = make_renderer(cfg, net)
    H = 256
    mk = lambda: synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True), dev)
    whole = rend.render(mk())
    assert whole.rgb_map.shape == (1, H * H, 3)
    for world in (2, 3):
        base = mk()
        P = base.ray_o.shape[1]
        pl = shard.make_plan(P, world, base, dev, mask=base.mask_at_box.cpu(), ground=True, render_chunk_size=cfg.render_chunk_size)
        rgb, acc = torch.zeros_like(whole.rgb_map[0]), torch.zeros_like(whole.acc_map[0])
        for r in range(world):
            out = rend.render(shard.shard_batch(base, r, world, cfg.render_chunk_size, pl, ground=True))
            rgb[pl.ground.idx[r]], acc[pl.ground.idx[r]] = out.rgb_map[0], out.acc_map[0]
        assert torch.equal(rgb, whole.rgb_map[0]) and torch.equal(acc, whole.acc_map[0]), world
        assert not bool(base.mask_at_box.all())               # the shards worked on their own copies of the mask


def test_edge_cases(relight):
    cfg, net, dev, body, eng = relight
    from relightableavatar_amd.renderer import make_renderer
    rend = make_renderer(cfg, net)
    # empty ray set (chunkify's zero-length case)
    b = synthetic.to_device(synthetic.make_batch(64, 64, seed=0, posed=True), dev)
    for k in ('ray_o', 'ray_d', 'near', 'far'):
        b[k] = b[k][:, :0]
    out = rend.render(b)
    assert out.rgb_map.shape == (1, 0, 3)
    # rays that all miss the body
    b = synthetic.to_device(synthetic.make_batch(64, 64, seed=0, posed=True), dev)
    b.ray_d = torch.nn.functional.normalize(b.ray_d * torch.tensor([1.0, 1.0, -1.0], device=dev), dim=-1)
    out = rend.render(b)
    assert float(out.acc_map.abs().max()) == 0.0 and float(out.rgb_map.abs().max()) == 0.0
    # ragged size (not a multiple of any tile) through the operator API
    x = (torch.rand(1, 1237, 3, device=dev) - 0.5)
    s = net.inference_world_distance_field(x, body, smooth_transition=True)
    assert s.shape == (1, 1237, 1) and torch.isfinite(s).all()


def test_multi_chunk_matches_oracle():
    """chunkify + the per-chunk in-place bbox growth (quirk 1): 3 chunks on the GPU vs the oracle with the same chunking"""
    from oracle import ra_oracle as O
    from relightableavatar_amd.renderer import make_renderer
    torch.set_num_threads(16)
    cfg, net, dev = build('relight', render_chunk_size=24)
    b_cpu = synthetic.make_batch(128, 128, seed=0, posed=True, crop=8)
    assert b_cpu.ray_o.shape[1] == 64
    out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(128, 128, seed=0, posed=True, crop=8), dev))
    ref = O.render_sphere_tracing(O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg), cfg), b_cpu)
    # 64 rays / ceil(64/24)=3 chunks -> the box grew three times
    np.testing.assert_allclose(b_cpu.wbounds.numpy()[0, 1] - synthetic.make_body(0).wbounds.numpy()[0, 1], 0.75, atol=1e-6)
    assert bool(((out.acc_map.cpu() > 0) == (ref.acc_map > 0)).all())
    within(out, ref, 'albedo_map', 5e-4, 0.98)
    within(out, ref, 'shade_map', 2e-2, 0.95)
    within(out, ref, 'rgb_map', 1e-2, 0.95)
    assert_contract(out.rgb_map, ref.rgb_map, 'multi_chunk')


def test_anisdf_sphere_tracing_vs_oracle_other_pose():
    """a second body pose / seed than the golden fixtures (identity pose: A = big_A = I, R = I)"""
    from oracle import ra_oracle as O
    from relightableavatar_amd.renderer import make_renderer
    torch.set_num_threads(16)
    cfg, net, dev = build('sphere_tracing')
    b_cpu = synthetic.make_batch(96, 96, seed=3, posed=False, crop=16)
    out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(96, 96, seed=3, posed=False, crop=16), dev))
    ref = O.render_sphere_tracing(O.OracleNet(synthetic.make_state_dict(0, relight=False, cfg=cfg), cfg), b_cpu)
    assert float(((out.acc_map.cpu() > 0) == (ref.acc_map > 0)).float().mean()) > 0.99
    within(out, ref, 'rgb_map', 5e-3, 0.97)
    within(out, ref, 'norm_map', 2e-2, 0.95)
    assert_contract(out.rgb_map, ref.rgb_map, 'other_pose')
    pn = psnr(out.norm_map, ref.norm_map)
    print(f'other_pose: normals {pn:.1f} dB')
    assert pn >= 60.0            # measured 70.1 dB (round 3, plain f16 surface trace: 58.1; the emulated-f16 floor of the normals alone: 59.3)


def test_errors_are_python_exceptions():
    from relightableavatar_amd import _lib
    from relightableavatar_amd.engine import Engine
    dev = _dev()
    eng = Engine(make_cfg('relight'), dev, relight=True)
    with pytest.raises(_lib.RaError, match='weights not finalized'):
        eng.hdq_sdf(torch.zeros(4, 3, device=dev), 0.1, True)
    with pytest.raises(_lib.RaError, match='missing'):
        eng.load_state_dict({'residual_deformation_network.mlp.linears.0.weight': torch.zeros(256, 219)})
    K, R, Tc = synthetic.make_camera(16, 16)
    with pytest.raises(_lib.RaError, match='bad image size'):
        eng.gen_rays(0, 16, K, R, Tc, torch.tensor([[-1., -1., -1.], [1., 1., 1.]]))
    # the ground pass needs a finalised relight network, and refuses chunks whose (pixel x light) count overflows an int
    with pytest.raises(_lib.RaError, match='weights not finalized'):
        z3 = torch.zeros(4, 3, device=dev)
        eng.render_ground_chunk(z3, z3, torch.ones(4, device=dev), [-1, -1, -1, 1, 1, 1], torch.ones(16, 32, 3, device=dev),
                                eng.ground_params(), {})


def test_ground_pass_edge_cases(relight):
    """ground chunks: nothing to trace (acc = 0 everywhere) gives zeros; rays looking away from the plane get no ground"""
    cfg, net, dev, body, eng = relight
    P = 300
    g = torch.Generator().manual_seed(3)
    ro = torch.tensor([0., 0., -2.], device=dev).expand(P, 3).contiguous()
    rd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    probe = net.global_env_map.to(dev, torch.float32).contiguous()
    gp = eng.ground_params()
    gp.normal[0], gp.normal[1], gp.normal[2] = 0.0, -1.0, 0.0
    gp.origin[0], gp.origin[1], gp.origin[2] = 0.0, 0.45, 0.0
    bbox = [-0.7, -0.7, -0.7, 0.7, 0.7, 0.7]
    outs = {k: torch.full((P, 3), 7.0, device=dev) for k in ('rgb', 'surf', 'albedo', 'shade', 'spec')}
    outs['depth'] = torch.full((P,), 7.0, device=dev)
    eng.render_ground_chunk(ro, rd, torch.zeros(P, device=dev), bbox, probe, gp, outs)
    assert float(outs['rgb'].abs().max()) == 0.0 and float(outs['shade'].abs().max()) == 0.0        # no pixel to trace
    eng.render_ground_chunk(ro, rd, torch.ones(P, device=dev), bbox, probe, gp, outs)
    assert torch.isfinite(outs['rgb']).all() and float(outs['rgb'].min()) >= 0.0
    up = rd[:, 1] < 0                                    # looking away from the plane (y = 0.45, normal -y): t <= 0
    assert bool((outs['depth'][up] <= 0).all())
    # beyond env_r the ground fades into the unshadowed light sum: identical shade for all far / upward pixels
    far = outs['shade'][up]
    assert float((far - far[0]).abs().max()) < 1e-5


@pytest.mark.gpu
def test_rccl_path_world1_under_torchrun():
    """the launcher contract of bench.py (torch.distributed.run, nccl == RCCL) at the world size a 1-GPU box allows:
    process-group init on the device, the frame all_gather, barrier + MAX all_reduce, and the JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1']
    r = subprocess.run(base + ['--master-port', '29631', os.path.join(root, 'tools', 'nccl_world1.py')],
                       capture_output=True, text=True, env=env, timeout=600)
    assert 'NCCL_WORLD1_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run(base + ['--master-port', '29632', os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1',
                               '--size', '128', '--no-cpu-baseline'], capture_output=True, text=True, env=env, timeout=600)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['roofline']['achieved'] > 0, r.stdout[-2000:] + r.stderr[-2000:]
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in line, k
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(line['roofline']) and 'workload' in line['config']


RANK_PROGRAM_CASES = {
    # BASELINE config 4: 512 x 512 full relight, (rgb, acc) gathered
    'config4': (['--mode', 'relight', '--size', '512'], (2, 4)),
    # config 5: 1024 x 1024, 8 probes re-shaded per frame, the 24-channel payload
    'config5': (['--mode', 'novel_light', '--size', '1024', '--probes', '8'], (4,)),
    # the README's relight command (readme.md:64): novel lights + the ground-plane pass, full-frame maps blended per rank
    'readme': (['--mode', 'novel_light', '--ground', '--size', '512', '--probes', '2'], (2,)),
}


@pytest.mark.parametrize('case', list(RANK_PROGRAM_CASES))
def test_rank_program_as_processes_sharing_the_gpu(case, tmp_path):
    """The REAL rank program with N > 1 (the reference's rendezvous convention train.py:116-122): bench.py started by its own
    launcher (torch.distributed.run, env:// on 127.0.0.1) as N fresh rank processes that all render on GPU 0 — HIP engine, per-process
    context creation, shard plan in C, pinned staging, launch-variant hints, gate, three frames in flight — with the frame all_gather
    staged through pinned host memory over gloo (RCCL refuses two ranks on one device; RCCL itself runs at world 1 in
    test_rccl_path_world1_under_torchrun).  The frame rank 0 assembled must equal the single-process frame BIT FOR BIT, and the JSON line
    must show every rank took part with its share of the work.  Everything of an 8-GPU run except the xGMI transport."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags, worlds = RANK_PROGRAM_CASES[case]
    common = ['--steps', '3', '--warmup', '1', '--soak', '0', '--no-cpu-baseline', '--no-sequential', '--frames-in-flight', '3']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')

    def run(world):
        out = str(tmp_path / f'{case}_w{world}.npy')
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--dump-frame', out] + flags + common
        if world > 1:
            cmd += ['--backend', 'gloo', '--share-gpu']
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert r.returncode == 0 and lines, (world, r.stdout[-3000:], r.stderr[-3000:])
        return json.loads(lines[-1]), np.load(out)
    line1, whole = run(1)
    assert np.isfinite(whole).all() and float(np.abs(whole).max()) > 0
    for world in worlds:
        line, frame = run(world)
        assert line['n_gpus'] == world and line['ranks_seen'] == world, line
        hp = line['per_rank']['hit_pixels_per_frame']
        assert len(hp) == world and min(hp) > 0 and abs(sum(hp) - line1['config']['hit_pixels_per_frame']) <= 1, (hp, line1['config']['hit_pixels_per_frame'])
        assert max(hp) <= 1.25 * (sum(hp) / world)           # the 8 x 8 tile deal balances the expensive pixels
        assert line['gather']['channels'] == whole.shape[-1] or case == 'config4'
        assert frame.shape == whole.shape and np.array_equal(frame, whole), (case, world, float(np.abs(frame - whole).max()))
        print(f'{case}: {world} rank processes on one GPU == the single-process frame bit for bit ({whole.shape}); per rank: {line["per_rank"]}')


def test_ray_generation_on_device(golden, relight):
    """N2: ra_gen_rays vs the reference's get_rays_within_bounds outputs (golden rays.npz): same in-box
    pixels in the same order, directions within 1 ulp-ish (2e-7), near/far within 2e-6; then full-size properties."""
    _, _, dev, body, eng = relight
    g = golden('rays.npz')
    for tag in ('a', 'b'):
        H, W = int(g[f'{tag}_H']), int(g[f'{tag}_W'])
        o = eng.gen_rays(H, W, g[f'{tag}_K'], g[f'{tag}_R'], g[f'{tag}_T'], g['bounds'])
        ref_mask = T(g[f'{tag}_mask'])
        assert int((o.mask_at_box.cpu() != ref_mask).sum()) <= 2                  # pixels grazing a box edge may flip
        if bool((o.mask_at_box.cpu() == ref_mask).all()):
            assert float(err(o.ray_d, g[f'{tag}_ray_d']).max()) < 2e-7 and float(err(o.ray_o, g[f'{tag}_ray_o']).max()) < 1e-7
            assert float(err(o.near, g[f'{tag}_near']).max()) < 2e-6 and float(err(o.far, g[f'{tag}_far']).max()) < 2e-6
    # 512 x 512: matches the host-side set-up the benchmark batches are built with, row-major order, unit directions
    from relightableavatar_amd.data_utils import get_rays_within_bounds
    K, R, Tc = synthetic.make_camera(512, 512)
    ro, rd, near, far, mask = get_rays_within_bounds(512, 512, K, R, Tc, body.wbounds[0], eng)
    b = synthetic.make_batch(512, 512, seed=0, posed=True)
    assert int((mask.cpu().reshape(1, -1) != b.mask_at_box).sum()) <= 4
    if ro.shape[0] == b.ray_o.shape[1]:
        assert float(err(rd, b.ray_d[0]).max()) < 3e-7 and float(err(near, b.near[0]).max()) < 5e-6 and float(err(far, b.far[0]).max()) < 5e-6
    assert float((rd.norm(dim=-1) - 1).abs().max()) < 1e-6 and bool((near < far).all())
    # degenerate: a box behind the camera yields no rays
    e = eng.gen_rays(16, 16, K, R, Tc, torch.tensor([[10., 10., -9.], [11., 11., -8.]]))
    assert e.ray_o.shape[0] == 0 and not bool(e.mask_at_box.any())


def test_frame_ground(golden):
    """N1: relit frame with the ground-plane pass vs the reference's frame (golden) and vs the oracle on a
    larger frame; the ground layer is fp32 except for the shadow trace's fine distance queries (dist_th 5 mm)."""
    from relightableavatar_amd.renderer import make_renderer
    from oracle import ra_oracle as O
    ref = golden('frame_ground.npz')
    kw = dict(vis_ground_shading=True, ground_normal=[float(v) for v in ref['ground_normal']],
              ground_origin=[float(v) for v in ref['ground_origin']], render_chunk_size=int(ref['render_chunk_size']))
    cfg, net, dev = build('relight', **kw)
    H, crop = int(ref['H']), int(ref['crop'])
    batch = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop), dev)
    rend = make_renderer(cfg, net)
    m = batch.mask_at_box.reshape(1, -1).cpu()
    rend.ground_inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]   # the scatter order of the CPU reference run
    out = rend.render(batch)
    np.testing.assert_allclose(batch.wbounds.cpu().numpy(), ref['wbounds_after'], atol=1e-6)
    assert out.rgb_map.shape == (1, H * H, 3) and bool(batch.mask_at_box.all())
    e = err(out.rgb_map, ref['rgb_map'])
    # the contract on every pixel of the blended frame (measured 78.2 dB / 1.5e-3; round 3, plain f16 surface trace: 63.1 dB with four
    # silhouette pixels, where the human layer's alpha decides between the two layers, over 5e-3)
    assert_contract(out.rgb_map, ref['rgb_map'], 'frame_ground.npz', 'frame_ground', bad=torch.zeros(H * H, dtype=torch.bool))
    assert float((e < 5e-3).float().mean()) > 0.999
    assert float(err(out.albedo_map, ref['albedo_map']).max()) < 1e-2            # ground: fp32 probe lookups; human pixels: f16 heads
    near = T(ref['surf_map'])[0].abs().amax(-1) < 1e3        # rays parallel to the plane: t = x / (0 + eps * |random edge|^2) in the reference
    assert float((err(out.surf_map, ref['surf_map'])[0][near] < 1e-3).float().mean()) > 0.99
    assert float((err(out.shade_map, ref['shade_map']) < 5e-3).float().mean()) > 0.99
    assert float(err(out.acc_map, ref['acc_map']).max()) < 3e-2          # acc = 1 - min_i 500 d_i / t_i: a 1e-4 distance error is 2.5e-2 of alpha at t = 2
    # the shadow on the ground exists: some ground pixels near the body are darker than the unshadowed ground
    acc_h = out.acc_map[0]
    g = out.shade_map[0][acc_h == 0].sum(-1)
    assert float(g.min()) < 0.9 * float(g.median())


def test_query_skip_is_exact():
    """rays that did not move since their last query (clamped at far/near) and shadow rays whose visibility already reached 0
    are not re-queried: the frame (relight + ground pass) must be BIT-identical to the one rendered with cfg.query_skip = False
    (the reference's schedule, sphere_tracing_renderer.py:144-205) — and cheaper"""
    from relightableavatar_amd.renderer import make_renderer
    outs, fine = [], []
    for skip in (True, False):
        cfg, net, dev = build('relight', vis_ground_shading=True, ground_normal=[0.0, -1.0, 0.0], ground_origin=[0.0, 0.45, 0.0], query_skip=skip)
        out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(256, 256, seed=0, posed=True), dev))
        outs.append({k: out[k].cpu() for k in ('rgb_map', 'acc_map', 'shade_map', 'surf_map')})
        fine.append(net.engine().counters().n_fine_sdf)
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert fine[0] < fine[1], fine


def test_merged_ground_chunks_are_exact():
    """cfg.ground_chunk_rays: consecutive render chunks of the ground-plane pass share one launch sequence, every pixel clipped against
    the box the reference's in-place growth had reached at ITS chunk (sphere_tracing_renderer.py:1054-1056) — bit-identical to chunking
    exactly as the reference does (ground_chunk_rays = 0), here with 8 ground chunks of 2048 pixels and therefore 8 different boxes"""
    from relightableavatar_amd.renderer import make_renderer
    outs = []
    for merged in (262144, 5000, 0):              # all 8 chunks in one launch; groups of two; the reference's chunking
        cfg, net, dev = build('relight', vis_ground_shading=True, ground_normal=[0.0, -1.0, 0.0], ground_origin=[0.0, 0.45, 0.0],
                              render_chunk_size=2048, ground_chunk_rays=merged)
        out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(128, 128, seed=0, posed=True), dev))
        outs.append({k: out[k].cpu() for k in ('rgb_map', 'acc_map', 'shade_map', 'surf_map')})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[2][k]), k
        assert torch.equal(outs[1][k], outs[2][k]), k
    assert float(outs[0]['shade_map'].abs().sum()) > 0


def test_merged_sphere_chunks_are_exact():
    """cfg.sphere_chunk_rays: consecutive render chunks of the sphere-tracing renderer share ONE launch sequence (one 16-iteration surface
    loop for the frame), the shadow rays of every ray clipped against the box the reference's in-place growth had reached at ITS chunk
    (sphere_tracing_renderer.py:1020-1022) — bit-identical to chunking exactly as the reference does (sphere_chunk_rays = 0); here four
    chunks of 700 rays and therefore four boxes, whole frame / groups of two / one by one, and one shard of two"""
    from relightableavatar_amd import shard
    from relightableavatar_amd.renderer import make_renderer
    outs, parts = [], []
    for merged in (262144, 1500, 0):
        cfg, net, dev = build('relight', render_chunk_size=700, sphere_chunk_rays=merged, vis_specular_map=True)
        base = synthetic.to_device(synthetic.make_batch(128, 128, seed=0, posed=True), dev)
        assert 2100 < base.ray_o.shape[1] <= 2800
        rend = make_renderer(cfg, net)
        out = rend.render(base)
        outs.append({k: out[k].cpu() for k in ('rgb_map', 'acc_map', 'shade_map', 'spec_map', 'surf_map', 'norm_map', 'albedo_map')})
        b2 = synthetic.to_device(synthetic.make_batch(128, 128, seed=0, posed=True), dev)
        parts.append(rend.render(shard.shard_batch(b2, 1, 2, cfg.render_chunk_size)).rgb_map.cpu())      # a shard walks the frame's chunks, some nearly empty
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[2][k]), k
        assert torch.equal(outs[1][k], outs[2][k]), k
    assert torch.equal(parts[0], parts[2]) and torch.equal(parts[1], parts[2])
    assert float(outs[0]['shade_map'].abs().sum()) > 0


def test_full_size_properties_frame_filling_subject():
    """The frame-filling case of BASELINE's frame (bench.py --coverage 0.35: camera at 0.96 m, every one of the 262 144 pixels inside the
    box, ~40 % hit pixels): four of the reference's 65 536-ray render chunks in ONE launch sequence (cfg.sphere_chunk_rays).  Size-independent
    properties: finite maps in range, misses black, the expected coverage, the 2-shard merge bit-identical to the whole frame (a shard walks
    the frame's four chunks and their grown boxes), and a strided sample of the rays rendered on its own equals the same rays of the
    whole frame bit for bit (rays are independent units)."""
    import math
    from relightableavatar_amd import shard
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('relight')
    rend = make_renderer(cfg, net)
    cam = 0.8 * 0.4 / math.sqrt(0.35 / math.pi)
    mk = lambda: synthetic.make_batch(512, 512, seed=0, posed=True, cam_dist=cam)
    base = synthetic.to_device(mk(), dev)
    P = base.ray_o.shape[1]
    assert P == 512 * 512 and P == 4 * cfg.render_chunk_size
    wb0 = base.wbounds.clone()
    out = rend.render(base)
    rgb, acc = out.rgb_map.clone(), out.acc_map.clone()
    hit = acc[0] > 0
    assert torch.isfinite(rgb).all() and float(rgb.min()) >= 0 and float(rgb.max()) <= 1.0 + 1e-6
    assert 0.30 < float(hit.float().mean()) < 0.50
    assert float(rgb[0][~hit].abs().max()) == 0.0
    c = net.engine().counters()
    assert c.n_hit_pixels == int(hit.sum()) and c.n_shadow_rays > 100 * c.n_hit_pixels
    merged = torch.zeros_like(rgb[0])
    for r in range(2):
        base.wbounds.copy_(wb0)
        o = rend.render(shard.shard_batch(base, r, 2, cfg.render_chunk_size))
        merged[shard.shard_indices(P, r, 2, base, merged.device)] = o.rgb_map[0]
    assert float((merged - rgb[0]).abs().max()) == 0.0
    # every 509th ray on its own: one chunk, its shadow rays clipped against the FIRST chunk's box, so only rays of the first chunk compare
    sub, _, stride = synthetic.sample_rays(mk(), 512)
    o = rend.render(synthetic.to_device(sub, dev))
    idx = torch.arange(0, P, stride, device=rgb.device)[:o.rgb_map.shape[1]]
    first = idx < cfg.render_chunk_size
    assert int(first.sum()) > 100 and float((o.rgb_map[0][first] - rgb[0][idx[first]]).abs().max()) == 0.0


def test_launch_variant_hints_change_speed_only():
    """Launch-variant hints (csrc/ra_ctx.hpp HintSlot): a render call's fine counts, copied to pinned memory behind an event, size the
    fused MLP kernel's workgroup width for the same call of a later frame.  The first frame of a context runs without hints (the ground
    pass's distance launches take the 8-wave kernel: their bound is pixels x 512 lights), later ones with (the narrow kernel): the frames
    must be bit-identical, and the variant must really have changed."""
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('relight', vis_ground_shading=True)
    rend = make_renderer(cfg, net)
    eng = net.engine()
    eng.enable_timing(True)
    frames, wide, narrow = [], [], []
    for k in range(4):
        batch = synthetic.to_device(synthetic.make_batch(256, 256, seed=0, posed=True), dev)
        eng.reset_counters()
        out = rend.render(batch)
        torch.cuda.synchronize()
        frames.append({kk: out[kk].clone() for kk in ('rgb_map', 'acc_map', 'shade_map')})
        wide.append(eng.kernel_time(2)[1])
        narrow.append(eng.kernel_time(3)[1])
    eng.enable_timing(False)
    print(f'8-wave / narrow distance launches per frame: {list(zip(wide, narrow))}')
    for f in frames[1:]:
        for kk in f:
            assert torch.equal(f[kk], frames[0][kk]), kk
    assert wide[0] > wide[-1] and narrow[-1] > narrow[0]            # the ground pass's launches moved to the narrow kernel
    assert wide[0] + narrow[0] == wide[-1] + narrow[-1]            # ... and nothing else changed


def test_stale_hints_cost_no_rays_and_no_cliff():
    """Launch-variant hints come from an EARLIER frame.  After a camera cut (here: three 64 x 64 frames, then a 512 x 512 one through the
    same context) the counts are two orders of magnitude too small: the frame must be bit-identical to the one a context with matching
    hints renders, and — since round 6 the hint picks the workgroup WIDTH only, the grid is made for at least an eighth of the host-side
    bound (csrc/ra_common.hpp mlp_grid) — it costs the narrow variants' lower rate, not a handful of workgroups on millions of points."""
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('relight')
    rend = make_renderer(cfg, net)
    big = lambda: synthetic.to_device(synthetic.make_batch(512, 512, seed=0, posed=True), dev)
    small = lambda: synthetic.to_device(synthetic.make_batch(64, 64, seed=0, posed=True), dev)

    def timed(batch):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = rend.render(batch)
        b.record()
        torch.cuda.synchronize()
        return {k: out[k].clone() for k in ('rgb_map', 'acc_map', 'shade_map')}, a.elapsed_time(b)
    timed(big())                                   # allocations, first launches
    timed(big())
    want, t_good = timed(big())                    # hints of a frame like itself
    for _ in range(3):
        timed(small())                             # the same call slots now hold a 64 x 64 frame's counts
    got, t_stale = timed(big())
    for k in want:
        assert torch.equal(got[k], want[k]), k
    print(f'512 x 512 frame with matching hints {t_good:.1f} ms, with the hints of a 64 x 64 frame {t_stale:.1f} ms')
    assert t_stale < 4.0 * t_good, (t_good, t_stale)


def test_frames_in_flight_are_bit_identical():
    """relightableavatar_amd/pipeline.py: frames rendered two at a time on two HIP streams (contexts sharing a gate that serialises
    their light-visibility stages) equal the frames rendered one after the other, bit for bit — alternating poses, so that a frame
    picking up its neighbour's body state, scratch or counters would show"""
    from relightableavatar_amd.pipeline import FramePipeline
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('relight')
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    keys = ('rgb_map', 'acc_map', 'shade_map', 'surf_map', 'norm_map')
    batches = [synthetic.to_device(synthetic.make_batch(192, 192, seed=k % 2, posed=True), dev) for k in range(6)]
    serial = make_renderer(cfg, net)
    want = []
    for b in batches:
        out = serial.render(b)
        want.append({k: out[k].clone() for k in keys})
    pipe = FramePipeline(cfg, sd, dev, depth=2)
    # fresh batches (render() grows wbounds in place), NOT kept by the caller: the pipeline must keep a frame's inputs alive and off the
    # allocator's free list while the replica's stream still reads them (a dropped batch once came back as the next frame's memory)
    pending = [pipe.submit(synthetic.to_device(synthetic.make_batch(192, 192, seed=k % 2, posed=True), dev)) for k in range(6)]
    for k, p in enumerate(pending):
        out = p.result()
        for key in keys:
            assert torch.equal(out[key], want[k][key]), (k, key)
    assert not torch.equal(want[0]['rgb_map'], want[1]['rgb_map'])         # the two poses do differ
    # the gate is what the contexts share: detaching works, and a pipeline of depth 1 is plain sequential rendering
    for e in pipe.engines():
        e.set_gate(None)
    one = FramePipeline(cfg, sd, dev, depth=1)
    out = one.submit(synthetic.to_device(synthetic.make_batch(192, 192, seed=1, posed=True), dev)).result()
    assert torch.equal(out['rgb_map'], want[1]['rgb_map'])


# kernels of OTHER libraries that may run between FramePipeline.submit() and Pending.result(), by name: they are not built with this library's
# flags (-fno-slp-vectorize: DESIGN.md section 8, the packed-fp32 hazard beside another wave's MFMAs), so each one is listed with the reason
# it is harmless there.  A new name fails test_only_known_kernels_run_between_submit_and_result: look at it before adding it.
FOREIGN_KERNELS_IN_FLIGHT = {
    '__amd_rocclr_copyBuffer': 'the runtime\'s blit kernel (hipMemcpyAsync device-to-device / pinned): integer moves',
    '__amd_rocclr_fillBufferAligned': 'the runtime\'s memset kernel: integer stores',
    'at::native::CatArrayBatchedCopy_contig': 'torch.stack of the frame\'s probes (novel-light renderer): moves 4-byte words as an opaque type, no floating-point instruction',
}


def library_kernel_stems():
    """the __global__ kernels of csrc/ (their names without template arguments): compiled with the library's flags and scanned by
    tools/check_packed_fp32.py at link time; rocprim's / hipcub's sort kernels are instantiated inside csrc/ra_trace.hip and are part of
    the same objects"""
    import glob
    import re
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stems = set()
    for f in glob.glob(os.path.join(here, 'relightableavatar_amd', 'csrc', '*.h*')):
        stems |= set(re.findall(r'__global__\s+(?:__launch_bounds__\([^)]*\)\s*)?void\s+(\w+)\s*\(', open(f).read()))
    return stems


def test_only_known_kernels_run_between_submit_and_result():
    """Frames in flight put other kernels beside this library's MFMA kernels on the same SIMDs.  Round 5 found that compiler-formed packed
    fp32 (`v_pk_* op_sel:[..1..]`) miscomputes there; the library's own objects are built without it and checked at link time, but a kernel
    of ANOTHER library (the host framework's elementwise / index kernels) is not.  This test records every kernel that runs while three
    frames are in flight — whole frames and one rank's share of a sharded frame, relight and the novel-light re-shade — and fails on any name
    that is neither a kernel of csrc/ nor listed in FOREIGN_KERNELS_IN_FLIGHT."""
    from torch.profiler import ProfilerActivity, profile
    from relightableavatar_amd import shard
    from relightableavatar_amd.pipeline import FramePipeline
    dev = _dev()
    stems = library_kernel_stems()
    assert {'shadow_gen_kernel', 'hdq_coarse_kernel', 'mlp_sdf_stream_kernel', 'key_lights_kernel', 'gather_shard_rays_kernel'} <= stems, sorted(stems)[:8]
    seen = {}
    for mode in ('relight', 'novel_light'):
        cfg = make_cfg(mode, mlp_dtype='f16', novel_light_timing=False)
        pipe = FramePipeline(cfg, synthetic.make_state_dict(0, relight=True, cfg=cfg), dev, depth=3)
        mk = lambda k: synthetic.to_device(synthetic.make_batch(160, 160, seed=k % 2, posed=True, n_novel_lights=2 if mode == 'novel_light' else 0), dev)
        batches = [mk(k) for k in range(12)]           # made BEFORE the recorded region: the loader's uploads are not part of a frame
        plans = [shard.make_plan(b.ray_o.shape[1], 4, b, dev, mask=b.mask_at_box.cpu(), render_chunk_size=cfg.render_chunk_size, use_cache=False) for b in batches]

        def frames(first):
            pend = []
            for k in range(first, first + 6):
                b = batches[k]
                if k % 2:      # one rank's share of a 4-rank frame: shard_batch's gather runs on the replica's stream too
                    pend.append(pipe.submit(fn=lambda net, rend, b=b, pl=plans[k]: rend.render(shard.shard_batch(b, 1, 4, cfg.render_chunk_size, pl))))
                else:
                    pend.append(pipe.submit(b))
            for p_ in pend:
                p_.result()
            torch.cuda.synchronize()
        frames(0)                                        # first frames: allocations, first launches
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            frames(6)
        for e in prof.events():
            if str(e.device_type).endswith('CUDA') and e.name and not e.name.startswith(('Memcpy', 'Memset', 'hip')):
                seen[e.name] = seen.get(e.name, 0) + 1
    if not seen:
        pytest.skip('the profiler recorded no device activity on this box')
    import re
    unknown = {}
    for name, n in seen.items():
        base = re.sub(r'^void\s+', '', name).replace('(anonymous namespace)::', '')
        stem = re.split(r'[<(]', base)[0].strip()
        m = re.match(r'_ZN12_GLOBAL__N_1\d+(\w+?_kernel)', base)          # a mangled name (anonymous namespace + template)
        stem = m.group(1) if m else stem
        if stem in stems or stem in FOREIGN_KERNELS_IN_FLIGHT or 'rocprim' in base or 'hipcub' in base:
            continue
        unknown[name[:120]] = n
    print(f'{len(seen)} kernel names between submit() and result(); of another library: '
          f'{sorted(k[:60] for k in seen if any(f in k for f in FOREIGN_KERNELS_IN_FLIGHT))}')
    assert not unknown, f'kernels of another library ran beside frames in flight: {unknown} (tests/test_gpu_parity.py FOREIGN_KERNELS_IN_FLIGHT)'


def test_envmap_rotation_and_probe_inset(golden, relight):
    """N4: ra_shift_envmap / ra_add_light_probe vs the reference's rotate_envmap / add_light_probe outputs"""
    from relightableavatar_amd import relight_utils
    from relightableavatar_amd.base_utils import dotdict
    cfg, net, dev, body, eng = relight
    g = golden('envmap.npz')
    lights = synthetic.make_novel_lights(3, 0)
    nl = dotdict({k: dotdict(probe=v.probe, image=T(g['images'][i])[None]) for i, (k, v) in enumerate(lights.items())})
    repeat = int(g['repeat'])
    for index in (0, 5, 37, 128 + 77, 2 * 128 + 127):
        name, env = relight_utils.rotate_envmap(nl, index, repeat, 32, 48, eng)
        assert name == str(g[f'rot{index}_name'])
        assert float(err(env.probe[0], g[f'rot{index}_probe']).max()) < 2e-5          # HDR values up to 100
        assert float(err(env.image[0], g[f'rot{index}_image']).max()) < 1e-6
    H, W = int(g['H']), int(g['W'])
    batch = dotdict(meta=dotdict(H=torch.tensor([H]), W=torch.tensor([W])), cam_R=T(g['cam_R'])[None])
    c2 = dotdict(env_h=16, env_w=32, probe_size_ratio=0.2)
    out = relight_utils.add_light_probe(T(g['rgb_in'])[None], lights['probe00'].probe, batch, c2, eng)
    assert out.shape == (1, H * W, 3)
    assert float(err(out[0], g['rgb_out']).max()) < 5e-5
    # a full turn brings the probe back
    full = eng.shift_envmap(lights['probe00'].probe, 32.0)
    assert float(err(full, lights['probe00'].probe).max()) < 1e-5


def test_rotating_light_sequence():
    """cfg.vis_rotate_light: every probe is re-shaded at rotate_ratio * env_w headings from ONE traced frame; heading 0 is the
    unrotated probe, and a constant probe is rotation invariant"""
    from relightableavatar_amd.renderer import make_renderer
    from relightableavatar_amd.base_utils import dotdict
    # The frame is traced ONCE for all headings; which of its shadow rays run in compensated arithmetic depends on the probes it will be
    # shaded with (cfg.key_light_share: the key lights of every heading), so "heading 0 == the unrotated probe alone" is exact only
    # without that tier, and holds to the tiers' own difference with it.
    for share, tol in ((0.0, 1e-6), (0.0078, 5e-3)):
        cfg, net, dev = build('novel_light', vis_rotate_light=True, rotate_ratio=1, test_light=[], key_light_share=share)
        batch = synthetic.make_batch(96, 96, seed=0, posed=True, crop=16, n_novel_lights=1)
        batch.novel_lights['flat'] = dotdict(probe=torch.full((1, 16, 32, 3), 0.7))
        out = make_renderer(cfg, net).render(synthetic.to_device(batch, dev))
        names = [k for k in out if k != 'diff']
        assert len(names) == 2 * 32 and 'probe00-0000' in names and 'flat-0031' in names
        cfg2, net2, _ = build('novel_light', test_light=[], key_light_share=share)
        ref = make_renderer(cfg2, net2).render(synthetic.to_device(synthetic.make_batch(96, 96, seed=0, posed=True, crop=16, n_novel_lights=1), dev))
        d = float((out['probe00-0000'].rgb_map - ref['probe00'].rgb_map).abs().max())
        print(f'rotating-light sequence, key_light_share {share}: heading 0 vs the unrotated probe alone: max |diff| {d:.2e}')
        assert d < tol, (share, d)
    assert float((out['flat-0000'].rgb_map - out['flat-0017'].rgb_map).abs().max()) < 1e-5
    assert float((out['probe00-0000'].rgb_map - out['probe00-0016'].rgb_map).abs().max()) > 1e-3      # half a turn changes the picture


def test_streamed_k3_tile_boundaries(relight):
    """the production K3 (weights streamed, 32 points per wave, 64/128/256-point tiles) against the forward kernel of the full query
    (the stage hook: same networks, its own launch geometry), for fine counts around every tile boundary (ragged last tiles, a single
    point, an empty set)"""
    _, _, dev, body, eng = relight
    g = torch.Generator().manual_seed(21)
    vid = torch.randint(0, 6890, (70000,), generator=g)
    wv = (body.pverts[0] @ body.R[0].T + body.Th[0])[vid.to(dev)]
    x_all = wv + 0.01 * torch.nn.functional.normalize(torch.randn(70000, 3, generator=g), dim=-1).to(dev)     # all within dist_th
    for n in (1, 31, 32, 33, 63, 64, 65, 127, 129, 255, 256, 257, 511, 513, 16383, 16385, 32769, 65537):
        x = x_all[:n].contiguous()
        o = eng.debug_hdq(x, 0.125)
        assert o.fine_count == n
        sdf3 = eng.hdq_sdf(x, 0.125, False)                      # production path: coarse level + streamed K3, no blend
        _, sdf1, _ = eng.debug_mlp(o.bpts)                       # the full query's forward kernel on the same big-pose points
        e = (sdf3 - sdf1).abs()
        assert torch.isfinite(sdf3).all() and float(e.max()) < 6e-4 and float(e.mean()) < 6e-5, (n, float(e.max()), float(e.mean()))
    assert eng.hdq_sdf(x_all[:0], 0.125, True).numel() == 0
    # the three launch geometries (2 / 4 / 8 waves; the narrow ones run two row blocks at a time on a pair-ordered weight stream)
    # sum every accumulator's k-steps in the same order: a point's distance does not depend on the launch it is in, bit for bit
    big = eng.hdq_sdf(x_all, 0.125, False)                       # 70 000 points: 8 waves
    for n in (1000, 16384, 30000, 65536):                         # 2 waves up to 16 384, 4 waves up to 65 536
        assert torch.equal(eng.hdq_sdf(x_all[:n].contiguous(), 0.125, False), big[:n]), n
    # the 8-wave kernel spreads a partly filled LAST round of tiles over all workgroups (32 * w points each on w <= 4 waves, the other
    # waves only keep the weight stream going): 70 000 points = one round + 18 tiles -> spread with w = 1; 140 000 = two rounds + 35
    # tiles -> w = 2; 110 000 = one round + 174 tiles -> more than half the workgroups, plain deal.  Always the same bits per point.
    assert torch.equal(eng.hdq_sdf(x_all[65536:].contiguous(), 0.125, False), big[65536:])
    twice = torch.cat([x_all, x_all]).contiguous()
    for n in (140000, 110000, 131072 + 1, 131072 + 32 * 256 + 1):
        got = eng.hdq_sdf(twice[:n].contiguous(), 0.125, False)
        assert torch.equal(got[:70000], big) and torch.equal(got[70000:], big[:n - 70000]), n


@pytest.mark.parametrize('mode', ['relight', 'anisdf'])
def test_full_query_tile_boundaries(mode):
    """the reverse-mode full query (forward with tape + backward, 256-point tiles of 8 x 32 points; the tape is sized by whole
    tiles) for counts around every tile / wave boundary, both head variants (material heads, colour net): a point's raw channels
    — normals included — must not depend on the batch it is evaluated in, bit for bit (ragged last tile, dead waves, a single
    point, an empty set), and must agree with the oracle"""
    from oracle import ra_oracle as O
    cfg, net, dev = build(mode)
    body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
    eng = net.set_frame(body)
    g = torch.Generator().manual_seed(33)
    N = 2100
    vid = torch.randint(0, 6890, (N,), generator=g)
    wv = (body.pverts[0] @ body.R[0].T + body.Th[0])[vid.to(dev)]
    x_all = (wv + 0.002 * torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)).contiguous()     # all within dist_th
    v_all = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev).contiguous()
    th = 0.005
    c0 = eng.counters().n_fine_full
    whole = eng.forward(x_all, v_all, th).clone()
    assert eng.counters().n_fine_full - c0 == N and torch.isfinite(whole).all()
    nrm = whole[:, 13:16] if mode == 'relight' else whole[:, 9:12]
    assert float((nrm.norm(dim=-1) - 1).abs().max()) < 1e-4                          # unit normals out of the backward pass
    for n in (1, 31, 32, 33, 63, 65, 255, 256, 257, 511, 512, 513, 1023, 1025, 2047, 2049):
        part = eng.forward(x_all[:n].contiguous(), v_all[:n].contiguous(), th)
        assert part.shape == (n, whole.shape[1]) and torch.equal(part, whole[:n]), (mode, n, float((part - whole[:n]).abs().max()))
    assert eng.forward(x_all[:0], v_all[:0], th).shape[0] == 0
    # the same 257 points through the fp32 oracle
    n = 257
    onet = O.OracleNet(synthetic.make_state_dict(0, relight=mode == 'relight', cfg=cfg), cfg)
    ref, _ = O.network_forward(onet, x_all[:n].cpu(), v_all[:n].cpu(), O._frame(synthetic.make_body(0, posed=True)), th)
    e = err(whole[:n], ref)
    ncol = slice(13, 16) if mode == 'relight' else slice(9, 12)
    other = [c for c in range(whole.shape[1]) if not (ncol.start <= c < ncol.stop)]
    assert float(e[:, ncol].max()) < 1.5e-2 and float(e[:, ncol].mean()) < 1e-3, (float(e[:, ncol].max()), float(e[:, ncol].mean()))
    assert float(e[:, other].max()) < 2e-3, float(e[:, other].max())


def test_pose_frame_on_device(golden, relight):
    """N3: ra_pose_frame vs the reference's own functions (golden lbs.npz: bone transforms, T-pose / posed /
    world vertices, bounds) and vs the oracle (vertex normals, full vertex set)."""
    from oracle import ra_oracle as O
    _, _, dev, _, eng = relight
    g = golden('lbs.npz')
    sk = synthetic.make_skeleton(0)
    big_A = T(g['big_A'])
    o = eng.pose_frame(sk.poses, sk.tjoints, sk.parents, T(sk.tverts).to(dev), T(sk.weights).to(dev), big_A, sk.faces, sk.Rh, sk.Th)
    sel = T(g['sel'])
    assert float(err(o.A, g['A']).max()) < 1e-6 and float(err(o.joints, g['joints']).max()) < 1e-6 and float(err(o.R, g['R']).max()) < 1e-6
    assert float(err(o.tverts.cpu()[sel], g['txyz']).max()) < 3e-6
    assert float(err(o.pverts.cpu()[sel], g['pxyz']).max()) < 3e-6
    assert float(err(o.wverts.cpu()[sel], g['wxyz']).max()) < 3e-6
    ref = O.pose_frame(T(sk.poses), T(sk.tjoints), T(sk.parents), T(sk.tverts), T(sk.weights), big_A, T(sk.faces), T(sk.Rh), T(sk.Th))
    assert float(err(o.pverts, ref.pverts).max()) < 3e-6 and float(err(o.wverts, ref.wverts).max()) < 3e-6
    assert float(err(o.pnorm, ref.pnorm).max()) < 1e-4
    assert float(err(o.pbounds, ref.pbounds).max()) < 3e-6 and float(err(o.wbounds, ref.wbounds).max()) < 3e-6
    # determinism (fixed summation order) and reuse of the cached adjacency
    o2 = eng.pose_frame(sk.poses, sk.tjoints, sk.parents, T(sk.tverts).to(dev), T(sk.weights).to(dev), big_A, sk.faces, sk.Rh, sk.Th)
    assert torch.equal(o.pnorm, o2.pnorm) and torch.equal(o.pverts, o2.pverts)



def test_animation_inputs_are_asynchronous(relight):
    """N3 + N2 for an animation loop (round-3 verdict, item 3): ra_pose_frame stages its host inputs (the caller may overwrite them right
    after the call) and computes the bone transforms on the device; ra_gen_rays culls against a box that is still on the device; the ray
    count arrives behind an event.  Everything must equal the synchronous path bit for bit."""
    from relightableavatar_amd.data_utils import DeviceFrameLoader
    _, _, dev, _, eng = relight
    sk = synthetic.make_skeleton(0)
    tv, w = T(sk.tverts).to(dev), T(sk.weights).to(dev)
    eye = np.tile(np.eye(4, dtype=np.float32), (52, 1, 1))
    big_A = eng.pose_frame(sk.big_poses, sk.tjoints, sk.parents, tv, w, eye, sk.faces, np.zeros(3, np.float32), np.zeros(3, np.float32)).A.cpu().numpy()
    o = eng.pose_frame(sk.poses, sk.tjoints, sk.parents, tv, w, big_A, sk.faces, sk.Rh, sk.Th)
    torch.cuda.synchronize()
    p2, r2, t2 = sk.poses.copy(), sk.Rh.copy(), sk.Th.copy()
    o2 = eng.pose_frame(p2, sk.tjoints, sk.parents, tv, w, big_A, sk.faces, r2, t2)
    p2[:], r2[:], t2[:] = 9.0, 9.0, 9.0                 # the host arrays were staged before the call returned
    for k in ('A', 'joints', 'pverts', 'wverts', 'pnorm', 'R', 'wbounds', 'pbounds', 'tverts'):
        assert torch.equal(o[k], o2[k]), k
    assert torch.equal(o2.poses.cpu(), T(sk.poses)) and torch.equal(o2.Th.cpu(), T(sk.Th))
    K, R, Tc = synthetic.make_camera(96, 96)
    ref = eng.gen_rays(96, 96, K, R, Tc, o.wbounds.cpu())
    pend = eng.gen_rays_async(96, 96, K, R, Tc, o.wbounds, mask_to_host=True)
    got = pend.result()
    assert got.ray_o.shape == ref.ray_o.shape and got.ray_o.shape[0] > 100
    for k in ('ray_o', 'ray_d', 'near', 'far', 'mask_at_box'):
        assert torch.equal(got[k], ref[k]), k
    assert torch.equal(got.wbounds_host[0], o.wbounds.cpu()) and torch.equal(got.mask_host, ref.mask_at_box.reshape(-1).cpu())
    # the loader: a frame issued ahead renders exactly like the same frame assembled synchronously
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, _, _, _ = relight
    loader = DeviceFrameLoader(96, 96, K, R, Tc, sk.tjoints, sk.parents, tv, w, big_A, sk.faces)
    rend = make_renderer(cfg, net)
    pends = [loader.issue(eng, sk.poses * s, sk.Rh, sk.Th) for s in (1.0, 0.5)]          # two frames in the queue before either is consumed
    outs = [rend.render(loader.batch(p)).rgb_map.clone() for p in pends]
    again = rend.render(loader.batch(loader.issue(eng, sk.poses * 1.0, sk.Rh, sk.Th))).rgb_map
    assert torch.equal(again, outs[0]) and (outs[0].shape != outs[1].shape or not torch.equal(outs[0], outs[1]))
    assert float(outs[0].max()) > 0.05


def test_animated_frames_in_flight_are_bit_identical():
    """An ANIMATED sequence through the device-side loader with frames in flight: every frame poses a new body (ra_pose_frame) and culls
    its rays (ra_gen_rays) a pipeline turn ahead, on its replica's stream, while the other replicas render — and must equal the same
    frame posed, culled and rendered strictly one after the other, bit for bit, INCLUDING the loader's own outputs.
    Round 5: this failed in 10-15 % of the frames (tools/soak_pipeline.py's animated leg; it had not been part of the GPU suite):
    lbs_verts_kernel returned wrong z coordinates for groups of 16 consecutive vertices — the low half of compiler-formed packed-fp32
    chains (SLP: v_pk_fma_f32 / v_pk_add_f32 with op_sel swizzles) — whenever its waves shared SIMDs with another frame's MFMA kernels,
    never alone; the library is built with -fno-slp-vectorize throughout since (csrc/Makefile)."""
    from relightableavatar_amd.data_utils import DeviceFrameLoader
    from relightableavatar_amd.pipeline import FramePipeline
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('relight')
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    sk = synthetic.make_skeleton(0)
    H = 192
    K, R, Tc = synthetic.make_camera(H, H)
    tv, w = T(sk.tverts).to(dev), T(sk.weights).to(dev)
    eng0 = net.engine()
    eye = np.tile(np.eye(4, dtype=np.float32), (52, 1, 1))
    big_A = eng0.pose_frame(sk.big_poses, sk.tjoints, sk.parents, tv, w, eye, sk.faces, np.zeros(3, np.float32), np.zeros(3, np.float32)).A.cpu().numpy()
    loader = DeviceFrameLoader(H, H, K, R, Tc, sk.tjoints, sk.parents, tv, w, big_A, sk.faces)
    ph = np.arange(sk.poses.size, dtype=np.float32).reshape(sk.poses.shape)
    NA, N = 8, 32
    seq = [((sk.poses + 0.06 * np.sin(0.37 * f + ph)).astype(np.float32), sk.Rh, (sk.Th + np.float32(0.01 * np.sin(0.3 * f))).astype(np.float32)) for f in range(NA)]
    keys_in = ('pverts', 'pnorm', 'wbounds', 'ray_o')

    def one(rend, pend):
        b = loader.batch(pend)
        snap = {'in_' + k: b[k].clone() for k in keys_in}          # before the renderer grows wbounds in place
        out = rend.render(b)
        snap.update({k: out[k].clone() for k in ('rgb_map', 'acc_map', 'surf_map')})
        return snap
    serial = make_renderer(cfg, net)
    want = []
    for q in seq:
        want.append(one(serial, loader.issue(eng0, *q)))
        torch.cuda.synchronize()
    for depth in (2, 3):
        pipe = FramePipeline(cfg, sd, dev, depth=depth)
        ahead, fno = [None] * depth, [0]

        def frame(net_r, rend_r):
            f = fno[0]
            r = f % depth
            eng = net_r.engine()
            pend = ahead[r] or loader.issue(eng, *seq[f % NA])
            out = one(rend_r, pend)
            ahead[r] = loader.issue(eng, *seq[(f + depth) % NA])
            fno[0] += 1
            return out
        pend = [pipe.submit(fn=frame) for _ in range(N)]
        bad = {}
        for k, p in enumerate(pend):
            out = p.result()
            for key, x in out.items():
                y = want[k % NA][key]
                if x.shape != y.shape or not torch.equal(x, y):
                    bad.setdefault(key, []).append(k)
        torch.cuda.synchronize()
        assert not bad, (depth, {k: v[:6] for k, v in bad.items()})


def test_frame_novel_ground(golden):
    """the README's relight command (readme.md:64: vis_novel_light + vis_ground_shading): main + every probe, the human AND the
    ground layer re-shaded per probe and blended per light (novel_light_sphere_tracing.py:70-99,138-213) vs the reference's
    own frame (smooth skinning field -> the SURVEY.md:409 contract applies)"""
    from relightableavatar_amd.renderer import make_renderer
    ref = golden('frame_novel_ground.npz')
    kw = dict(vis_ground_shading=True, ground_normal=[float(v) for v in ref['ground_normal']],
              ground_origin=[float(v) for v in ref['ground_origin']], render_chunk_size=int(ref['render_chunk_size']))
    cfg, net, dev = build('novel_light', **kw)
    H = int(ref['H'])
    batch = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True, crop=int(ref['crop']), n_novel_lights=2,
                                                     skin_noise=float(ref['skin_noise'])), dev)
    rend = make_renderer(cfg, net)
    m = batch.mask_at_box.reshape(1, -1).cpu()
    rend.ground_inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]      # the scatter order of the CPU reference run
    out = rend.render(batch)
    np.testing.assert_allclose(batch.wbounds.cpu().numpy(), ref['wbounds_after'], atol=1e-6)
    assert set(out.keys()) == {'main', 'probe00', 'probe01', 'diff'}
    for name in ('main', 'probe00', 'probe01'):
        sub = {k[len(name) + 1:]: v for k, v in ref.items() if k.startswith(name + '.')}
        o = out[name]
        assert o.rgb_map.shape == (1, H * H, 3)
        p, mx = psnr(o.rgb_map, sub['rgb_map']), float(err(o.rgb_map, sub['rgb_map']).max())
        print(f'frame_novel_ground {name}: rgb PSNR {p:.1f} dB, max {mx:.2e}')
        # SURVEY.md:409's contract over ALL pixels.  Round 6: 80.5-84.7 dB, max 1.1e-3 .. 1.9e-3.  Up to round 5 one interior pixel of `main`
        # sat at 1.1e-2 (its brightest light grazes the body: that light's visibility off by 0.2 on plain f16 operands) and this test
        # asserted max <= 2e-2, <= 3 elements over 1e-2; the brightest lights are key lights now (cfg.key_light_share)
        assert p >= 60.0 and mx <= 1e-2
        assert psnr(o.shade_map, sub['shade_map']) >= 50.0 and float(err(o.spec_map, sub['spec_map']).max()) < 1e-2
        assert float(err(o.albedo_map, sub['albedo_map']).max()) < 2e-3 and float(err(o.acc_map, sub['acc_map']).max()) < 2e-2
    assert float((out.probe00.rgb_map - out.probe01.rgb_map).abs().max()) > 0.05


def test_frame_anisdf_volume_128_samples(golden):
    """BASELINE config 2's sample count (128 per ray; the depth-major tiled sample layout depends on S)"""
    out, ref, _, _ = _frame('anisdf', 'frame_anisdf128.npz', golden)
    assert int(ref['n_samples']) == 128
    for k, tol in (('acc_map', 5e-4), ('depth_map', 1e-3), ('cpts_map', 2e-4), ('resd_map', 1e-5), ('norm_map', 2e-3), ('rgb_map', 3e-4)):
        within(out, ref, k, tol, 1.0)
    assert psnr(out.rgb_map, ref['rgb_map']) > 80 and float(err(out.rgb_map, ref['rgb_map']).max()) <= 1e-2


def test_network_field_methods(golden, relight):
    """the Network surface the reference renderer binds for its ablation modes (sphere_tracing_renderer.py:955-961)"""
    _, net, dev, body, eng = relight
    g = golden('fields.npz')
    x = T(g['obs_x'])[None].to(dev)
    assert float(err(net.inference_observed_distance_field(x, body)[0], g['obs_sdf']).max()) < 3e-4
    assert float(err(net.inference_observed_distance_field(x, body, smooth_transition=True, filtering=True, dist_th=0.125)[0], g['obs_sdf_filtered']).max()) < 3e-4
    assert float(err(net.inference_observed_distance_field(x, body, smooth_transition=False, filtering=True, dist_th=0.125)[0],
                     g['obs_sdf_filtered_nosmooth']).max()) < 3e-4
    # rows in point order here; the reference returns them in geodesic_knn's compaction order (w2b_inds / b2w_inds)
    w2b = net.world_to_bigpose_transform(T(g['w2b_x'])[None].to(dev), body)
    assert w2b.shape == (1, 300, 4, 4)
    assert float(err(w2b[0].cpu()[T(g['w2b_inds'])], g['w2b']).max()) < 5e-6
    b2w = net.bigpose_to_world_transform(x, body)
    assert float(err(b2w[0].cpu()[T(g['b2w_inds'])], g['b2w']).max()) < 5e-6
    # the frame state is restored after the template-space queries
    s = net.inference_world_distance_field(T(golden('ops.npz')['hdq_x'])[None].to(dev), body, smooth_transition=True, dist_th=0.125)
    assert float(err(s[0], golden('ops.npz')['hdq_sdf']).max()) < 3e-4


def test_fix_material_rule(golden):
    """base_network.py:501-503: with fix_material = -1 (always_fix_material) the colour net sees train_motion.poses[:, -1];
    a batch without train_motion is an error, as in the reference"""
    g = golden('fixmat.npz')
    cfg, net, dev = build('anisdf', fix_material=-1)
    body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
    raw = net(T(g['x'])[None].to(dev), T(g['v'])[None].to(dev), 0.005, body).raw[0]
    assert float(err(raw[:, 12:15], T(g['raw'])[:, 12:15]).max()) < 2e-3                # rgb under the last training pose
    assert float(err(raw[:, 0:9], T(g['raw'])[:, 0:9]).max()) < 5e-6
    cfg0, net0, _ = build('anisdf', fix_material=0)
    raw0 = net0(T(g['x'])[None].to(dev), T(g['v'])[None].to(dev), 0.005, body).raw[0]
    assert float((raw0[:, 12:15] - raw[:, 12:15]).abs().max()) > 1e-3
    nb = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
    del nb['train_motion']
    with pytest.raises(ValueError, match='train_motion'):
        net(T(g['x'])[None].to(dev), T(g['v'])[None].to(dev), 0.005, nb)


def test_new_frames_are_never_mistaken_for_the_cached_one():
    """render frame A, free it, render frame B allocated the same way (CPU-resident batches that the renderer moves
    itself): B must be rendered with B's pose — the cache keys on live tensor identity, never on recycled addresses"""
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('sphere_tracing')
    rend = make_renderer(cfg, net)
    outs = {}
    for tag, seed in (('a', 0), ('b', 5), ('a2', 0)):
        b = synthetic.make_batch(64, 64, seed=seed, posed=True, crop=16)            # CPU tensors, freed after each frame
        outs[tag] = rend.render(b).rgb_map.clone()
        del b
    assert torch.equal(outs['a'], outs['a2'])
    assert float((outs['a'] - outs['b']).abs().max()) > 1e-2
    fresh_cfg, fresh_net, _ = build('sphere_tracing')
    ref_b = make_renderer(fresh_cfg, fresh_net).render(synthetic.make_batch(64, 64, seed=5, posed=True, crop=16)).rgb_map
    assert torch.equal(outs['b'], ref_b)


def test_unused_stage_fixtures(ops, relight):
    """fixtures of ops.npz that only the oracle used to read: the shadow-ray box clip incl. its direction-clamp quirk
    (net_utils.py:1698), light_visibility on given surface points, the microfacet BRDF on arbitrary direction pairs"""
    cfg, net, dev, body, eng = relight
    near, far = eng.debug_aabb(ops['aabb_o'].to(dev), ops['aabb_d'].to(dev), ops['aabb_bounds'].reshape(-1).tolist())
    e_n, e_f = err(near, ops['aabb_near']), err(far, ops['aabb_far'])
    big = (T(np.abs(ops['aabb_near'].numpy()) > 1e6)) | (T(np.abs(ops['aabb_far'].numpy()) > 1e6))      # divisions by the 1e-8 clamp value
    assert float(e_n[~big].max()) < 1e-5 and float(e_f[~big].max()) < 1e-5
    assert float((e_n[big] / ops['aabb_near'][big].abs()).max()) < 1e-5 if bool(big.any()) else True
    brdf = eng.debug_brdf(ops['mf_p2l'].to(dev), ops['mf_p2c'].to(dev), ops['mf_n'].to(dev), ops['mf_albedo'].to(dev), ops['mf_rough'].to(dev))
    ref = ops['mf_brdf']
    assert float((err(brdf, ref) / (ref.abs() + 1e-3)).max()) < 2e-3
    lvis, ldot = eng.debug_lvis(ops['lv_surf'].to(dev), ops['lv_norm'].to(dev), ops['lv_acc'].to(dev), ops['lv_bbox'].reshape(-1).tolist())
    assert float(err(ldot.T, ops['lv_ldot']).max()) < 1e-5
    e = err(lvis.T, ops['lv_lvis'])
    assert float(e.mean()) < 3e-3 and float((e < 3e-2).float().mean()) > 0.98
    front = ops['lv_ldot'] > 1e-4
    assert float(lvis.T.cpu()[ops['lv_ldot'] < -1e-4].abs().max()) == 0.0                  # back-facing lights: exactly 0
    # ... and with every distance query compensated (cfg.trace_precision 2) the same 24 x 512 visibilities agree with the reference's to
    # its own fp32 noise: the tolerances above are the plain tier's operand rounding x d * sharp / (2 t)
    cfg2, net2, _ = build('relight', trace_precision=2)
    eng2 = net2.set_frame(synthetic.to_device(synthetic.make_body(0, posed=True), dev))
    lvis2, _ = eng2.debug_lvis(ops['lv_surf'].to(dev), ops['lv_norm'].to(dev), ops['lv_acc'].to(dev), ops['lv_bbox'].reshape(-1).tolist())
    e2 = err(lvis2.T, ops['lv_lvis'])
    print(f'light_visibility on 24 surface points x 512 lights: plain tier mean {float(e.mean()):.2e} max {float(e.max()):.2e}; all compensated mean {float(e2.mean()):.2e} '
          f'max {float(e2.max()):.2e}, within 1e-3: {float((e2 < 1e-3).float().mean()) * 100:.2f} %')
    # measured: plain tier mean 4.7e-4 with ONE of the 12 288 rays flipped outright (0 instead of 1: the state machine's own discontinuity,
    # reached by a 5e-5 distance error); all compensated mean 1.8e-6, max 2.4e-4
    assert float(e2.mean()) < 1e-4 and float(e2.max()) < 1e-3


def test_sharded_multi_chunk_frame_equals_the_whole_frame():
    """a frame that spans several render chunks: the per-chunk in-place growth of batch.wbounds (quirk 1) makes a ray's shadow
    box depend on its chunk; shards carry the frame's chunk boundaries (render_chunks), so merged shards == the whole frame"""
    from relightableavatar_amd import shard
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('relight', render_chunk_size=700)
    rend = make_renderer(cfg, net)
    base = synthetic.to_device(synthetic.make_batch(128, 128, seed=0, posed=True), dev)
    P = base.ray_o.shape[1]
    assert P > 3 * 700
    wb0 = base.wbounds.clone()
    whole = rend.render(base).rgb_map.clone()
    grown = base.wbounds.clone()
    for world in (2, 3):
        merged = torch.zeros_like(whole[0])
        for r in range(world):
            base.wbounds.copy_(wb0)
            sb = shard.shard_batch(base, r, world, cfg.render_chunk_size)
            merged[shard.shard_indices(P, r, world, base, merged.device)] = rend.render(sb).rgb_map[0]
            assert torch.equal(sb.wbounds, grown)                                   # every shard grew the box as often as the frame did
        assert torch.equal(merged, whole[0]), world


def test_visualiser_normalisations(golden, relight):
    """N4, third item: Visualizer.generate_image through ra_map_to_image / ra_add_light_probe vs the reference's
    own images (visual.npz) for every output type of the hot path, from the reference frame's maps (so no MLP noise enters)"""
    from relightableavatar_amd import config
    from relightableavatar_amd.base_utils import dotdict
    from relightableavatar_amd.visualizers import Output, Visualizer
    cfg, net, dev, body, eng = relight
    g = golden('visual.npz')
    ref = golden('frame_relight_smooth.npz')
    H = int(ref['H'])
    batch = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True, crop=int(ref['crop']), skin_noise=float(ref['skin_noise'])), dev)
    batch.cam_R = synthetic.tilted_cam_R().to(dev)
    out = dotdict({k: T(v).to(dev) for k, v in ref.items() if k.endswith('_map')})
    out.envmap = dotdict(probe=net.global_env_map[None])
    config.set_active_cfg(cfg)
    Visualizer.engine = eng
    for t in (Output.Surface, Output.Residual, Output.Depth, Output.Alpha, Output.Normal, Output.Specular, Output.Albedo, Output.Roughness,
              Output.Shading, Output.Rendering):
        img = T(Visualizer.generate_image(out, batch, t))
        r = T(g[f'img_{t.name}'])
        assert img.shape == r.shape == (H, H, 4), t
        assert bool((img.isnan() == r.isnan()).all()), t
        assert float((img - r).nan_to_num(0.0).abs().max()) < 5e-5, (t, float((img - r).nan_to_num(0.0).abs().max()))
    alt = make_cfg('relight', normalize_shading=True, store_alpha_channel=False, probe_size_ratio=0.0, tonemapping_albedo=False)
    config.set_active_cfg(alt)
    for t in (Output.Shading, Output.Albedo, Output.Rendering):
        img = T(Visualizer.generate_image(out, batch, t))
        assert img.shape == (H, H, 3) and float((img - T(g[f'alt_{t.name}'])).abs().max()) < 5e-5, t
    o2 = dotdict(out)
    o2.depth_map = torch.where(torch.isfinite(out.depth_map), out.depth_map, torch.full_like(out.depth_map, 1.7))
    img = T(Visualizer.generate_image(o2, batch, Output.Depth))
    assert bool(torch.isfinite(img).all()) and float((img - T(g['alt_Depth_finite'])).abs().max()) < 5e-5
    assert np.allclose(Visualizer.generate_image(out, batch, Output.Envmap), g['img_Envmap'], atol=1e-6)      # torch's softplus on the GPU vs the CPU
    config.set_active_cfg(cfg)
    # full-frame maps (what the ground pass returns): no scatter, identity pixel order
    full = dotdict(rgb_map=torch.rand(1, H * H, 3, device=dev), acc_map=torch.rand(1, H * H, device=dev))
    img = T(Visualizer.generate_image(full, batch, Output.Rendering))
    assert torch.equal(img[..., :3].reshape(-1, 3)[H * 30:], full.rgb_map[0].cpu()[H * 30:])          # below the probe inset
    assert torch.equal(img[..., 3].reshape(-1), full.acc_map[0].cpu())


def test_full_size_properties_volume_config2():
    """BASELINE config 2 at full size (512x512, 128 samples per ray, 8192-ray chunks): finite, bounded, deterministic, and the
    image does not depend on how the rays are chunked or dealt to ranks (rays are independent)"""
    from relightableavatar_amd import shard
    from relightableavatar_amd.renderer import make_renderer
    cfg, net, dev = build('anisdf', volume_chunk_rays=0)                            # the reference's chunking: 8192 rays
    assert cfg.n_samples == 128 and cfg.render_chunk_size == 8192
    rend = make_renderer(cfg, net)
    base = synthetic.to_device(synthetic.make_batch(512, 512, seed=0, posed=True), dev)
    out = rend.render(base)
    rgb, acc = out.rgb_map.clone(), out.acc_map.clone()
    assert torch.isfinite(rgb).all() and float(rgb.min()) >= 0 and float(rgb.max()) <= 1.0 + 1e-5
    assert float(acc.min()) >= 0 and float(acc.max()) <= 1.0 + 1e-5 and 0.2 < float((acc > 0.5).float().mean()) < 0.9
    nn = out.norm_map.norm(dim=-1)                                                   # a weighted sum of unit normals: |sum w n| <= sum w
    assert float((nn - acc).max()) < 2e-3 and float(nn[acc > 0.99].median()) > 0.3
    assert torch.equal(rend.render(base).rgb_map, rgb)                              # deterministic
    for kw in (dict(render_chunk_size=3000, volume_chunk_rays=0), dict()):          # other chunkings (3000 rays; the device-sized default: one launch sequence) -> same pixels
        cfg2, net2, _ = build('anisdf', **kw)
        rgb2 = make_renderer(cfg2, net2).render(base).rgb_map
        assert float((rgb2 - rgb).abs().max()) == 0.0
        del net2
    P = rgb.shape[1]
    merged = torch.zeros_like(rgb[0])
    for r in range(2):
        merged[shard.shard_indices(P, r, 2, base, merged.device)] = rend.render(shard.shard_batch(base, r, 2, cfg.render_chunk_size)).rgb_map[0]
    assert float((merged - rgb[0]).abs().max()) == 0.0
    c = net.engine().counters()
    assert c.n_fine_full > 1_000_000


def test_full_size_properties_config5():
    """BASELINE config 5 on one GPU: 1024x1024 full relight + 8 novel probes re-shaded in one launch; the frame spans three
    render chunks, so the 2-shard merge exercises the per-chunk box growth (render_chunks); batched probes == one by one"""
    from relightableavatar_amd import shard
    from relightableavatar_amd.renderer import make_renderer
    # (key_light_share 0: which shadow rays run compensated depends on the SET of probes a frame is shaded with, so "one probe alone == inside
    # the batch" is exact only without the key-light tier; test_config5_key_light_tier_at_full_size holds the tier itself at this size)
    cfg, net, dev = build('novel_light', test_light=[], key_light_share=0.0)
    rend = make_renderer(cfg, net)
    base = synthetic.to_device(synthetic.make_batch(1024, 1024, seed=0, posed=True, n_novel_lights=8), dev)
    P = base.ray_o.shape[1]
    assert P > 2 * cfg.render_chunk_size
    wb0 = base.wbounds.clone()
    out = rend.render(base)
    names = list(base.novel_lights.keys())
    assert len(names) == 8 and all(n in out for n in names)
    whole = torch.cat([out[n].rgb_map for n in names], dim=-1).clone()                 # (1, P, 24)
    assert torch.isfinite(whole).all() and float(whole.min()) >= 0 and float(whole.max()) <= 1.0 + 1e-6
    assert float((out[names[0]].rgb_map - out[names[1]].rgb_map).abs().max()) > 0.05
    merged = torch.zeros_like(whole[0])
    for r in range(2):
        base.wbounds.copy_(wb0)
        o = rend.render(shard.shard_batch(base, r, 2, cfg.render_chunk_size))
        merged[shard.shard_indices(P, r, 2, base, merged.device)] = torch.cat([o[n].rgb_map for n in names], dim=-1)[0]
    assert float((merged - whole[0]).abs().max()) == 0.0
    # one probe alone gives the same image as inside the batch of 8
    base.wbounds.copy_(wb0)
    from relightableavatar_amd.base_utils import dotdict
    single = dotdict(base)
    single.novel_lights = dotdict({names[3]: base.novel_lights[names[3]]})
    o1 = rend.render(single)
    assert float((o1[names[3]].rgb_map - out[names[3]].rgb_map).abs().max()) == 0.0


def test_key_lights_match_the_rule(relight):
    """csrc/ra_trace.hip key_lights_kernel against the rule's restatement (oracle.key_lights): a light is a key light when it holds >=
    max(key_light_share, 4 / L) of a probe's power under ANY of the frame's probes, the 48 lights with the largest such share at most;
    probes of two sizes accumulate (the learned 32 x 64 map + novel 16 x 32 probes), and n = 0 returns to per-call key lights"""
    from oracle import ra_oracle as O
    cfg, net, dev, body, eng = relight
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    on = O.OracleNet(sd, cfg)
    L = cfg.env_h * cfg.env_w
    lights = synthetic.make_novel_lights(8, 0)
    novel = torch.stack([lights[k].probe[0] for k in lights])                 # 7 lognormal + 1 OLAT-style
    front = O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg, env='front'), cfg).global_env_map
    for probes_sets in ([on.global_env_map], [front], [on.global_env_map, novel], [novel[-1:]]):
        eng.set_key_probes([p.to(dev) for p in probes_sets])
        key, share = eng.debug_key_lights(L)
        flat = [p if p.ndim == 3 else None for p in probes_sets]
        allp = [p for p in probes_sets if p.ndim == 3] + [q for p in probes_sets if p.ndim == 4 for q in p]
        want = O.key_lights(on, allp, cfg.key_light_share)
        d = normalize_rows(on.light_xyz.reshape(-1, 3))
        smax = torch.zeros(L)
        for pr in allp:
            w = (O.sample_envmap_image(pr, d).mean(-1) * on.light_area.reshape(-1)).clamp_min(0)
            smax = torch.maximum(smax, w / w.sum())
        assert float((share.cpu() - smax).abs().max()) < 1e-5 * float(smax.max()) + 1e-9          # fp32 sums in another order
        thr = max(cfg.key_light_share, 4.0 / L)
        clear = (smax - thr).abs() > 1e-5 * thr                               # lights not sitting on the threshold itself
        got = key.cpu()
        if int(want.sum()) < 48:                                              # below the cap the rule is a plain threshold
            assert bool((got == want)[clear].all()), (int(got.sum()), int(want.sum()))
        else:                                                                 # at the cap: the 48 largest shares
            assert int(got.sum()) == 48 and float(smax[got].min()) >= float(smax[~got].max()) - 1e-7
        print(f'{len(allp)} probe(s): {int(got.sum())} key lights, largest share {float(smax.max()):.3f}')
    eng.set_key_probes([])
    with pytest.raises(_lib_error()):
        eng.debug_key_lights(L)


def _lib_error():
    from relightableavatar_amd import _lib
    return _lib.RaError


def normalize_rows(v):
    return v / (v.norm(dim=-1, keepdim=True) + 1e-8)


def test_config5_key_light_tier_at_full_size():
    """cfg.key_light_share at BASELINE config 5's size (1024 x 1024, 8 probes on the smooth body: seven heavy-tailed lognormal ones and an
    OLAT-style one, a light of 100 over an ambient 0.25): every probe's frame with round 5's tiers (all shadow rays plain f16), with the
    shipped tiers (+ the rays towards the frame's <= 48 key lights compensated) and with every distance query compensated.  The surface
    trace is the same arithmetic in all three (identical hit masks).  What this size shows (profiles/r06_hard_cases.txt):
      * the DFSS state machine has discontinuities of its own (the accept conditions of the claybook estimate, :157-172; occ == 0 ends a
        ray): a 5e-5 distance error flips a few of a frame's 18 M shadow rays outright, and a flipped ray towards a light that holds 1-2 %
        of a heavy-tailed probe's power moves its pixel by 2e-2 .. 2e-1 — 14-35 of 71 492 hit pixels per probe with round 5's tiers;
      * the key lights (48 at most) remove the flips that matter most: the OLAT probe goes from 21 pixels / 0.23 to <= 2 / 0.03, the others
        lose two fifths of theirs (9-17 left), at 11 % of the fine queries compensated;
      * max |err| <= 1e-2 on EVERY pixel of a heavy-tailed probe needs every ray compensated (trace_precision 2): per pixel the bound holds
        on >= 99.95 % of the hit pixels in both tiers, asserted below."""
    from relightableavatar_amd.renderer import make_renderer
    frames, comp = {}, {}
    for label, tp, share in (('round5', 1, 0.0), ('shipped', 1, 0.0078), ('all', 2, 0.0)):
        cfg, net, dev = build('novel_light', test_light=[], trace_precision=tp, key_light_share=share, novel_light_timing=False)
        out = make_renderer(cfg, net).render(synthetic.to_device(synthetic.make_batch(1024, 1024, seed=0, posed=True, n_novel_lights=8, skin_noise=0.0), dev))
        names = [k for k in out if k != 'diff']
        frames[label] = {n: out[n].rgb_map.clone() for n in names}
        frames[label]['acc'] = out[names[0]].acc_map.clone()
        c = net.engine().counters()
        comp[label] = c.n_fine_sdf_comp / max(c.n_fine_sdf, 1)
        del out, net
    assert torch.equal(frames['round5']['acc'], frames['all']['acc']) and torch.equal(frames['shipped']['acc'], frames['all']['acc'])
    hit = frames['all']['acc'][0] > 0
    worst, over = {}, {}
    for label in ('round5', 'shipped'):
        worst[label] = {n: float((frames[label][n] - frames['all'][n]).abs().max()) for n in names}
        over[label] = {n: int(((frames[label][n] - frames['all'][n]).abs()[0].amax(-1) > 1e-2).sum()) for n in names}
        print(f'config 5, {label} tiers vs all-compensated, per probe: max |diff| ' + ', '.join(f'{n} {v:.2e}' for n, v in worst[label].items()) +
              f'; pixels over 1e-2: {over[label]} of {int(hit.sum())} hit pixels; compensated share of the fine queries {comp[label]:.3f}')
    nh = int(hit.sum())
    olat = names[-1]
    assert over['shipped'][olat] <= 3 and over['round5'][olat] >= 3 * max(over['shipped'][olat], 1), (over['round5'][olat], over['shipped'][olat])
    assert sum(over['shipped'].values()) < sum(over['round5'].values())
    assert all(v <= 4e-4 * nh for v in over['shipped'].values()) and all(v <= 7e-4 * nh for v in over['round5'].values())      # measured: <= 17 / <= 35 of 71 492
    assert comp['shipped'] < 0.16, comp            # 2 % (the surface trace) + the key lights' rays (48 lights at most)
