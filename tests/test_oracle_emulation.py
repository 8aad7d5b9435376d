"""The oracle's operand-rounding emulation (16-bit MFMA operands, fp32 accumulate) and the committed precision floors.  CPU only."""
import json
import os

import numpy as np
import torch

from oracle import ra_oracle as O
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg

HERE = os.path.dirname(os.path.abspath(__file__))
T = torch.from_numpy


def _nets(**kw):
    cfg = make_cfg('relight')
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    return cfg, sd


def test_emulation_off_is_the_fp32_oracle_and_on_rounds_operands(golden):
    cfg, sd = _nets()
    ops = golden('ops.npz')
    fr = O._frame(synthetic.make_body(0, posed=True))
    x = T(ops['mlp_bpts'])
    plain = O.observed_sdf(O.OracleNet(sd, cfg), x, fr)
    assert torch.equal(plain, O.observed_sdf(O.OracleNet(sd, cfg, emulate='f32'), x, fr))
    assert float((plain - T(ops['mlp_sdf'])).abs().max()) < 2e-6
    e16 = float((O.observed_sdf(O.OracleNet(sd, cfg, emulate='f16'), x, fr) - plain).abs().mean())
    eb16 = float((O.observed_sdf(O.OracleNet(sd, cfg, emulate='bf16'), x, fr) - plain).abs().mean())
    assert 1e-5 < e16 < 2e-4 and 4 * e16 < eb16 < 2e-3            # 3 mantissa bits apart
    # the kernel-like variant (fp32 pose bias, hi + lo coordinates) is no worse than plain rounding
    ek = float((O.observed_sdf(O.OracleNet(sd, cfg, emulate='f16', kernel_like=True), x, fr) - plain).abs().mean())
    assert ek < 1.1 * e16


def test_committed_floor_is_reproducible(golden):
    """re-derives one entry of tests/golden/precision_floor.json (tools/precision_floor.py) — the frame the SURVEY.md:409
    contract is asserted on — and checks what the file says about the SURVEY 8d body"""
    floors = json.load(open(os.path.join(HERE, 'golden', 'precision_floor.json')))
    ref = golden('frame_relight_smooth.npz')
    cfg = make_cfg('relight', vis_specular_map=True)
    net = O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg), cfg, emulate='f16', kernel_like=True)
    batch = synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']), skin_noise=float(ref['skin_noise']))
    out = O.render_sphere_tracing(net, batch)
    p = O.psnr(out.rgb_map, T(ref['rgb_map']))
    f = floors['frame_relight_smooth.npz:f16']['rgb_map']
    assert abs(p - f['psnr']) < 1.0, (p, f)                       # thread-count dependent summation order moves it a little
    assert f['psnr'] >= 60 and f['max_abs'] <= 1e-2               # f16 operands meet the contract where the trace converges
    g = floors['frame_relight.npz:f16']['rgb_map']
    assert g['max_abs'] > 1e-2 and g['psnr_trim2pct'] > g['psnr'] + 10     # ... and cannot on the SURVEY 8d body: 2 % of the pixels decide
    assert floors['frame_relight.npz:bf16']['rgb_map']['psnr'] < g['psnr'] - 4


def test_reference_trace_limit_cycle_on_the_survey_body():
    """why: on the SURVEY 8d body the fp32 ORACLE's own surface trace does not converge on ~9 % of the hit rays (|d| of each of the
    last three iterations stays above a millimetre: a limit cycle across a jump of the warp), several times more often than
    with a smooth skinning field"""
    cfg, sd = _nets()
    net = O.OracleNet(sd, cfg)
    res = {}
    for sn in (2.0, 0.0):
        b = synthetic.make_batch(128, 128, seed=0, posed=True, crop=16, skin_noise=sn)
        fr = O._frame(b)
        log = []

        def f(x):
            s = O.hdq_sdf(net, x, fr, cfg.dist_th, True)
            log.append(s[:, 0].abs().clone())
            return s
        O.sphere_tracing(b.ray_o[0], b.ray_d[0], b.near[0][:, None], b.far[0][:, None], f, iter=16, offset=0.02)
        res[sn] = float((torch.stack(log[-3:]).amin(0) > 1e-3).float().mean())       # rays still > 1 mm off the surface
    assert res[2.0] > 0.05 and res[0.0] < 0.5 * res[2.0], res


def test_compensated_and_float64_variants_of_the_oracle():
    """round 4's deciding experiment (tools/precision_tiers.py): `f64acc` (every nn.Linear in float64, rounded once) is a differently
    associated fp32 and must agree with the fp32 oracle at rounding level; `f16x2` (f16 hi + lo operand pairs, three products: the
    arithmetic of the HIP kernel K3C) must be as accurate as fp32 itself, two orders of magnitude better than plain f16 operands"""
    cfg, sd = _nets()
    fr = O._frame(synthetic.make_body(0, posed=True))
    g = torch.Generator().manual_seed(11)
    d = torch.nn.functional.normalize(torch.randn(4000, 3, generator=g), dim=-1)
    x = d * (0.38 + 0.12 * torch.rand(4000, 1, generator=g))
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt())
    f64 = O.observed_sdf(O.OracleNet(sd, cfg, emulate='f64acc'), x, fr)
    f32 = O.observed_sdf(O.OracleNet(sd, cfg), x, fr)
    x2 = O.observed_sdf(O.OracleNet(sd, cfg, emulate='f16x2', kernel_like=True), x, fr)
    f16 = O.observed_sdf(O.OracleNet(sd, cfg, emulate='f16', kernel_like=True), x, fr)
    assert rms(f32, f64) < 3e-7 and rms(x2, f64) < 3e-7 and rms(f16, f64) > 2e-5
    tiers = json.load(open(os.path.join(HERE, 'golden', 'precision_tiers.json')))
    for case in ('frame_relight', 'bench_sample', 'full_size_sample'):
        assert tiers[f'{case}:f64acc']['rgb_map']['psnr'] > 85 and tiers[f'{case}:f64acc']['rgb_map']['rays_over_1e2'] == 0       # fp32 pins these frames
        t, k4, plain = (tiers[f'{case}:{v}']['rgb_map'] for v in ('tier', 'tier_k4', 'f16'))
        assert t['psnr'] >= 60 and t['max_abs'] <= 1e-2 and abs(t['psnr'] - k4['psnr']) < 0.5      # only the surface trace needs the tier
        assert plain['max_abs'] > 1e-2                                                              # plain f16 operands do not meet the contract


def test_fp32_unstable_rays_fixture_is_reproducible():
    """tests/golden/fp32_unstable_rays.json (tools/fp32_stability.py): re-derive the smallest ray set and check the shape of all"""
    fx = json.load(open(os.path.join(HERE, 'golden', 'fp32_unstable_rays.json')))
    cfg, sd = _nets()
    net = O.OracleNet(sd, cfg)
    bad = O.fp32_unstable_rays(net, synthetic.make_batch(128, 128, seed=0, posed=True, crop=8))
    assert [int(i) for i in bad.nonzero()[:, 0]] == fx['smoke']['unstable'] == fx['multi_chunk']['unstable']
    for k, v in fx.items():
        if k.startswith('_'):
            continue
        # a handful of rays, not a tolerance in disguise: <= 2 % — and 5 % on the one hard case that stacks everything the reference's trace
        # dislikes (trained-like weights on the noisy skinning field of a body pulled apart: 6 of 144 rays)
        assert len(v['unstable']) <= max(2, (0.05 if k == 'switches.npz:sharp_split' else 0.02) * v['n_rays']), k
    # the mechanism: noise far below any rendering tolerance moves such a ray's surface point by millimetres
    st0, _ = O.surface_trace(net, synthetic.make_batch(128, 128, seed=0, posed=True, crop=8))
    moved = 0
    for s in range(8):
        st, _ = O.surface_trace(net, synthetic.make_batch(128, 128, seed=0, posed=True, crop=8), 1e-7, torch.Generator().manual_seed(s))
        moved += int((st - st0).abs()[bad].max() > 1e-3)
    assert moved >= 1
