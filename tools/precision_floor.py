#!/usr/bin/env python3
"""Precision floor of 16-bit MFMA operands on the golden frames (CPU, oracle only — test infrastructure).

Renders each golden relight frame with the oracle's operand-rounding emulation (every nn.Linear input and weight rounded to
f16 / bf16, fp32 accumulate — what SURVEY.md:305 measured — in the kernel-like variant: pose condition through an fp32 bias,
hi + lo coordinates on the SDF net's encoding-fed layers) and compares with the REFERENCE's own fp32 output
(tests/golden/frame_*.npz).  No 16-bit-operand kernel can be closer to the reference than this on that frame; the GPU parity
tests hold the HIP path to these floors (tests/test_gpu_parity.py) and tests/test_oracle_emulation.py re-derives one of them.

    python tools/precision_floor.py            # rewrites tests/golden/precision_floor.json (about 4 minutes on 8 cores)

Findings recorded in DESIGN.md section 2: on the SURVEY 8d body (per-vertex noise in the skinning logits) the world -> big-pose
warp jumps by ~1 cm wherever the nearest vertices change, the reference's own 16-iteration sphere trace ends in a limit cycle on
~9 % of the hit rays, and a 1e-4 distance perturbation flips the phase of that cycle on ~1 % of the pixels (4 mm surface jumps, 0.05 rgb):
the rgb PSNR of ANY 16-bit path is set by those one or two pixels.  With a spatially smooth skinning field (skin_noise = 0, a
real SMPL body's situation) the trace converges and f16 operands reproduce the reference to > 60 dB.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ra_oracle as O                      # noqa: E402
from relightableavatar_amd import synthetic            # noqa: E402
from relightableavatar_amd.config import make_cfg      # noqa: E402

FRAMES = {'frame_relight.npz': 2.0, 'frame_relight_smooth.npz': 0.0}


def stats(a, b):
    a, b = a.float().reshape(-1, a.shape[-1]), b.float().reshape(-1, b.shape[-1])
    e = (a - b).abs()
    per_pix = e.amax(-1)
    keep = per_pix <= per_pix.kthvalue(max(1, int(round(0.98 * per_pix.numel())))).values      # drop the worst 2 % of the pixels
    return dict(psnr=round(O.psnr(a, b), 2), psnr_trim2pct=round(O.psnr(a[keep], b[keep]), 2), max_abs=float(e.max()),
                n_over_1e2=int((e > 1e-2).sum()), mean_abs=float(e.mean()))


def frame_floor(fname, skin_noise, emu):
    ref = dict(np.load(os.path.join(ROOT, 'tests', 'golden', fname)))
    cfg = make_cfg('relight', vis_specular_map=True)
    net = O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg), cfg, emulate=emu, kernel_like=True)
    batch = synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']), skin_noise=skin_noise)
    out = O.render_sphere_tracing(net, batch)
    return {k: stats(out[k][0], torch.from_numpy(ref[k])[0]) for k in ('rgb_map', 'shade_map', 'norm_map', 'surf_map')}


def novel_ground_floor(emu):
    """the README command's frame (vis_novel_light + vis_ground_shading): main and the probes, blended layers"""
    ref = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'frame_novel_ground.npz')))
    kw = dict(vis_ground_shading=True, ground_normal=[float(v) for v in ref['ground_normal']], ground_origin=[float(v) for v in ref['ground_origin']],
              render_chunk_size=int(ref['render_chunk_size']))
    cfg = make_cfg('novel_light', **kw)
    net = O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg), cfg, emulate=emu, kernel_like=True)
    H = int(ref['H'])
    batch = synthetic.make_batch(H, H, seed=0, posed=True, crop=int(ref['crop']), n_novel_lights=2, skin_noise=float(ref['skin_noise']))
    m = batch.mask_at_box.reshape(1, -1)
    inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]
    out = O.render_novel_light(net, batch, ground_inds=inds)
    return {f'{n}.rgb_map': stats(out[n].rgb_map[0], torch.from_numpy(ref[f'{n}.rgb_map'])[0]) for n in ('main', 'probe00', 'probe01')}


def novel_floor(emu):
    """frame_novel.npz (vis_novel_light, three probes, no ground): per-probe rgb of the emulated oracle vs the reference's"""
    ref = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'frame_novel.npz')))
    cfg = make_cfg('novel_light')
    net = O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg), cfg, emulate=emu, kernel_like=True)
    H = int(ref['H'])
    batch = synthetic.make_batch(H, H, seed=0, posed=True, crop=int(ref['crop']), n_novel_lights=3)
    out = O.render_novel_light(net, batch)
    return {f'{n}.rgb_map': stats(out[n].rgb_map[0], torch.from_numpy(ref[f'{n}.rgb_map'])[0]) for n in ('main', 'probe00', 'probe01', 'probe02')}


def ground_floor(emu):
    """frame_ground.npz (relight + ground-plane pass, blended layers)"""
    ref = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'frame_ground.npz')))
    kw = dict(vis_ground_shading=True, ground_normal=[float(v) for v in ref['ground_normal']], ground_origin=[float(v) for v in ref['ground_origin']],
              render_chunk_size=int(ref['render_chunk_size']))
    cfg = make_cfg('relight', **kw)
    net = O.OracleNet(synthetic.make_state_dict(0, relight=True, cfg=cfg), cfg, emulate=emu, kernel_like=True)
    H = int(ref['H'])
    batch = synthetic.make_batch(H, H, seed=0, posed=True, crop=int(ref['crop']))
    m = batch.mask_at_box.reshape(1, -1)
    inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]
    out = O.render_sphere_tracing(net, batch, ground_inds=inds)
    return {k: stats(out[k][0], torch.from_numpy(ref[k])[0]) for k in ('rgb_map', 'shade_map')}


def multi_chunk_floor(emu):
    """the multi-chunk case of tests/test_gpu_parity.py (128 x 128, crop 8, three render chunks of 24 rays): no golden from the
    reference exists for it, so the floor is emulated oracle vs fp32 oracle (the same comparand the GPU test uses)"""
    cfg = make_cfg('relight', render_chunk_size=24)
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    mk = lambda: synthetic.make_batch(128, 128, seed=0, posed=True, crop=8)
    ref = O.render_sphere_tracing(O.OracleNet(sd, cfg), mk())
    out = O.render_sphere_tracing(O.OracleNet(sd, cfg, emulate=emu, kernel_like=True), mk())
    return {k: stats(out[k][0], ref[k][0]) for k in ('rgb_map', 'shade_map')}


def other_pose_floor(emu):
    """tests/test_gpu_parity.py::test_anisdf_sphere_tracing_vs_oracle_other_pose (identity pose, seed 3, AniSDF colour net through the
    sphere-tracing renderer): emulated oracle vs fp32 oracle, the comparand of that test"""
    cfg = make_cfg('sphere_tracing')
    sd = synthetic.make_state_dict(0, relight=False, cfg=cfg)
    mk = lambda: synthetic.make_batch(96, 96, seed=3, posed=False, crop=16)
    ref = O.render_sphere_tracing(O.OracleNet(sd, cfg), mk())
    out = O.render_sphere_tracing(O.OracleNet(sd, cfg, emulate=emu, kernel_like=True), mk())
    return {k: stats(out[k][0], ref[k][0]) for k in ('rgb_map', 'norm_map')}


def full_size_sample_floor(emu):
    """tests/test_gpu_parity.py::test_full_size_sample_meets_the_contract: every ~40th in-box ray of BASELINE's 512 x 512 relit frame on
    the smooth-skinning body; emulated oracle vs fp32 oracle (the comparand of that test).  Also which rays carry the error."""
    cfg = make_cfg('relight')
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    mk = lambda: synthetic.sample_rays(synthetic.make_batch(512, 512, seed=0, posed=True, skin_noise=0.0), 1024)[0]
    ref = O.render_sphere_tracing(O.OracleNet(sd, cfg), mk())
    out = O.render_sphere_tracing(O.OracleNet(sd, cfg, emulate=emu, kernel_like=True), mk())
    res = {k: stats(out[k][0], ref[k][0]) for k in ('rgb_map', 'shade_map', 'norm_map', 'surf_map')}
    e = (out.rgb_map[0] - ref.rgb_map[0]).abs().amax(-1)
    res['rgb_map']['n_rays'] = int(e.numel())
    res['rgb_map']['n_rays_over_1e2'] = int((e > 1e-2).sum())
    res['rgb_map']['psnr_trim1pct'] = round(O.psnr(out.rgb_map[0][e <= e.kthvalue(int(round(0.99 * e.numel()))).values],
                                                   ref.rgb_map[0][e <= e.kthvalue(int(round(0.99 * e.numel()))).values]), 2)
    return res


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    res = {'_about': 'emulated 16-bit-operand oracle vs the reference fp32 goldens; written by tools/precision_floor.py'}
    only = sys.argv[1:]
    path = os.path.join(ROOT, 'tests', 'golden', 'precision_floor.json')
    if only and os.path.exists(path):
        res = json.load(open(path))           # partial update: python tools/precision_floor.py novel ground multi_chunk
    for name, fn in (('frame_novel.npz', novel_floor), ('frame_ground.npz', ground_floor), ('multi_chunk', multi_chunk_floor),
                     ('other_pose', other_pose_floor), ('full_size_sample', full_size_sample_floor)):
        if only and name.replace('frame_', '').replace('.npz', '') not in only:
            continue
        for emu in ('f16', 'bf16'):
            t0 = time.time()
            res[f'{name}:{emu}'] = fn(emu)
            print(name, emu, f'{time.time() - t0:.0f} s', json.dumps(res[f'{name}:{emu}']), flush=True)
    if only:
        with open(path, 'w') as f:
            json.dump(res, f, indent=1, sort_keys=True)
        return
    for emu in ('f16', 'bf16'):
        res[f'frame_novel_ground.npz:{emu}'] = novel_ground_floor(emu)
        print('frame_novel_ground.npz', emu, json.dumps(res[f'frame_novel_ground.npz:{emu}']['main.rgb_map']), flush=True)
    for fname, sn in FRAMES.items():
        for emu in ('f16', 'bf16'):
            t0 = time.time()
            res[f'{fname}:{emu}'] = frame_floor(fname, sn, emu)
            print(fname, emu, f'{time.time() - t0:.0f} s', json.dumps(res[f'{fname}:{emu}']['rgb_map']), flush=True)
    with open(os.path.join(ROOT, 'tests', 'golden', 'precision_floor.json'), 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
