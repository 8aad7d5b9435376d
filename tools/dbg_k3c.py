#!/usr/bin/env python3
"""First-light checks of K3C (the compensated distance query, csrc/ra_k3c.hpp) on the GPU: stage accuracy against the fp32 / float64
oracle, bit-identity of its two workgroup widths, whole frames with cfg.trace_precision 0 / 1 / 2 against the goldens and the oracle."""
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
from relightableavatar_amd.networks import make_network  # noqa: E402
from relightableavatar_amd.renderer import make_renderer  # noqa: E402

dev = torch.device('cuda:0')


def build(mode, **kw):
    cfg = make_cfg(mode, **kw)
    net = make_network(cfg)
    net.load_state_dict(synthetic.make_state_dict(0, relight=mode in ('relight', 'novel_light'), cfg=cfg))
    return cfg, net.to(dev).eval()


def stage():
    g = torch.Generator().manual_seed(11)
    d = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
    bpts = d * (0.38 + 0.12 * torch.rand(20000, 1, generator=g))
    body = synthetic.make_body(0, posed=True)
    fr = O._frame(body)
    cfg = make_cfg('relight')
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    f32 = O.observed_sdf(O.OracleNet(sd, cfg), bpts, fr)[:, 0]
    f64 = O.observed_sdf(O.OracleNet(sd, cfg, emulate='f64acc'), bpts, fr)[:, 0]
    emu = O.observed_sdf(O.OracleNet(sd, cfg, emulate='f16x2', kernel_like=True), bpts, fr)[:, 0]
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt())
    mx = lambda a, b: float((a - b).abs().max())
    print(f'oracle: fp32 vs f64acc rms {rms(f32, f64):.2e} max {mx(f32, f64):.2e}; f16x2 emulation vs f64acc rms {rms(emu, f64):.2e}')
    for tp in (0, 2):
        cfg, net = build('relight', trace_precision=tp)
        eng = net.set_frame(synthetic.to_device(body, dev))
        hip = eng.observed_sdf(bpts.to(dev)).cpu()
        print(f'trace_precision {tp}: HIP vs f64acc rms {rms(hip, f64):.2e} max {mx(hip, f64):.2e}; vs fp32 rms {rms(hip, f32):.2e} max {mx(hip, f32):.2e}')
        if tp == 2:
            a = eng.observed_sdf(bpts[:9000].to(dev)).cpu()          # 2-wave launch
            print('  widths bit-identical (2 vs 4 waves):', bool(torch.equal(a, hip[:9000])))
            x = bpts * 1.02
            t0 = time.time()
            h = eng.hdq_sdf(x.to(dev), 0.125, True).cpu()
            ref = O.hdq_sdf(O.OracleNet(sd, cfg), x, fr, 0.125, True)[:, 0]
            print(f'  hdq_sdf vs fp32 oracle rms {rms(h, ref):.2e} max {mx(h, ref):.2e}')


def frames():
    def psnr(a, b):
        return float(-10 * torch.log10(((a.float().cpu() - torch.as_tensor(b).float()) ** 2).mean()))
    for name, sn in (('frame_relight.npz', 2.0), ('frame_relight_smooth.npz', 0.0)):
        ref = dict(np.load(os.path.join(ROOT, 'tests', 'golden', name)))
        for tp in (0, 1, 2):
            cfg, net = build('relight', vis_specular_map=True, trace_precision=tp)
            batch = synthetic.to_device(synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']), skin_noise=sn), dev)
            out = make_renderer(cfg, net).render(batch)
            e = (out.rgb_map.cpu() - torch.from_numpy(ref['rgb_map'])).abs()
            es = (out.surf_map.cpu() - torch.from_numpy(ref['surf_map'])).abs()
            c = net.engine().counters()
            print(f'{name} trace_precision {tp}: rgb PSNR {psnr(out.rgb_map, ref["rgb_map"]):.2f} dB max {float(e.max()):.2e} over1e-2 {int((e.amax(-1) > 1e-2).sum())}'
                  f' surf max {float(es.max()):.2e}  fine {c.n_fine_sdf} comp {c.n_fine_sdf_comp}')
    # the bench sample (bench.py psnr_vs_oracle) against the oracle
    cfg = make_cfg('relight')
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    for sn in (2.0, 0.0):
        mk = lambda: synthetic.sample_rays(synthetic.make_batch(512, 512, seed=0, posed=True, skin_noise=sn), 512)[0]
        ref = O.render_sphere_tracing(O.OracleNet(sd, cfg), mk())
        for tp in (0, 1):
            cfg2, net = build('relight', trace_precision=tp)
            out = make_renderer(cfg2, net).render(synthetic.to_device(mk(), dev))
            e = (out.rgb_map.cpu() - ref.rgb_map).abs()
            print(f'bench sample skin_noise {sn} trace_precision {tp}: rgb PSNR {psnr(out.rgb_map, ref.rgb_map):.2f} dB max {float(e.max()):.2e} over1e-2 {int((e.amax(-1) > 1e-2).sum())} of {e.shape[1]}')


if __name__ == '__main__':
    torch.set_num_threads(16)
    stage()
    frames()
