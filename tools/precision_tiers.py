#!/usr/bin/env python3
"""Which part of the rgb error of a 16-bit-operand frame is the arithmetic's, and which is the frame's own instability?
(CPU, oracle only — test infrastructure; round-3 verdict, "next round" item 1b.)

For the golden frame of the SURVEY 8d body (`frame_relight`, skinning noise 2.0), the strided sample of the benchmarked 512 x 512 frame
(bench.py's `psnr_vs_oracle` rays) and the full-size sample on the smooth body, the fp32 oracle (= the reference's arithmetic, pinned to
the reference's own outputs) is compared with

  f64acc   every nn.Linear evaluated in float64, rounded once to fp32: a differently associated (more accurate) fp32.  Whatever this
           changes, fp32 itself does not pin — the reference's own result on such a ray depends on its BLAS' summation order;
  tier     the tiered path: surface trace + surface samples / normals / material in the compensated arithmetic (f16 hi + lo
           operands, three products: `f16x2`), the 512 shadow rays per hit pixel with plain f16 operands (kernel-like);
  tier_k4  the same with plain f16 operands also for the surface samples / normals / material (only the surface TRACE compensated);
  f16      everything with plain f16 operands (round 3's floor).

    python tools/precision_tiers.py [frame_relight|bench_sample|bench_sample_smooth|full_size_sample ...]   # tests/golden/precision_tiers.json
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


def stats(a, b):
    a, b = a.float().reshape(-1, a.shape[-1]), b.float().reshape(-1, b.shape[-1])
    e = (a - b).abs()
    per = e.amax(-1)
    keep = per <= per.kthvalue(max(1, int(round(0.99 * per.numel())))).values
    return dict(psnr=round(O.psnr(a, b), 2), psnr_best99=round(O.psnr(a[keep], b[keep]), 2), max_abs=float(e.max()), n_rays=int(per.numel()),
                rays_over_1e2=int((per > 1e-2).sum()), rays_over_1e3=int((per > 1e-3).sum()), mean_abs=float(e.mean()))


def nets(sd, cfg):
    f16 = O.OracleNet(sd, cfg, emulate='f16', kernel_like=True)
    tier = O.OracleNet(sd, cfg, emulate='f16x2', kernel_like=True)
    tier.shadow_net = f16
    tier_k4 = _TraceOnly(sd, cfg, f16)
    return {'f64acc': O.OracleNet(sd, cfg, emulate='f64acc'), 'tier': tier, 'tier_k4': tier_k4, 'f16': f16}


class _TraceOnly(O.OracleNet):
    """compensated arithmetic for the distance queries of the surface trace only: the full query (network_forward: has autograd
    enabled) and the shadow rays run with plain f16 operands"""
    def __init__(self, sd, cfg, f16):
        super().__init__(sd, cfg, emulate='f16x2', kernel_like=True)
        self.shadow_net = f16

    def lin(self, x, w, b, **kw):
        if torch.is_grad_enabled():               # forward_geometry runs under enable_grad, the trace under no_grad
            return self.shadow_net.lin(x, w, b, **kw)
        return super().lin(x, w, b, **kw)


CASES = {
    'frame_relight': lambda: synthetic.make_batch(128, 128, seed=0, posed=True, crop=16, skin_noise=2.0),
    'bench_sample': lambda: synthetic.sample_rays(synthetic.make_batch(512, 512, seed=0, posed=True, skin_noise=2.0), 512)[0],
    'bench_sample_smooth': lambda: synthetic.sample_rays(synthetic.make_batch(512, 512, seed=0, posed=True, skin_noise=0.0), 512)[0],
    'full_size_sample': lambda: synthetic.sample_rays(synthetic.make_batch(512, 512, seed=0, posed=True, skin_noise=0.0), 1024)[0],
}


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    path = os.path.join(ROOT, 'tests', 'golden', 'precision_tiers.json')
    res = json.load(open(path)) if os.path.exists(path) else {}
    res['_about'] = 'oracle variants vs the fp32 oracle on whole frames; written by tools/precision_tiers.py'
    cfg = make_cfg('relight')
    sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
    for name in (sys.argv[1:] or list(CASES)):
        with torch.no_grad():
            ref = O.render_sphere_tracing(O.OracleNet(sd, cfg), CASES[name]())
            for vname, net in nets(sd, cfg).items():
                t0 = time.time()
                out = O.render_sphere_tracing(net, CASES[name]())
                r = {k: stats(out[k][0], ref[k][0]) for k in ('rgb_map', 'surf_map', 'norm_map')}
                r['hit_mask_flips'] = int(((out.acc_map > 0) != (ref.acc_map > 0)).sum())
                res[f'{name}:{vname}'] = r
                print(name, vname, f'{time.time() - t0:.0f} s', json.dumps(r['rgb_map']), 'surf max', f"{r['surf_map']['max_abs']:.2e}", 'flips', r['hit_mask_flips'], flush=True)
        with open(path, 'w') as f:
            json.dump(res, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
