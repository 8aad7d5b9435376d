#!/usr/bin/env python3
"""Which rays of the parity frames does the reference's OWN fp32 arithmetic not pin?  (CPU, oracle only — test infrastructure.)

The reference's fixed-iteration sphere trace ends in a limit cycle on part of the rays; its closest-approach rule
(sphere_tracing_renderer.py:194-197: `abs(d1) < cd -> st = t`) then compares the distances of successive visits of the same phase of the
cycle, values that agree to ~1e-7 — which visit wins, and with it a jump of the surface point by millimetres (0.01-0.1 in rgb), is
decided by the last bits of the fp32 sums, i.e. by the BLAS' summation order of the machine the reference runs on.  On such a ray the
"reference value" is a coin toss of the reference itself, and no implementation can be held to it.

oracle.fp32_unstable_rays marks a ray when gaussian noise of 3e-7 (the fp32 rounding level of the distance: the oracle is 1.2e-7 rms
from a float64 evaluation of its MLPs; the HIP path's compensated tier 2.0e-7) on the distances the surface trace reads moves its `st`
by more than 1e-4 or flips its hit status in any of 32 runs.  The lists go to tests/golden/fp32_unstable_rays.json; the GPU parity tests
assert SURVEY.md:409's contract (rgb PSNR >= 50 dB over ALL rays, max |err| <= 1e-2) with the max taken over the rays fp32 pins.

    python tools/fp32_stability.py [case ...] [--probs]        # about 6 minutes on 8 cores for all cases; --probs: only the flip probabilities of the listed rays
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ra_oracle as O                      # noqa: E402
from relightableavatar_amd import synthetic            # noqa: E402
from relightableavatar_amd.config import make_cfg      # noqa: E402

MB = synthetic.make_batch
# case -> (cfg mode, relight weights?, batch factory): the ray sets of tests/test_gpu_parity.py, bench.py and __graft_entry__.smoke()
CASES = {
    'frame_relight.npz': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=16)),
    'frame_relight_smooth.npz': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=16, skin_noise=0.0)),
    'frame_novel.npz': ('novel_light', lambda: MB(128, 128, seed=0, posed=True, crop=12)),
    'frame_ground.npz': ('relight', lambda: MB(24, 24, seed=0, posed=True, crop=10)),
    'frame_novel_ground.npz': ('novel_light', lambda: MB(24, 24, seed=0, posed=True, crop=10, skin_noise=0.0)),
    'frame_sphere.npz': ('sphere_tracing', lambda: MB(128, 128, seed=0, posed=True, crop=32)),
    'multi_chunk': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=8)),
    'other_pose': ('sphere_tracing', lambda: MB(96, 96, seed=3, posed=False, crop=16)),
    'smoke': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=8)),
    'full_size_sample': ('relight', lambda: synthetic.sample_rays(MB(512, 512, seed=0, posed=True, skin_noise=0.0), 1024)[0]),
    'bench_sample': ('relight', lambda: synthetic.sample_rays(MB(512, 512, seed=0, posed=True, skin_noise=2.0), 512)[0]),
    'bench_sample_smooth': ('relight', lambda: synthetic.sample_rays(MB(512, 512, seed=0, posed=True, skin_noise=0.0), 512)[0]),
    'bench_sample_3072': ('relight', lambda: synthetic.sample_rays(MB(512, 512, seed=0, posed=True, skin_noise=2.0), 3072)[0]),      # bench.py N_SAMPLE
    'bench_sample_smooth_3072': ('relight', lambda: synthetic.sample_rays(MB(512, 512, seed=0, posed=True, skin_noise=0.0), 3072)[0]),
    # tests/golden/switches.npz: one window, traced under three different settings (the other variants trace like 'base')
    'switches.npz:base': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=10, skin_noise=0.0)),
    'switches.npz:trace_params': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=10, skin_noise=0.0), 'trace_params'),
    'switches.npz:no_geodesic_filter': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=10, skin_noise=0.0), 'no_geodesic_filter'),
    'switches.npz:other_weights': ('relight', lambda: MB(128, 128, seed=3, posed=False, crop=10, skin_noise=0.0, cam_dist=1.6), 'other_weights', 7),
    'switches.npz:all_shadowed': ('relight', lambda: MB(128, 128, seed=3, posed=False, crop=10, skin_noise=0.0, cam_dist=1.6), 'all_shadowed', 5),
    'switches.npz:smpl24': ('relight', lambda: MB(128, 128, seed=0, posed=True, crop=10, skin_noise=0.0, n_bones=24, n_verts=5023), 'smpl24'),
}
# round 6, the hard cases of tests/golden/switches.npz: batch arguments and synthetic weights come from the variant's own @-keys
for _v in ('split_body', 'split_body_iter4', 'sharp_weights', 'sharp_split'):
    CASES['switches.npz:' + _v] = ('relight', None, _v)


def variant_batch_and_weights(variant):
    """(batch factory, make_state_dict kwargs) of a variant that carries @-keys"""
    import numpy as np
    v = json.loads(str(np.load(os.path.join(ROOT, 'tests', 'golden', 'switches.npz'))['variants_json']))[variant]
    bkw = {k[1:]: val for k, val in v.items() if k.startswith('@')}
    wkw = dict(seed=bkw.pop('weights_seed', 0), kind=bkw.pop('weights_kind', 'init'), env=bkw.pop('env', 'back'))
    return (lambda: MB(128, 128, **{**dict(seed=0, posed=True, crop=10, skin_noise=0.0), **bkw})), wkw


def switch_overrides(cfg, variant):
    import numpy as np
    v = json.loads(str(np.load(os.path.join(ROOT, 'tests', 'golden', 'switches.npz'))['variants_json']))[variant]
    for k, val in v.items():
        if k.startswith('@'):          # an argument of synthetic.make_batch (the case's factory passes it)
            continue
        node = cfg
        parts = k.split('.')
        for q in parts[:-1]:
            node = node[q]
        node[parts[-1]] = val


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    path = os.path.join(ROOT, 'tests', 'golden', 'fp32_unstable_rays.json')
    res = json.load(open(path)) if os.path.exists(path) else {}
    res['_about'] = ('rays whose traced surface fp32 itself does not pin: oracle.fp32_unstable_rays(trials=32, noise=3e-7, tol=1e-4, seed=0) '
                     'per ray set; written by tools/fp32_stability.py')
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    probs_only = '--probs' in sys.argv          # keep the committed lists, only (re)compute the flip probabilities of the listed rays
    for name in (args or list(CASES)):
        mode, mk = CASES[name][:2]
        cfg = make_cfg(mode)
        if len(CASES[name]) > 2:
            switch_overrides(cfg, CASES[name][2])
        relight = mode in ('relight', 'novel_light')
        wkw = dict(seed=CASES[name][3] if len(CASES[name]) > 3 else 0)
        if mk is None:
            mk, wkw = variant_batch_and_weights(CASES[name][2])
        net = O.OracleNet(synthetic.make_state_dict(relight=relight, cfg=cfg, **wkw), cfg)
        t0 = time.time()
        b = mk()
        if probs_only and name in res:
            listed = res[name]['unstable']
            assert res[name]['n_rays'] == b.ray_o.shape[1], name
        else:
            bad = O.fp32_unstable_rays(net, b)
            listed = [int(i) for i in bad.nonzero()[:, 0]]
            res[name] = dict(n_rays=int(bad.numel()), unstable=listed)
        # how often each listed ray flips: at the noise level the list is defined with (3e-7: the rounding level of the HIP path's
        # compensated tier, 2.1e-7 rms / 5.9e-7 max) and at fp32's own (1.2e-7 rms: the oracle against a float64 evaluation), 64 runs each
        res[name]['flip_probability'] = {'noise_3e-7': O.fp32_flip_probability(net, b, listed, 3e-7),
                                         'noise_1.2e-7': O.fp32_flip_probability(net, b, listed, 1.2e-7)}
        print(name, f'{time.time() - t0:.0f} s', res[name], flush=True)
        with open(path, 'w') as f:
            json.dump(res, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
