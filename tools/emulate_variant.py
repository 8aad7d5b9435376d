#!/usr/bin/env python3
"""CPU emulation of the HIP path's precision tiers on a variant of tests/golden/switches.npz, against the reference's output (oracle only — test
infrastructure): which arithmetic of the shadow rays reaches the pixels.

    python tools/emulate_variant.py split_body [MODE]
MODE: none (the fp32 oracle) | f16 | f16x2 (everything in that arithmetic) | tier (surface trace + full query compensated, shadow rays plain
f16: round 5's tiers) | tier:f16w2 / tier:f16a2 (shadow rays with ONE operand side as hi + lo pairs) | key:SHARE (the shipped tiers: shadow
rays towards the key lights compensated, oracle.key_lights)."""
import os, sys, json, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from relightableavatar_amd import synthetic
from oracle import ra_oracle as O
import importlib.util
from test_oracle_frames import switch_cfg, switch_batch_kw
name = sys.argv[1]
emulate = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != 'none' else None
ov = json.loads(str(np.load(os.path.join(ROOT, 'tests', 'golden', 'switches.npz'))['variants_json']))[name]
cfg = switch_cfg(ov); bkw = switch_batch_kw(ov)
sd = synthetic.make_state_dict(bkw.pop('weights_seed', 0), relight=True, cfg=cfg, kind=bkw.pop('weights_kind', 'init'), env=bkw.pop('env', 'back'))
if emulate and emulate.startswith('key:'):
    net = O.OracleNet(sd, cfg, emulate='f16x2', kernel_like=True); net.shadow_net = O.OracleNet(sd, cfg, emulate='f16', kernel_like=True)
    net.key_lights = O.key_lights(net, [net.global_env_map], float(emulate[4:])); print('key lights', int(net.key_lights.sum()))
elif emulate and emulate.startswith('tier:'):
    net = O.OracleNet(sd, cfg, emulate='f16x2', kernel_like=True); net.shadow_net = O.OracleNet(sd, cfg, emulate=emulate[5:], kernel_like=True)
elif emulate == 'tier':
    net = O.OracleNet(sd, cfg, emulate='f16x2', kernel_like=True); net.shadow_net = O.OracleNet(sd, cfg, emulate='f16', kernel_like=True)
else:
    net = O.OracleNet(sd, cfg, emulate=emulate, kernel_like=bool(emulate))
H = 24 if name.startswith('g_') else 128
batch = synthetic.make_batch(H, H, **{**dict(seed=0, posed=True, crop=10, skin_noise=0.0), **bkw})
t = time.time()
kw = {}
if name.startswith('g_'):
    m = batch.mask_at_box.reshape(1, -1); kw['ground_inds'] = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]
out = O.render_sphere_tracing(net, batch, **kw)
print('oracle', time.time() - t, 's')
_z = np.load(os.path.join(ROOT, 'tests', 'golden', 'switches.npz'))
ref = {k[len(name) + 1:]: _z[k] for k in _z.files if k.startswith(name + '.')}
for k in ref:
    if k in out:
        a, b = out[k].numpy(), ref[k]
        e = np.abs(a - b)
        e = np.where(np.isfinite(e), e, 0)
        print(f'{k}: max {e.max():.2e} mean {e.mean():.2e}', 'psnr %.1f' % (-10 * np.log10((e ** 2).mean() + 1e-30)) if k in ('rgb_map', 'shade_map') else '')
if 'rgb_map' in ref:
    e = np.abs(out['rgb_map'].numpy() - ref['rgb_map'])[0].max(-1)
    print('worst rays', np.argsort(e)[-3:], e[np.argsort(e)[-3:]]); print('rays over 1e-2:', int((e > 1e-2).sum()), 'over 5e-3:', int((e > 5e-3).sum()), 'of', e.size, 'worst', np.round(np.sort(e)[-6:], 4))
print('rgb mean', float(ref['rgb_map'].mean()), 'hit', int((ref['acc_map'] > 0).sum()), 'shade range', float(ref['shade_map'].min()), float(ref['shade_map'].max()))
