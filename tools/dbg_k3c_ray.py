#!/usr/bin/env python3
"""Where does the HIP distance field (compensated tier) differ from the fp32 oracle along the rays of frame_relight_smooth?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ra_oracle as O
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
dev = torch.device('cuda:0')
torch.set_num_threads(16)
ref = dict(np.load(os.path.join(ROOT, 'tests/golden/frame_relight_smooth.npz')))
cfg = make_cfg('relight', vis_specular_map=True, trace_precision=2)
sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
net = make_network(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
mk = lambda: synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']), skin_noise=0.0)
b = mk()
out = make_renderer(cfg, net).render(synthetic.to_device(mk(), dev))
es = (out.surf_map.cpu() - torch.from_numpy(ref['surf_map'])).abs().amax(-1)[0]
w = int(es.argmax())
print('worst ray', w, 'surf err', float(es[w]), 'rays with surf err > 1e-4:', int((es > 1e-4).sum()))
fr = O._frame(b)
onet = O.OracleNet(sd, cfg)
eng = net.set_frame(synthetic.to_device(b, dev))
o, d, n, f = b.ray_o[0], b.ray_d[0], b.near[0], b.far[0]
# dense sampling along every ray
T = 400
tt = n[:, None] + (f - n)[:, None] * torch.linspace(0, 1, T)[None]
x = (o[:, None] + tt[..., None] * d[:, None]).reshape(-1, 3)
with torch.no_grad():
    r = O.hdq_sdf(onet, x, fr, cfg.dist_th, True)[:, 0]
h = eng.hdq_sdf(x.to(dev), cfg.dist_th, True).cpu()
e = (h - r).abs().reshape(-1, T)
print('field along all rays: max |HIP - oracle| %.2e, rms %.2e; worst ray of the field %d (max %.2e)' % (e.max(), e.pow(2).mean().sqrt(), int(e.amax(1).argmax()), e.amax(1).max()))
print('field along the worst surf ray: max %.2e at sample %d' % (e[w].max(), int(e[w].argmax())))
big = (e > 1e-5).nonzero()
print('samples with |diff| > 1e-5:', big.shape[0])
for i in range(min(10, big.shape[0])):
    ri, si = int(big[i, 0]), int(big[i, 1])
    print('  ray', ri, 'sample', si, 'hip', float(h.reshape(-1, T)[ri, si]), 'oracle', float(r.reshape(-1, T)[ri, si]))
# the trace itself, iteration by iteration, on the worst ray: oracle state machine fed by either field
def trace(fn):
    hist = []
    def f2(p):
        v = fn(p)
        hist.append(v.clone())
        return v
    res = O.sphere_tracing(o[w:w+1], d[w:w+1], n[w:w+1, None], f[w:w+1, None], f2, iter=16, relax=0.0, offset=0.02, eps=1e-8, shadow_skip_iter=1, soft_shadow=False)
    return res, hist
with torch.no_grad():
    (sa, _, _, sta, _), ha = trace(lambda p: O.hdq_sdf(onet, p, fr, cfg.dist_th, True))
    (sb, _, _, stb, _), hb = trace(lambda p: eng.hdq_sdf(p.to(dev), cfg.dist_th, True).cpu()[:, None])
print('oracle st', float(sta), 'with HIP field st', float(stb))
for i, (a, bb) in enumerate(zip(ha, hb)):
    print(i, float(a), float(bb), float(a - bb))
