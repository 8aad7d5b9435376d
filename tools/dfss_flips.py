#!/usr/bin/env python3
"""Which discontinuity of the reference's DFSS state machine (sphere_tracing_renderer.py:157-179) do plain-f16 distance errors cross?
(CPU, oracle only — test infrastructure.)  The shadow rays of a variant of tests/golden/switches.npz are traced twice from the same
surface points — distances from the fp32 oracle and from its f16 operand-rounding emulation — and for every ray whose visibility differs
by more than 0.1 the first accept decision that differs is named.      python tools/dfss_flips.py [split_body | sharp_split | ...]"""
import sys, json, torch, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from relightableavatar_amd import synthetic
from oracle import ra_oracle as O
from test_oracle_frames import switch_cfg, switch_batch_kw, switch_state_dict
name = sys.argv[1] if len(sys.argv) > 1 else 'split_body'
ov = json.loads(str(np.load(os.path.join(ROOT, 'tests', 'golden', 'switches.npz'))['variants_json']))[name]
cfg = switch_cfg(ov); bkw = switch_batch_kw(ov)
sd = switch_state_dict(bkw, cfg)
net = O.OracleNet(sd, cfg); net16 = O.OracleNet(sd, cfg, emulate='f16', kernel_like=True)
batch = synthetic.make_batch(128, 128, **{**dict(seed=0, posed=True, crop=10, skin_noise=0.0), **bkw})
fr = O._frame(batch)
# surface points / normals from the fp32 render
rec = {}
orig_lv = O.light_visibility
def grab(n_, surf, norm, acc, fr_, bbox, lcfg, fac):
    rec.update(surf=surf, norm=norm, acc=acc, bbox=bbox, lcfg=lcfg)
    return orig_lv(n_, surf, norm, acc, fr_, bbox, lcfg, fac)
O.light_visibility = grab
O.render_sphere_tracing(net, batch)
O.light_visibility = orig_lv
surf, norm, acc, bbox, lcfg = rec['surf'], rec['norm'], rec['acc'], rec['bbox'], rec['lcfg']
xyz = net.light_xyz.reshape(-1, 3); sharp = net.light_sharp.reshape(-1)
Ld = O.normalize(xyz); ldot = Ld @ norm.T
li, pi = torch.nonzero((ldot > 0) & (acc > 0)[None], as_tuple=True)
ro, rd = surf[pi], Ld[li]
n, f = O.get_near_far_aabb(bbox, ro, rd); n, f = n.clip(lcfg['near_offset']), f.clip(lcfg['near_offset'])
box = n < f
ro, rd, n, f, li = ro[box], rd[box], n[box][:, None], f[box][:, None], li[box]
tan = 1.0 / sharp[li][:, None]
def trace(sdf_fn, log):
    P = ro.shape[0]; ones = torch.ones(P, 1)
    off = ones * lcfg['offset']; rlx = ones * lcfg['relax']; occ = ones.clone(); d0 = ones * 1e9; t = n.clone(); near = n
    for i in range(lcfg['iter']):
        d1 = sdf_fn(ro + t * rd)
        if i >= 1:
            dx0 = d0 + rlx * d0 + off; dx1 = d1 + rlx * d1 + off
            dy = dx1 ** 2 / (2 * dx0); dx = ((dx1 ** 2 - dy ** 2).sqrt() - off) / (1 + rlx)
            cls = dx.clip(0) / (t - dy).clip(near).clip(1e-8) / (tan * 2)
            conds = dict(lt=cls < occ, dy_t=dy < t, dx1=dx1 > 0, dx0=dx0 > 0, dx=dx > 0, dy0=dy > 0, dy_dx0=dy < dx0)
            msk = torch.stack(list(conds.values())).all(0)
            log.append(dict(i=i, kind='clay', conds={k: v.clone() for k, v in conds.items()}, msk=msk.clone(), cls=cls.clone(), occ_before=occ.clone()))
            occ = torch.where(msk, cls, occ)
            cls2 = d1.clip(0) / t.clip(near).clip(1e-8) / (tan * 2)
            m2 = cls2 < occ
            log.append(dict(i=i, kind='plain', msk=m2.clone(), cls=cls2.clone(), occ_before=occ.clone()))
            occ = torch.where(m2, cls2, occ)
        dt = d1 + rlx * d1 + off
        t = torch.maximum(torch.minimum(t + dt, f), near); d0 = d1
    return occ
with torch.no_grad():
    la, lb = [], []
    oa = trace(lambda x: O.hdq_sdf(net, x, fr, lcfg['dist_th'], True), la)
    ob = trace(lambda x: O.hdq_sdf(net16, x, fr, lcfg['dist_th'], True), lb)
d = (oa - ob).abs()[:, 0]
print(name, 'shadow rays', d.numel(), 'mean |docc|', float(d.mean()), 'rays with |docc| > 0.1:', int((d > 0.1).sum()), '> 0.5:', int((d > 0.5).sum()))
flip = d > 0.1
# first step at which the accept decision differs, and which condition differs there
first = {}
for a, b in zip(la, lb):
    diff = (a['msk'] != b['msk'])[:, 0] & flip
    for r in diff.nonzero()[:, 0].tolist():
        if r in first: continue
        if a['kind'] == 'clay':
            which = [k for k in a['conds'] if bool(a['conds'][k][r, 0] != b['conds'][k][r, 0])]
        else:
            which = ['plain cls<occ']
        first[r] = (a['i'], a['kind'], tuple(which), float(a['cls'][r, 0]), float(a['occ_before'][r, 0]))
import collections
c = collections.Counter((v[1], v[2]) for v in first.values())
print('first differing decision of the', len(first), 'flipped rays (of', int(flip.sum()), '):')
for k, v in c.most_common(): print('  ', k, v)
print('flipped rays without a differing decision (continuous drift):', int(flip.sum()) - len(first))
