"""debug aid: d sdf / d bpts of the full query (ra_debug_full) against the oracle's autograd"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ra_oracle as O
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
cfg = make_cfg('relight')
dev = torch.device('cuda:0')
kind = sys.argv[2] if len(sys.argv) > 2 else 'init'
sd = synthetic.make_state_dict(0, relight=True, cfg=cfg, kind=kind)
net = make_network(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
body = synthetic.make_body(0, posed=True)
eng = net.set_frame(synthetic.to_device(body, dev))
g = torch.Generator().manual_seed(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
bpts = d * (0.38 + 0.12 * torch.rand(n, 1, generator=g))
grad, sdf, feat, raw = eng.debug_full(bpts.to(dev))
on = O.OracleNet(sd, cfg)
fr = O._frame(body)
b = bpts.clone().requires_grad_(True)
with torch.enable_grad():
    cp = b + on.residuals(b, fr.cond)
    s, f = on.sdf_feat(cp)
    og = torch.autograd.grad(s.sum(), b)[0]
cpd = cp.detach().clone().requires_grad_(True)
with torch.enable_grad():
    s2, _ = on.sdf_feat(cpd)
    gc = torch.autograd.grad(s2.sum(), cpd)[0]
grad = grad.cpu()
print('sdf err max %.2e' % float((sdf.cpu() - s.detach()[:, 0]).abs().max()), 'feat err max %.2e' % float((feat.cpu() - f.detach()).abs().max()))
print('nan', int(grad.isnan().any(-1).sum()), 'of', n)
e = (grad - og).abs().nan_to_num(9.9)
print('grad vs autograd(bpts): max %.3e mean %.3e' % (float(e.max()), float(e.mean())))
e2 = (grad - gc).abs().nan_to_num(9.9)
print('grad vs autograd(cpts): max %.3e mean %.3e' % (float(e2.max()), float(e2.mean())))
nrm = lambda v: v / v.norm(dim=-1, keepdim=True)
en = (nrm(grad) - nrm(og)).abs().nan_to_num(9.9)
print(kind, 'unit normal (big-pose) err: max %.3e mean %.3e p99 %.3e; |grad| mean %.3f' % (float(en.max()), float(en.mean()), float(en.flatten().kthvalue(int(0.99 * en.numel())).values), float(og.norm(dim=-1).mean())))
print('hip ', grad[:4].tolist())
print('ref ', og[:4].tolist())
print('gc  ', gc[:4].tolist())
if os.environ.get('RA_DBG_PE'):
    # oracle: gradient wrt the encoding channels of the sdf net (51) = lin0 + lin4 contributions
    import math
    pe = O.positional_encoding(cp.detach(), 8).clone().requires_grad_(True)
    xx = pe
    for l, (w, bb) in enumerate(on.sdf):
        if l == 4:
            xx = torch.cat([xx, pe], dim=-1) / math.sqrt(2)
        xx = torch.nn.functional.linear(xx, w, bb)
        if l < 8:
            xx = O.softplus100(xx)
    gpe = torch.autograd.grad(xx[:, 0].sum(), pe)[0]          # (n, 51)
    hp = feat.cpu().flatten()[:n * 128].reshape(n, 128)[:, :64].reshape(n, 2, 32)     # [h][slot]
    def chan(q, h):
        if q < 24: return 3 + 6 * (q // 3) + 3 * h + q % 3
        if q < 27: return (51 + q - 24) if h else (q - 24)
        return -1
    worst = []
    for hh in (0, 1):
        for q in range(32):
            c = chan(q, hh)
            ref_v = gpe[:, c] if 0 <= c < 51 else torch.zeros(n)
            e = (hp[:, hh, q] - ref_v).abs()
            worst.append((float(e.max()), hh, q, c, float(hp[:, hh, q].abs().max()), float(ref_v.abs().max())))
    for w in sorted(worst, reverse=True)[:12]:
        print('slot err %.3e  h %d q %2d chan %2d  |hip| %.3e |ref| %.3e' % w)
ee = (grad - og).abs().nan_to_num(99.0).amax(-1)
per = ee[: (n // 32) * 32].reshape(-1, 32)
print('per 32-point group (wave) max err:', ['%.1e' % float(v) for v in per.amax(-1)[:40]])
print('bad lanes in group of first bad wave:', [int(i) for i in (per[per.amax(-1) > 1e-2][:1] > 1e-2).nonzero()[:, 1]] if bool((per.amax(-1) > 1e-2).any()) else [])
e3 = (grad - (gc if os.environ.get('RA_DBG_GC') else og)).abs().nan_to_num(99.0)
print('lane errs wave0:', ['%.0e' % float(v) for v in e3[:32].amax(-1)])
print('lane errs wave1:', ['%.0e' % float(v) for v in e3[32:64].amax(-1)])
print('comp errs first 8 pts:', e3[:8].tolist())
if os.environ.get('RA_DBG_LAYER'):
    import math
    L = int(os.environ['RA_DBG_LAYER'])
    pe = O.positional_encoding(cp.detach(), 8)
    xx = pe
    zs = []
    for l, (w, bb) in enumerate(on.sdf):
        if l == 4:
            xx = torch.cat([xx, pe], dim=-1) / math.sqrt(2)
        z = torch.nn.functional.linear(xx, w, bb)
        if l < 8:
            z.requires_grad_(True); z.retain_grad(); zs.append(z)
            xx = O.softplus100(z)
        else:
            xx = z
    xx[:, 0].sum().backward()
    ref_d = zs[L].grad                                  # (n, 256 or 205)
    hd = feat.cpu()[:, :ref_d.shape[1]]
    e = (hd - ref_d).abs()
    valid = min(ref_d.shape[1], 224)
    print('delta_%d: max err (features < %d) %.3e, ref max %.3e' % (L, valid, float(e[:, :valid].max()), float(ref_d.abs().max())))
    print('per 32-feature block max err:', ['%.1e' % float(e[:, 32 * b:32 * b + 32].max()) for b in range(ref_d.shape[1] // 32)])
    bad = e[:, :valid].amax(-1)
    print('points with err > 1e-2:', int((bad > 1e-2).sum()), 'of', n, ' lanes of wave 0:', [int(i) for i in (bad[:32] > 1e-2).nonzero()[:, 0]])
if os.environ.get('RA_DBG_PESPLIT'):
    import math
    which = int(os.environ['RA_DBG_LAYER'])      # -2: lin4's encoding columns, -3: lin0
    pe0 = O.positional_encoding(cp.detach(), 8).clone().requires_grad_(True)
    pe4 = O.positional_encoding(cp.detach(), 8).clone().requires_grad_(True)
    xx = pe0
    for l, (w, bb) in enumerate(on.sdf):
        if l == 4:
            xx = torch.cat([xx, pe4], dim=-1) / math.sqrt(2)
        xx = torch.nn.functional.linear(xx, w, bb)
        if l < 8:
            xx = O.softplus100(xx)
    g0, g4 = torch.autograd.grad(xx[:, 0].sum(), [pe0, pe4])
    gref = g4 if which == -2 else g0
    hp = feat.cpu().flatten()[:n * 128].reshape(n, 128)[:, :64].reshape(n, 2, 32)
    rows = []
    for hh in (0, 1):
        for q in range(32):
            c = chan(q, hh)
            ref_v = gref[:, c] if 0 <= c < 51 else torch.zeros(n)
            rows.append((float((hp[:, hh, q] - ref_v).abs().max()), hh, q, c, float(hp[:, hh, q].abs().max()), float(ref_v.abs().max())))
    print('split', which, 'worst:')
    for w in sorted(rows, reverse=True)[:6]:
        print('  slot err %.3e  h %d q %2d chan %2d  |hip| %.3e |ref| %.3e' % w)
    print('  best:', ['%.1e' % r[0] for r in sorted(rows)[:8]])
if os.environ.get('RA_DBG_PESPLIT'):
    # hypothesis: the first encoding block of lin4 multiplied the ring slot's OLD content (stage - 8 = lin4^T row block 0)
    pe = O.positional_encoding(cp.detach(), 8)
    xx = pe; zs = []
    for l, (w, bb) in enumerate(on.sdf):
        if l == 4:
            xx = torch.cat([xx, pe], dim=-1) / math.sqrt(2)
        z = torch.nn.functional.linear(xx, w, bb)
        if l < 8:
            z.requires_grad_(True); z.retain_grad(); zs.append(z); xx = O.softplus100(z)
        else:
            xx = z
    xx[:, 0].sum().backward()
    d4 = zs[4].grad                                        # (n,256)
    W4 = on.sdf[4][0] / math.sqrt(2)
    for name, rows in (('lin4^T rb0', W4[:, 0:32]), ('lin4^T rb1', W4[:, 32:64]), ('lin4^T rb7', W4[:, 224:256]), ('lin5^T rb0', on.sdf[5][0][:, 0:32])):
        u = (zs[5].grad if name.startswith('lin5') else d4) @ rows            # (n,32): rows of the transposed product
        best = 0
        for hh in (0, 1):
            for q in range(16):
                r = 8 * (q // 4) + 4 * hh + q % 4
                best = max(best, float((hp[:, hh, q] - u[:, r]).abs().max()))
        print('  vs', name, ': max diff %.3e' % best)
if os.environ.get('RA_DBG_PESPLIT'):
    obs = torch.zeros(n, 32)
    for hh in (0, 1):
        for q in range(16):
            obs[:, 8 * (q // 4) + 4 * hh + q % 4] = hp[:, hh, q]
    print('observed block-0 rows, point 0:', ['%.2f' % v for v in obs[0].tolist()])
    cands = []
    for l in range(1, 8):
        W = on.sdf[l][0] / (math.sqrt(2) if l == 4 else 1.0)          # [out][in]
        for dl in range(8):
            dd = zs[dl].grad
            if dd.shape[1] != W.shape[0]:
                continue
            for rb in range(W.shape[1] // 32):
                u = dd @ W[:, 32 * rb:32 * rb + 32]
                cands.append((float((obs - u).abs().max()), 'lin%d^T rb%d x delta_%d' % (l, rb, dl)))
    print('closest candidates:', sorted(cands)[:4])
if os.environ.get('RA_DBG_PESPLIT'):
    # hypothesis: some k-steps of the first encoding block multiplied fragments of ANOTHER stage: partial products over
    # 64-feature groups (4 k-steps) of delta against every transposed hidden block
    which = int(os.environ['RA_DBG_LAYER'])
    dcur = zs[4].grad if which == -2 else zs[0].grad
    cands = []
    for l in range(1, 8):
        W = on.sdf[l][0] / (math.sqrt(2) if l == 4 else 1.0)
        if W.shape[0] != dcur.shape[1]:
            continue
        for rb in range(W.shape[1] // 32):
            for g0 in range(0, 256, 64):
                u = dcur[:, g0:g0 + 64] @ W[g0:g0 + 64, 32 * rb:32 * rb + 32]
                cands.append((float((obs - u).abs().max()), 'lin%d^T rb%d, delta features %d..%d' % (l, rb, g0, g0 + 63)))
    print('closest partial candidates:', sorted(cands)[:3])
