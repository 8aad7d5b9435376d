#!/usr/bin/env python3
"""Stage-by-stage error report of the HIP path against the golden fixtures (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
dev = torch.device('cuda:0')


def stat(name, a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    same = (a == b) | (a.isnan() & b.isnan())
    e = torch.where(same, torch.zeros_like(a), (a - b).abs())
    print(f'{name:28s} max {e.max().item():.3e}  mean {e.mean().item():.3e}  p99 {e.flatten().kthvalue(max(1, int(0.99 * e.numel())))[0].item():.3e}  ref_absmax {b[~b.isnan()].abs().max().item():.3e}')


def psnr(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float(-10 * torch.log10(torch.mean((a - b) ** 2)))


DT = os.environ.get('RA_DTYPE', 'f16')
print('mlp dtype', DT)


def build(mode, **kw):
    cfg = make_cfg(mode, mlp_dtype=DT, **kw)
    relight = mode in ('relight', 'novel_light')
    net = make_network(cfg)
    net.load_state_dict(synthetic.make_state_dict(0, relight=relight, cfg=cfg))
    net = net.to(dev).eval()
    return cfg, net


ops = dict(np.load(os.path.join(G, 'ops.npz')))
T = lambda k: torch.from_numpy(ops[k])
cfg, net = build('relight')
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
print('== MLP stage')
resd, sdf, feat = eng.debug_mlp(T('mlp_bpts').to(dev))
torch.cuda.synchronize()
stat('resd', resd, T('mlp_resd'))
stat('sdf', sdf[:, None], T('mlp_sdf'))
stat('feat', feat, T('mlp_feat'))
print('== coarse')
x = T('hdq_x').to(dev)
o = eng.debug_hdq(x, 0.125)
print('fine count', o.fine_count, 'ref', int(T('knn_fine').sum()))
stat('sdf_batch', o.sdf_batch, T('knn_sdf_batch'))
print('nn mismatch', int((o.nn_batch.cpu().long() != T('knn_nn_batch')).sum()))
m = T('knn_fine')
stat('bpts', o.bpts.cpu()[m], T('warp_bpts')[m])
stat('tpts', o.tpts.cpu()[m], T('warp_tpts')[m])
stat('A_bw', o.mats.cpu()[m][:, :12], T('warp_A_bw')[m][:, :3, :].reshape(-1, 12))
stat('big_A_bw', o.mats.cpu()[m][:, 12:], T('warp_big_A_bw')[m][:, :3, :].reshape(-1, 12))
eng.set_knn_mode(False); eng.set_frame(body, force=True)
ob = eng.debug_hdq(x, 0.125)
eng.set_knn_mode(True); eng.set_frame(body, force=True)
print('bvh vs brute: nn mismatch', int((ob.nn_batch != o.nn_batch).sum()), 'd2 maxdiff', float((ob.d2 - o.d2).abs().max()), 'bpts maxdiff', float((ob.bpts - o.bpts).abs().max()))
xr = (torch.rand(200000, 3, device=dev) - 0.5) * 3.0
o1 = eng.debug_hdq(xr, 0.125)
eng.set_knn_mode(False); eng.set_frame(body, force=True)
o2 = eng.debug_hdq(xr, 0.125)
eng.set_knn_mode(True); eng.set_frame(body, force=True)
print('bvh vs brute (200k pts in a 3 m cube): nn mismatch', int((o1.nn_batch != o2.nn_batch).sum()), 'sdf maxdiff', float((o1.sdf_coarse - o2.sdf_coarse).abs().max()), 'fine', o1.fine_count, o2.fine_count)
print('== hdq')
s = net.inference_world_distance_field(x[None], body, smooth_transition=True, dist_th=0.125)
stat('hdq_sdf', s[0], T('hdq_sdf'))
s = net.inference_world_distance_field(x[None], body, smooth_transition=False, dist_th=0.125)
stat('hdq_sdf_nosmooth', s[0], T('hdq_sdf_nosmooth'))
print('== full kernel (identity warp) grad/feat')
g, sd_, ft, raw = eng.debug_full(T('mlp_bpts').to(dev))
stat('full sdf', sd_[:, None], T('mlp_sdf'))
stat('full feat', ft, T('mlp_feat'))
stat('full albedo', raw[:, 9:12], T('mlp_albedo'))
stat('full rough', raw[:, 12:13], T('mlp_rough'))
stat('full occ', raw[:, 16:17], T('mlp_occ'))
print('== forward raw')
raw = net(T('fwd_x')[None].to(dev), None, 0.005, body).raw[0]
ref = T('fwd_raw')
for nm, sl in (('cpts', slice(0, 3)), ('bpts', slice(3, 6)), ('resd', slice(6, 9)), ('albedo', slice(9, 12)), ('rough', slice(12, 13)), ('norm', slice(13, 16)), ('occ', slice(16, 17))):
    stat('raw ' + nm, raw[:, sl], ref[:, sl])
print('== sphere trace')
p = eng.trace_params(cfg.sphere_tracing, cfg.dist_th, False)
surf, occ, st, ot = eng.sphere_trace(T('st_o').to(dev), T('st_d').to(dev), T('st_near').to(dev), T('st_far').to(dev), p)
stat('st', st[:, None], T('st_st'))
stat('occ', occ[:, None], T('st_occ'))
stat('surf', surf, T('st_surf'))
print('hit flips', int(((occ.cpu() < 1) != (T('st_occ')[:, 0] < 1)).sum()), 'of', occ.numel())
p = eng.trace_params(cfg.obj_lvis, 0.125, True)
n = T('sh_o').shape[0]
surf, occ, st, ot = eng.sphere_trace(T('sh_o').to(dev), T('sh_d').to(dev), torch.full((n,), 0.02, device=dev), torch.full((n,), 0.8, device=dev), p, tan_i=T('sh_tan_i').to(dev))
stat('shadow occ', occ[:, None], T('sh_occ'))

for mode, fname, keys in (('sphere_tracing', 'frame_sphere.npz', ('acc_map', 'surf_map', 'norm_map', 'rgb_map', 'depth_map', 'cpts_map', 'resd_map')),
                          ('relight', 'frame_relight.npz', ('acc_map', 'surf_map', 'norm_map', 'albedo_map', 'roughness_map', 'shade_map', 'spec_map', 'rgb_map')),
                          ('anisdf', 'frame_anisdf.npz', ('acc_map', 'depth_map', 'norm_map', 'rgb_map', 'cpts_map'))):
    print('== frame', mode)
    ref = dict(np.load(os.path.join(G, fname)))
    kw = {}
    if mode == 'relight': kw['vis_specular_map'] = True
    if mode == 'anisdf': kw['n_samples'] = int(ref['n_samples'])
    cfg, net = build(mode, **kw)
    rend = make_renderer(cfg, net)
    batch = synthetic.to_device(synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop'])), dev)
    t0 = time.time()
    out = rend.render(batch)
    torch.cuda.synchronize()
    print('render time', time.time() - t0)
    for k in keys:
        stat(k, out[k], ref[k])
    print('rgb PSNR', psnr(out.rgb_map, ref['rgb_map']), 'counters', dict(net.engine().counters()))

print('== frame novel')
ref = dict(np.load(os.path.join(G, 'frame_novel.npz')))
cfg, net = build('novel_light')
rend = make_renderer(cfg, net)
batch = synthetic.to_device(synthetic.make_batch(int(ref['H']), int(ref['H']), seed=0, posed=True, crop=int(ref['crop']), n_novel_lights=3), dev)
out = rend.render(batch)
torch.cuda.synchronize()
stat('main rgb', out.main.rgb_map, ref['main.rgb_map'])
for n in batch.novel_lights:
    for k in ('rgb_map', 'shade_map', 'spec_map'):
        stat(f'{n}.{k}', out[n][k], ref[f'{n}.{k}'])
    print(n, 'rgb PSNR', psnr(out[n].rgb_map, ref[f'{n}.rgb_map']))
