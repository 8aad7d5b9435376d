"""debug aid: per-pixel errors of the novel-light x ground frame against the golden"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
ref = dict(np.load('tests/golden/frame_novel_ground.npz'))
kw = dict(vis_ground_shading=True, ground_normal=[float(v) for v in ref['ground_normal']], ground_origin=[float(v) for v in ref['ground_origin']], render_chunk_size=int(ref['render_chunk_size']))
cfg = make_cfg('novel_light', **kw)
dev = torch.device('cuda:0')
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
H = int(ref['H'])
batch = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True, crop=int(ref['crop']), n_novel_lights=2, skin_noise=float(ref['skin_noise'])), dev)
rend = make_renderer(cfg, net)
m = batch.mask_at_box.reshape(1, -1).cpu()
rend.ground_inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]
out = rend.render(batch)
T = torch.from_numpy
for name in ('main', 'probe00'):
    e = (out[name].rgb_map[0].cpu() - T(ref[f'{name}.rgb_map'])[0]).abs().amax(-1)
    idx = e.argsort(descending=True)[:6]
    for i in idx.tolist():
        print(name, 'pix', i, divmod(i, H), 'rgb err %.3e' % float(e[i]), 'acc', float(out[name].acc_map[0, i]), float(ref[f'{name}.acc_map'][0, i]),
              'shade err %.2e' % float((out[name].shade_map[0, i].cpu() - T(ref[f'{name}.shade_map'])[0, i]).abs().max()),
              'albedo err %.2e' % float((out[name].albedo_map[0, i].cpu() - T(ref[f'{name}.albedo_map'])[0, i]).abs().max()),
              'in-box', bool(m[0, i]))
