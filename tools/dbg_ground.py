"""ground pass on the GPU vs a golden frame made by the reference (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
ref = np.load(sys.argv[1])
dev = torch.device('cuda:0')
kw = dict(vis_ground_shading=True, ground_normal=[float(v) for v in ref['ground_normal']], ground_origin=[float(v) for v in ref['ground_origin']],
          render_chunk_size=int(ref['render_chunk_size']))
cfg = make_cfg('relight', **kw)
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
H, crop = int(ref['H']), int(ref['crop'])
batch = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True, crop=crop), dev)
rend = make_renderer(cfg, net)
m = batch.mask_at_box.reshape(1, -1).cpu()
rend.ground_inds = m.int().topk(int(m.sum()), dim=-1, sorted=False)[1][0]
out = rend.render(batch)
print('wbounds', batch.wbounds.cpu().numpy().ravel(), ref['wbounds_after'].ravel())
for k in ('acc_map', 'surf_map', 'depth_map', 'albedo_map', 'shade_map', 'spec_map', 'rgb_map', 'norm_map', 'roughness_map'):
    a, b = out[k].cpu().float(), torch.from_numpy(ref[k]).float()
    e = (a - b).abs()
    print(f'{k:14s} max {float(e.max()):.3e} mean {float(e.mean()):.3e} frac>1e-3 {float((e > 1e-3).float().mean()):.4f}  ref mean {float(b.abs().mean()):.3e}')
