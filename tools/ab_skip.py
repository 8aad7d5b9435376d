"""A/B: frame with and without the repeated-query skip (cfg.query_skip) must be bit-identical: ab_skip.py OUT.pt [ground] [noskip]; ab_skip.py cmp A.pt B.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
if sys.argv[1] == 'cmp':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a: print(k, 'max |diff|', float((a[k] - b[k]).abs().max()), 'equal', bool(torch.equal(a[k], b[k])))
    sys.exit(0)
dev = torch.device('cuda:0')
kw = dict(vis_ground_shading=True, ground_normal=[0.0, -1.0, 0.0], ground_origin=[0.0, 0.45, 0.0]) if 'ground' in sys.argv[2:] else {}
kw['query_skip'] = 'noskip' not in sys.argv[2:]
cfg = make_cfg('relight', **kw)
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
batch = synthetic.to_device(synthetic.make_batch(256, 256, seed=0, posed=True), dev)
out = make_renderer(cfg, net).render(batch)
torch.save({k: out[k].cpu() for k in ('rgb_map', 'acc_map', 'shade_map', 'surf_map')}, sys.argv[1])
