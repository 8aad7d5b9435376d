import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
dev = torch.device('cuda:0')
cfg = make_cfg('relight', vis_novel_light=True)
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
x = (torch.rand(300000, 3, device=dev) - 0.5) * 0.9
a = eng.hdq_sdf(x, 0.125, True); b = eng.hdq_sdf(x, 0.125, True)
print('hdq_sdf repeat maxdiff', float((a - b).abs().max()))
r1 = eng.forward(x[:50000], None, 0.125); r2 = eng.forward(x[:50000], None, 0.125)
print('forward repeat maxdiff', float((r1 - r2).abs().max()))
rend = make_renderer(cfg, net)
base = synthetic.to_device(synthetic.make_batch(256, 256, seed=0, posed=True), dev)
wb0 = base.wbounds.clone()
outs = []
for i in range(2):
    base.wbounds.copy_(wb0)
    o = rend.render(base)
    outs.append({k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)})
for k in outs[0]:
    d = (outs[0][k] - outs[1][k]).abs()
    print(f'{k:14s} maxdiff {float(d.max()):.3e}  n_diff {int((d > 0).sum())}')
print('--- mlp only')
bp = (torch.rand(100000, 3, device=dev) - 0.5) * 0.8
r1, s1, _ = eng.debug_mlp(bp, want_feat=False); r2, s2, _ = eng.debug_mlp(bp, want_feat=False)
print('debug_mlp repeat maxdiff sdf', float((s1 - s2).abs().max()), 'resd', float((r1 - r2).abs().max()))
perm = torch.randperm(bp.shape[0], device=dev)
r3, s3, _ = eng.debug_mlp(bp[perm].contiguous(), want_feat=False)
print('debug_mlp permuted maxdiff sdf', float((s1[perm] - s3).abs().max()), 'resd', float((r1[perm] - r3).abs().max()))
bad = ((s1[perm] - s3).abs() > 0).nonzero().flatten()
print('n bad', bad.numel(), 'first bad slots (permuted order)', bad[:20].tolist(), 'mod 128:', (bad[:20] % 128).tolist())
