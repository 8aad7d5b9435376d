import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
cfg = make_cfg('relight')
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
torch.manual_seed(0)
bp = (torch.rand(128 * 600, 3, device=dev) - 0.5) * 0.8
r1, s1, f1 = eng.debug_mlp(bp, want_feat=True)
for sh in (1, 2, 32, 64, 128, 256, 128 * 512):
    bq = torch.roll(bp, sh, 0).contiguous()
    r2, s2, f2 = eng.debug_mlp(bq, want_feat=True)
    d = (torch.roll(s1, sh, 0) - s2).abs()
    df = (torch.roll(f1, sh, 0) - f2).abs()
    bad = (d > 0).nonzero().flatten()
    print(f'shift {sh}: sdf n_bad {bad.numel()} max {float(d.max()):.2e}; feat n_bad_rows {int((df.max(1)[0] > 0).sum())}; bad slots mod 128: {sorted(set((bad % 128).tolist()))[:12]}')
# which half of the nets? compare resd too
