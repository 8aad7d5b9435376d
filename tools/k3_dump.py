"""distances of the production distance query on a fixed point set -> a .pt file (bit-identity checks between kernel variants:
RA_LIB_PATH=gpurun_tmp/variants/X.so python3 tools/k3_dump.py out.pt; python3 tools/k3_dump.py cmp a.pt b.pt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == 'cmp':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    d = (a - b).abs()
    print(f'{sys.argv[2]} vs {sys.argv[3]}: {int((a != b).sum())} of {a.numel()} differ, max |diff| {float(d.max()):.3e}')
    sys.exit(0)
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
cfg = make_cfg('relight', trace_precision=0)
net = make_network(cfg)
net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg))
net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
g = torch.Generator().manual_seed(1)
outs = []
for nv in (4700, 17, 1000):          # 300 800 points (8-wave kernel, partly filled last round), 1 088 (2-wave), 64 000 (4-wave)
    vid = torch.randint(0, 6890, (nv,), generator=g)
    wv = (body.pverts[0] @ body.R[0].T + body.Th[0])[vid.to(dev)]
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1).to(dev)
    x = (wv[:, None, :] + 0.02 * dirs[None]).reshape(-1, 3).contiguous()
    outs.append(eng.hdq_sdf(x, 0.125, True).reshape(-1).cpu())
torch.save(torch.cat(outs), sys.argv[1])
print('saved', sys.argv[1], sum(o.numel() for o in outs))
