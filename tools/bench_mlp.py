"""micro-benchmark of the fused HDQ MLP kernel (K3) on 64 * RA_NV (default 5.1 M) fine points; prints HIP-event time and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
cfg = make_cfg('relight', mlp_dtype=os.environ.get('RA_DTYPE', 'f16'))
net = make_network(cfg)
net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg))
net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
g = torch.Generator().manual_seed(0)
vid = torch.randint(0, 6890, (int(os.environ.get('RA_NV', '80000')),), generator=g)
wv = (body.pverts[0] @ body.R[0].T + body.Th[0])[vid.to(dev)]
dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1).to(dev)
x = (wv[:, None, :] + 0.02 * dirs[None]).reshape(-1, 3).contiguous()
for _ in range(2):
    eng.hdq_sdf(x, 0.125, True)
eng.reset_counters(); eng.enable_timing(True)
for _ in range(5):
    eng.hdq_sdf(x, 0.125, True)
ms, n = eng.mlp_time(); c = eng.counters()
print(f'ablate={os.environ.get("RA_MLP_ABLATE", "0")}: {ms / n:.3f} ms per launch, {c.n_fine_sdf / n:.0f} pts, {c.n_fine_sdf * 1901568 / (ms * 1e-3) / 1e12:.0f} TFLOP/s (algorithmic)')
