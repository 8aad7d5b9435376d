"""micro-benchmark of the coarse level on coherent point sets (run under rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
cfg = make_cfg('relight')
net = make_network(cfg)
net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg))
net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
g = torch.Generator().manual_seed(0)
G = 80000
vid = torch.randint(0, 6890, (G,), generator=g)
wv = (body.pverts[0] @ body.R[0].T + body.Th[0])[vid.to(dev)]           # world-space surface points
dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1).to(dev)
for r in (0.02, 0.05, 0.1, 0.2, 0.4):
    x = (wv[:, None, :] + r * dirs[None]).reshape(-1, 3).contiguous()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        o = eng.debug_hdq(x, 0.125)
        torch.cuda.synchronize(); dt = time.time() - t0
    c = eng.counters(); eng.reset_counters()
    print(f'r={r}: {x.shape[0]} pts, fine {o.fine_count}, wall {dt*1e3:.2f} ms; per wave: leaves scanned {c.n_shadow_rays / (2 * x.shape[0] / 64):.1f}, supers opened {c.n_hit_pixels / (2 * x.shape[0] / 64):.1f}')
xr = (torch.rand(5120000, 3, device=dev) - 0.5) * 1.2
o = eng.debug_hdq(xr, 0.125); torch.cuda.synchronize()
t0 = time.time(); o = eng.debug_hdq(xr, 0.125); torch.cuda.synchronize()
print(f'random: wall {(time.time()-t0)*1e3:.2f} ms fine {o.fine_count}')
