"""N2: time of the on-device ray set-up (ra_gen_rays, includes the count read-back) vs the numpy path + upload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
cfg = make_cfg('relight')
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
for H in (512, 1024):
    K, R, T = synthetic.make_camera(H, H)
    eng.gen_rays(H, H, K, R, T, body.wbounds[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): o = eng.gen_rays(H, H, K, R, T, body.wbounds[0])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    ro, rd, near, far, mask = synthetic.rays_within_bounds(H, H, K, R, T, body.wbounds[0].cpu().numpy().astype(np.float64))
    up = [torch.from_numpy(a).to(dev) for a in (ro, rd, near, far)]; torch.cuda.synchronize()
    dc = time.perf_counter() - t0
    print(f'{H}x{H}: device {dt*1e3:.3f} ms ({o.ray_o.shape[0]} rays; {H*H*1 + o.ray_o.shape[0]*32} B written), numpy + upload {dc*1e3:.1f} ms')
