import os, sys
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
eng = net.engine()
print('set_frame...'); sys.stdout.flush()
eng.set_frame(body)
torch.cuda.synchronize()
print('set_frame ok'); sys.stdout.flush()
x = (torch.rand(1000, 3, device=dev) - 0.5)
o = eng.debug_hdq(x, 0.125)
torch.cuda.synchronize()
print('hdq ok', o.fine_count)
