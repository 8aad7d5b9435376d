"""gen-2 (pipelined) vs gen-1 K3 kernel: results must be bit-identical. Run twice with RA_MLP_GEN=1/2 and compare dumps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
cfg = make_cfg('relight', mlp_dtype=os.environ.get('RA_DTYPE', 'f16'))
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
g = torch.Generator().manual_seed(5)
x = ((torch.rand(300001, 3, generator=g) - 0.5) * 0.9).to(dev)
s = eng.hdq_sdf(x, 0.125, True)
torch.save(s.cpu(), sys.argv[1])
print('saved', sys.argv[1], float(s.abs().mean()))
