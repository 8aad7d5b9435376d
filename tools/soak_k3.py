"""Race screen for the streamed K3 (LDS-DMA ring): the same 2 M-point query N times, every result must be bit-identical
to the first, for the 8-, 4- and 2-wave variants (RA_STREAM_NW is read once per process: run once per variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
cfg = make_cfg('relight')
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
eng = net.set_frame(body)
g = torch.Generator().manual_seed(11)
x = ((torch.rand(2000003, 3, generator=g) - 0.5) * 0.9).to(dev)
ref = eng.hdq_sdf(x, 0.125, True).clone()
bad = 0
for i in range(n_rep):
    s = eng.hdq_sdf(x, 0.125, True)
    if not torch.equal(s, ref):
        bad += 1
        print('run', i, 'differs in', int((s != ref).sum()), 'points')
print(f'soak NW={os.environ.get("RA_STREAM_NW", "auto")}: {n_rep} runs, {bad} differing, {int(torch.isnan(ref).sum())} NaN')
sys.exit(1 if bad else 0)
