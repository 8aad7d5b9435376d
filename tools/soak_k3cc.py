"""Race screen for K3CC (csrc/ra_k3cc.hpp: weights straight into named AGPRs, activations exchanged through LDS between four waves):
launches of random sizes up to 8 Ki points, N times, beside a second context that keeps the chip busy with large plain-K3 launches on
another stream; every result must be bit-identical to the per-wave kernel's (K3C) on the same points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
nets = []
for tp in (2, 0):
    cfg = make_cfg('relight', trace_precision=tp)
    net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); nets.append(net.to(dev).eval())
eng = nets[0].set_frame(body)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eng_bg = nets[1].set_frame(body)
g = torch.Generator().manual_seed(5)
d = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
bpts = (d * (0.38 + 0.12 * torch.rand(20000, 1, generator=g))).to(dev)
ref = eng.observed_sdf(bpts).clone()             # 20 000 points: the 8-wave per-wave kernel
xbg = ((torch.rand(1500000, 3, generator=g) - 0.5) * 0.9).to(dev)
torch.cuda.synchronize()
bad = 0
for i in range(n_rep):
    if i % 4 == 0:
        with torch.cuda.stream(side):
            eng_bg.hdq_sdf(xbg, 0.125, True)
    n = int(torch.randint(1, 8193, (1,), generator=g))
    o = int(torch.randint(0, 20000 - n + 1, (1,), generator=g))
    s = eng.observed_sdf(bpts[o:o + n].contiguous())
    if not torch.equal(s, ref[o:o + n]):
        bad += 1
        print('run', i, 'n', n, 'differs in', int((s != ref[o:o + n]).sum()), 'points')
torch.cuda.synchronize()
print(f'soak K3CC: {n_rep} launches of 1..8192 points beside large K3 launches, {bad} differing, {int(torch.isnan(ref).sum())} NaN')
sys.exit(1 if bad else 0)
