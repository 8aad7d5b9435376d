"""micro-benchmark of K3C (compensated distance query) beside the plain K3 on launch sizes of the surface trace; HIP-event time per launch.
RA_LIB_PATH selects a variant library (tools/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
d = torch.nn.functional.normalize(torch.randn(70000, 3, generator=g), dim=-1)
bpts = (d * (0.38 + 0.12 * torch.rand(70000, 1, generator=g))).to(dev)
body = synthetic.to_device(synthetic.make_body(0, posed=True), dev)
ref = None
for tp in (2, 0):
    cfg = make_cfg('relight', trace_precision=tp)
    net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
    eng = net.set_frame(body)
    for n in [int(v) for v in os.environ.get("RA_K3C_SIZES", "2400,8000,16384,20800,32768,65536").split(",")]:
        x = bpts[:n].contiguous()
        for _ in range(3):
            out = eng.observed_sdf(x)
        eng.reset_counters(); eng.enable_timing(True)
        for _ in range(10):
            eng.observed_sdf(x)
        ms, k = eng.kernel_time(4 if tp else 0)
        eng.enable_timing(False)
        print(f'trace_precision {tp} n {n}: {ms / max(k, 1) * 1e3:.1f} us per launch ({k} launches)', flush=True)
    if tp == 2:
        print('checksum', float(eng.observed_sdf(bpts[:20000]).double().sum()))
