#!/bin/bash
# A/B of a K3 variant library against the in-tree one: bit-identity on 300 k points, kernel time on 5.1 M points, the relight frame
V=${1:-w64}
python - <<PY
import os, sys, subprocess, torch
sys.path.insert(0, '.')
code = '''
import sys, torch
sys.path.insert(0, ".")
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device("cuda:0")
cfg = make_cfg("relight", trace_precision=0)
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
eng = net.set_frame(synthetic.to_device(synthetic.make_body(0, posed=True), dev))
g = torch.Generator().manual_seed(5)
d = torch.nn.functional.normalize(torch.randn(300001, 3, generator=g), dim=-1)
x = (d * (0.38 + 0.12 * torch.rand(300001, 1, generator=g))).to(dev)
out = eng.observed_sdf(x)
torch.save(out.cpu(), sys.argv[1])
print("checksum", float(out.double().sum()))
'''
open('/tmp/ab_w64_probe.py', 'w').write(code)
for lib, f in ((None, '/tmp/ab_ref.pt'), ('gpurun_tmp/variants/$V.so', '/tmp/ab_var.pt')):
    env = dict(os.environ)
    if lib: env['RA_LIB_PATH'] = lib
    r = subprocess.run([sys.executable, '/tmp/ab_w64_probe.py', f], env=env, capture_output=True, text=True)
    print(lib or 'in-tree', r.stdout.strip()[-80:], r.stderr.strip()[-300:] if r.returncode else '')
a, b = torch.load('/tmp/ab_ref.pt'), torch.load('/tmp/ab_var.pt')
print('bit-identical:', bool(torch.equal(a, b)), 'max diff', float((a - b).abs().max()))
PY
for lib in "" gpurun_tmp/variants/$V.so; do
  echo "== lib: ${lib:-in-tree}"
  RA_LIB_PATH=$lib RA_NV=80000 python tools/bench_mlp.py 2>&1 | grep -v amdgpu.ids
  RA_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-sequential --trace-precision 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frame ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4), 'avg launch ms', round(d['roofline']['avg_launch_ms'],3))"
done
