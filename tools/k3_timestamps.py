"""Per-layer cycle table of the production K3 from its instrumented variant (build: tools/build_variant.sh k3ts ra_k3_f16.hip "-DRA_TIMESTAMPS";
run on the GPU box with RA_LIB_PATH=gpurun_tmp/variants/k3ts.so).  Wave 0 of every workgroup stamps s_memtime at the layer boundaries of
its first tile; the launch's duration comes from HIP events, so cycles / time = the clock the kernel really ran at.
  RA_NV (default 80000 -> 5.1 M points: the 8-wave variant, every CU busy for ~78 tiles); RA_NV=8 -> one tile per CU, 2-wave variant."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from relightableavatar_amd import synthetic, _lib
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
nv = int(os.environ.get('RA_NV', '80000'))
vid = torch.randint(0, 6890, (nv,), generator=g)
wv = (body.pverts[0] @ body.R[0].T + body.Th[0])[vid.to(dev)]
dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1).to(dev)
x = (wv[:, None, :] + 0.02 * dirs[None]).reshape(-1, 3).contiguous()
for _ in range(3):
    eng.hdq_sdf(x, 0.125, True)
eng.reset_counters(); eng.enable_timing(True)
eng.hdq_sdf(x, 0.125, True)
ms, n = eng.mlp_time(); cnt = eng.counters()
L = _lib.lib()
buf = np.zeros(256 * 48, dtype=np.int64)
L.ra_k3_read_timestamps.restype = C.c_int
assert L.ra_k3_read_timestamps(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(256, 48)
t = t[t[:, 21] > 0]
names = ['resd L0'] + [f'resd L{i}' for i in range(1, 8)] + ['resd head'] + ['sdf L0'] + [f'sdf L{i}' for i in range(1, 8)] + ['sdf head']
mf = [32] + [128] * 3 + [160] + [128] * 3 + [16] + [32] + [128] * 3 + [160] + [128] * 3 + [16]
d = np.diff(t[:, :20], axis=1).astype(np.float64)          # 19 intervals: 18 layers + tile tail
tile_cyc = (t[:, 19] - t[:, 0]).astype(np.float64)
wall_cyc = (t[:, 20] - t[:, 0]).astype(np.float64)          # first tile start -> last tile end of the workgroup
pts = cnt.n_fine_sdf / n
wps = 2 if pts > 65536 else 1                                # waves per SIMD of the variant the launcher picked
print(f'launch: {pts:.0f} points, {ms / n * 1e3:.1f} us, {pts * 1901568 / (ms / n * 1e-3) / 1e12:.0f} TFLOP/s algorithmic; {len(t)} workgroups, {t[:, 21].mean():.1f} tiles each')
print(f'clock: first-tile start -> last-tile end {np.median(wall_cyc):.0f} cycles per workgroup over {ms / n * 1e3:.1f} us launch  =>  >= {np.median(wall_cyc) / (ms / n * 1e3) / 1e3:.2f} GHz')
print(f'{"layer":10s} {"MFMA/wave":>9s} {"cycles (median)":>16s} {"cycles/MFMA/wave":>17s} {"cycles/MFMA/SIMD":>17s}')
for k, (nm, m) in enumerate(zip(names, mf)):
    c = np.median(d[:, k])
    print(f'{nm:10s} {m:9d} {c:16.0f} {c / m:17.1f} {c / m / wps:17.1f}')
print(f'{"tile":10s} {sum(mf):9d} {np.median(tile_cyc):16.0f} {np.median(tile_cyc) / sum(mf):17.1f} {np.median(tile_cyc) / sum(mf) / wps:17.1f}   (first tile of each workgroup; tail after the sdf head {np.median(d[:, 18]):.0f} cycles)')
r = d[:, 1:8].sum(1) / (3 * 128 + 160 + 3 * 128); sp = d[:, 10:17].sum(1) / (3 * 128 + 160 + 3 * 128)
print(f'ReLU net L1-L7: {np.median(r) / wps:.1f} cycles/MFMA/SIMD; softplus net L1-L7: {np.median(sp) / wps:.1f}  (MFMA pipe: 32)')
