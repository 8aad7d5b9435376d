"""Where does a wave of the coarse level (hdq_coarse_kernel) spend its cycles?  Instrumented build:
    tools/build_variant.sh hdqts ra_hdq.hip "-DRA_COARSE_TS"      then on the GPU box     RA_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_tmp/variants/hdqts.so python tools/coarse_timestamps.py [mode]
Renders frames of `mode` (default sphere_tracing, 512 x 512; an optional second argument N renders rank 0's shard of an N-rank job) and prints, for each of the last 64 coarse launches, the median cycles
per phase over the waves of the first 64 workgroups."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from relightableavatar_amd import synthetic, _lib
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
mode = sys.argv[1] if len(sys.argv) > 1 else 'sphere_tracing'
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device('cuda:0')
cfg = make_cfg(mode)
relight = mode in ('relight', 'novel_light')
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=relight, cfg=cfg)); net = net.to(dev).eval()
renderer = make_renderer(cfg, net)
from relightableavatar_amd import shard
for _ in range(2):
    batch = synthetic.to_device(synthetic.make_batch(512, 512, seed=0, posed=True, skin_noise=2.0), dev)
    if world > 1:
        batch = shard.shard_batch(batch, 0, world, cfg.render_chunk_size)
    renderer.render(batch)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(64 * 64 * 16 * 8, dtype=np.int64)
L.ra_coarse_read_timestamps.restype = C.c_int
assert L.ra_coarse_read_timestamps(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(64, 64, 16, 8)
names = ['point', 'seed', 'seed scan', 'sweep', 'merge', 'signs+compact']
print('launch slot | waves | median cycles per phase: ' + ' | '.join(names) + ' | total || leaves scanned, supers opened per wave (mean)')
for li in range(64):
    w = t[li].reshape(-1, 8)
    w = w[(w[:, 0] > 0) & (w[:, 4] > 0)]
    if len(w) == 0:
        continue
    nw = int((t[li, 0, :, 0] > 0).sum())
    d = np.diff(w[:, :7], axis=1).astype(np.float64)
    d[w[:, 1:7] == 0] = np.nan
    med = np.nanmedian(d, axis=0)
    tot = np.nanmedian((np.where(w[:, 6] > 0, w[:, 6], w[:, 5]) - w[:, 0]).astype(np.float64))
    # slowest wave of a workgroup: what the workgroup's queries wait for
    wg = t[li][:, :nw, :]
    span = (wg[:, :, 4] - wg[:, :, 0]).max(axis=1)
    span = span[wg[:, 0, 0] > 0]
    print(f'{li:3d} | {nw:2d} waves/wg | ' + ' | '.join(f'{x:7.0f}' for x in med) + f' | {tot:7.0f} || {np.mean(w[:, 7] >> 32):5.1f} {np.mean(w[:, 7] & 0xfff):5.1f} insert groups {np.mean((w[:, 7] >> 12) & 0xfffff):6.1f} | slowest wave of a workgroup start->sweep end {np.median(span):7.0f}')
