"""What the drop-in boundary costs over PCIe per frame: the C ABI takes DEVICE pointers, so the bench's `value` is HBM-resident
by construction; a caller that keeps its frame state on the host pays the upload of the per-frame arrays (the dataset's batch: posed
vertices, normals, bone matrices, rays) and, if it wants the image on the host, the download of the maps.   python tools/pcie_inclusive.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer

dev = torch.device('cuda:0')
cfg = make_cfg('relight')
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
rend = make_renderer(cfg, net)
host = synthetic.make_batch(512, 512, seed=0, posed=True)
nbytes = sum(v.numel() * v.element_size() for v in host.values() if isinstance(v, torch.Tensor))
N = 12


def run(upload, download):
    resident = synthetic.to_device(host, dev)
    out = None
    for k in range(N + 2):
        if k == 2:
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
        b = synthetic.to_device(host, dev) if upload else resident
        b = type(b)({kk: (vv.clone() if kk == 'wbounds' else vv) for kk, vv in b.items()})      # the renderer grows wbounds in place
        net.engine()._frame_key = None          # every frame is a new frame (no cached body state)
        out = rend.render(b)
        if download:
            img = (out.rgb_map.cpu(), out.acc_map.cpu())
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / N * 1e3


print(f'512 x 512 relight frame, sequential; per-frame batch {nbytes / 1e6:.2f} MB (rays of the {host.ray_o.shape[1]} in-box pixels, posed vertices / normals / bones, mask), image maps 4 floats x {host.ray_o.shape[1]} rays')
for name, u, d in (('inputs and outputs resident in HBM', False, False), ('+ the batch uploaded from pageable host memory every frame', True, False),
                   ('+ rgb and alpha read back to the host every frame', True, True)):
    print(f'{name:66s} {run(u, d):7.2f} ms/frame', flush=True)
