"""Throughput with whole frames in flight on several HIP streams (one context per stream): the latency-bound surface loop of one frame
runs beside the shadow pass of another.   python tools/frames_in_flight.py [--mode relight] [--depth 2] [--emulate-world N]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from relightableavatar_amd import shard, synthetic
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
from relightableavatar_amd.config import make_cfg

ap = argparse.ArgumentParser()
ap.add_argument('--mode', default='relight')
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--steps', type=int, default=24)
ap.add_argument('--depth', type=int, default=2)
ap.add_argument('--emulate-world', type=int, default=1)
args = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = make_cfg(args.mode)
relight = args.mode in ('relight', 'novel_light')
sd = synthetic.make_state_dict(0, relight=relight, cfg=cfg)
nets, rends, batches = [], [], []
for r in range(args.depth):
    net = make_network(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
    nets.append(net); rends.append(make_renderer(cfg, net))
    batches.append(synthetic.to_device(synthetic.make_batch(args.size, args.size, seed=0, posed=True, skin_noise=2.0), dev))
P = batches[0].ray_o.shape[1]
wb0 = batches[0].wbounds.clone(); wbh0 = wb0.cpu()
mask_host = batches[0].mask_at_box.cpu()
streams = [torch.cuda.Stream(dev) for _ in range(args.depth)]
nw = args.emulate_world

def frame(k, use_streams):
    r = k % args.depth
    with torch.cuda.stream(streams[r if use_streams else 0]):
        b = batches[r]
        b.wbounds.copy_(wb0)
        b.wbounds_host, b.wbounds_host_version = wbh0.clone(), b.wbounds._version
        nets[r].engine().set_frame(b, force=True)
        if nw > 1:
            pl = shard.make_plan(P, nw, b, dev, mask=mask_host, ground=False, render_chunk_size=cfg.render_chunk_size, use_cache=False)
            return rends[r].render(shard.shard_batch(b, 0, nw, cfg.render_chunk_size, pl, False))
        return rends[r].render(b)

def timeit(use_streams, label):
    for k in range(6): frame(k, use_streams)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps): frame(k, use_streams)
    torch.cuda.synchronize(dev)
    print(f'{label:40s} {(time.perf_counter() - t0) / args.steps * 1e3:8.3f} ms per frame', flush=True)

timeit(False, 'frames back to back on one stream')
timeit(True, f'{args.depth} frames in flight ({args.depth} streams)')
timeit(False, 'one stream again')
