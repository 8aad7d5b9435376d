"""Does the chip gain from running two independent halves of a frame on two HIP streams (VALU-bound coarse search of one half beside
the MFMA-bound K3 of the other)?  Two contexts (own arenas), shard r of 2 of the same 512 x 512 relight frame on stream r.
    python tools/overlap_experiment.py [--mode relight] [--steps 20]
Prints ms per full frame: one context serial (whole frame), two shards back to back on ONE stream, two shards on TWO streams."""
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
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--parts', type=int, default=2)
args = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = make_cfg(args.mode)
relight = args.mode in ('relight', 'novel_light')
sd = synthetic.make_state_dict(0, relight=relight, cfg=cfg)
nets, rends, batches = [], [], []
for r in range(args.parts):
    net = make_network(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
    nets.append(net); rends.append(make_renderer(cfg, net))
    batches.append(synthetic.to_device(synthetic.make_batch(args.size, args.size, seed=0, posed=True, skin_noise=2.0), dev))
P = batches[0].ray_o.shape[1]
wb0 = batches[0].wbounds.clone(); wbh0 = wb0.cpu()
mask_host = batches[0].mask_at_box.cpu()
streams = [torch.cuda.Stream(dev) for _ in range(args.parts)]

def fresh(b):
    b.wbounds.copy_(wb0)
    b.wbounds_host, b.wbounds_host_version = wbh0.clone(), b.wbounds._version

def whole():
    fresh(batches[0]); nets[0].engine().set_frame(batches[0], force=True)
    return rends[0].render(batches[0])

def parts(use_streams):
    outs = []
    pl = shard.make_plan(P, args.parts, batches[0], dev, mask=mask_host, ground=False, render_chunk_size=cfg.render_chunk_size, use_cache=False)
    for r in range(args.parts):
        ctx = torch.cuda.stream(streams[r]) if use_streams else torch.cuda.stream(streams[0])
        with ctx:
            fresh(batches[r]); nets[r].engine().set_frame(batches[r], force=True)
            outs.append(rends[r].render(shard.shard_batch(batches[r], r, args.parts, cfg.render_chunk_size, pl, False)))
    return outs

def timeit(fn, label):
    for _ in range(5): fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps): fn()
    torch.cuda.synchronize(dev)
    print(f'{label:44s} {(time.perf_counter() - t0) / args.steps * 1e3:8.3f} ms per frame', flush=True)

timeit(whole, 'whole frame, one context, one stream')
timeit(lambda: parts(False), f'{args.parts} shards, one stream')
timeit(lambda: parts(True), f'{args.parts} shards, {args.parts} streams')
timeit(whole, 'whole frame again')
