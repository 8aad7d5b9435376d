"""Race screen of frames in flight (relightableavatar_amd/pipeline.py): N frames of alternating poses through pipelines of depth 2 and 3,
batches dropped right after submit(), every frame compared bit for bit with sequential rendering — relight, relight + ground pass,
novel light + ground pass, volume path.   python tools/soak_pipeline.py [frames]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer
from relightableavatar_amd.pipeline import FramePipeline
dev = torch.device('cuda:0')
GROUND = dict(vis_ground_shading=True, ground_normal=[0.0, -1.0, 0.0], ground_origin=[0.0, 0.45, 0.0])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for mode, kw in (('relight', {}), ('relight', GROUND), ('novel_light', dict(GROUND, novel_light_timing=False)), ('anisdf', {})):
    cfg = make_cfg(mode, **kw)
    relight = mode in ('relight', 'novel_light')
    nl = 2 if mode == 'novel_light' else 0
    sd = synthetic.make_state_dict(0, relight=relight, cfg=cfg)
    net = make_network(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
    serial = make_renderer(cfg, net)
    H = 256 if relight else 128
    mk = lambda seed: synthetic.to_device(synthetic.make_batch(H, H, seed=seed, posed=True, n_novel_lights=nl), dev)
    pick = (lambda out: out[sorted(k for k in out.keys() if k not in ('diff', 'main'))[0]]) if mode == 'novel_light' else (lambda out: out)
    want = []
    for seed in (0, 1, 2):
        out = pick(serial.render(mk(seed)))
        want.append({k: out[k].clone() for k in ('rgb_map', 'acc_map')})
    bad = 0
    for depth in (2, 3):
        pipe = FramePipeline(cfg, sd, dev, depth=depth)
        pend = [pipe.submit(mk(k % 3)) for k in range(N)]
        for k, p in enumerate(pend):
            out = pick(p.result())
            for key in ('rgb_map', 'acc_map'):
                if not torch.equal(out[key], want[k % 3][key]):
                    bad += 1
        torch.cuda.synchronize()
    print(mode, sorted(kw.keys()), f'{2 * N} frames, differing from sequential rendering:', bad, flush=True)
    assert bad == 0
