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
# round 5: + sphere tracing, + a multi-chunk frame with 8 probes (config-5-like: merged render chunks, several boxes per launch sequence)
for mode, kw in (('relight', {}), ('relight', GROUND), ('novel_light', dict(GROUND, novel_light_timing=False)), ('anisdf', {}),
                 ('sphere_tracing', {}), ('novel_light', dict(novel_light_timing=False, render_chunk_size=4096, n_probes=8)),
                 # + the switches the round-5 matrix added to the product: hard shadows (the shadow rays carry the surface trace's state), the
                 # K-NN rule without the geodesic filter, the one-channel visibility debug output
                 ('relight', dict(GROUND, no_dfss=True)), ('relight', dict(use_geodesic_filter=False, vis_lvis_map=True)),
                 ('relight', dict(only_visibility=True, n_samples=1))):
    kw = dict(kw)
    relight = mode in ('relight', 'novel_light')
    nl = kw.pop('n_probes', 2) if mode == 'novel_light' else 0
    cfg = make_cfg(mode, **kw)
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
        pend = []                       # a sliding window: a long soak must not keep every frame's maps alive (500 ground frames are 270 GB)
        for k in range(N + 2 * depth):
            if k < N:
                pend.append((k, pipe.submit(mk(k % 3))))
            if len(pend) > 2 * depth or k >= N:
                if not pend:
                    break
                j, p = pend.pop(0)
                out = pick(p.result())
                for key in ('rgb_map', 'acc_map'):
                    if not torch.equal(out[key], want[j % 3][key]):
                        bad += 1
                del out, p
        torch.cuda.synchronize()
    print(mode, sorted(kw.keys()), f'{2 * N} frames, differing from sequential rendering:', bad, flush=True)
    assert bad == 0

# ---- SHARDED frames in flight (round 5): every frame is one rank's shard of a 4-rank job — plan rebuilt per frame (shard.make_plan, the C
# call), shard_batch on the replica's stream — against the same shard rendered sequentially
from relightableavatar_amd import shard
cfg = make_cfg('relight')
sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
net = make_network(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
serial = make_renderer(cfg, net)
H = 256
def shard_of(seed, rank):
    b = mk2(seed)
    P = b.ray_o.shape[1]
    pl = shard.make_plan(P, 4, b, dev, mask=b.mask_at_box.cpu(), render_chunk_size=cfg.render_chunk_size, use_cache=False)
    return shard.shard_batch(b, rank, 4, cfg.render_chunk_size, pl)
mk2 = lambda seed: synthetic.to_device(synthetic.make_batch(H, H, seed=seed, posed=True), dev)
want = []
for k in range(12):                      # frame k renders shard (k % 3, k % 4): period 12
    out = serial.render(shard_of(k % 3, k % 4))
    want.append({kk: out[kk].clone() for kk in ('rgb_map', 'acc_map')})
bad = 0
for depth in (2, 3):
    pipe = FramePipeline(cfg, sd, dev, depth=depth)
    fno = [0]
    def sframe(net_r, rend_r):
        k = fno[0]; fno[0] += 1
        return rend_r.render(shard_of(k % 3, k % 4))
    pend = []
    for k in range(N + 2 * depth):
        if k < N:
            pend.append((k, pipe.submit(fn=sframe)))
        if len(pend) > 2 * depth or k >= N:
            if not pend:
                break
            j, p = pend.pop(0)
            out = p.result()
            w = want[j % 12]
            for key in ('rgb_map', 'acc_map'):
                if out[key].shape != w[key].shape or not torch.equal(out[key], w[key]):
                    bad += 1
            del out, p
    torch.cuda.synchronize()
print('sharded relight (rank k % 4 of 4, plan rebuilt per frame)', f'{2 * N} frames, differing from sequential rendering:', bad, flush=True)
assert bad == 0

# ---- an ANIMATED sequence through the device-side loader (N3 + N2, relightableavatar_amd/data_utils.py): every frame a different pose, its
# rays generated a pipeline turn ahead, against the same frames posed, culled and rendered strictly one after the other
import numpy as np
from relightableavatar_amd.data_utils import DeviceFrameLoader
cfg = make_cfg('relight')
sd = synthetic.make_state_dict(0, relight=True, cfg=cfg)
sk = synthetic.make_skeleton(0)
H = 256
K, R, Tc = synthetic.make_camera(H, H)
tv, w = torch.from_numpy(sk.tverts).to(dev), torch.from_numpy(sk.weights).to(dev)
net = make_network(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
eng0 = net.engine()
eye = np.tile(np.eye(4, dtype=np.float32), (52, 1, 1))
big_A = eng0.pose_frame(sk.big_poses, sk.tjoints, sk.parents, tv, w, eye, sk.faces, np.zeros(3, np.float32), np.zeros(3, np.float32)).A.cpu().numpy()
loader = DeviceFrameLoader(H, H, K, R, Tc, sk.tjoints, sk.parents, tv, w, big_A, sk.faces)
ph = np.arange(sk.poses.size, dtype=np.float32).reshape(sk.poses.shape)
NA = 12
seq = [((sk.poses + 0.06 * np.sin(0.37 * f + ph)).astype(np.float32), sk.Rh, (sk.Th + np.float32(0.01 * np.sin(0.3 * f))).astype(np.float32)) for f in range(NA)]
serial = make_renderer(cfg, net)
want = []
for q in seq:
    out = serial.render(loader.batch(loader.issue(eng0, *q)))
    torch.cuda.synchronize()
    want.append({k: out[k].clone() for k in ('rgb_map', 'acc_map')})
bad = 0
for depth in (2, 3):
    pipe = FramePipeline(cfg, sd, dev, depth=depth)
    ahead = [None] * depth
    fno = [0]

    def frame(net_r, rend_r):
        f = fno[0]
        r = f % depth
        eng = net_r.engine()
        pend = ahead[r] or loader.issue(eng, *seq[f % NA])
        out = rend_r.render(loader.batch(pend))
        ahead[r] = loader.issue(eng, *seq[(f + depth) % NA])
        fno[0] += 1
        return out
    pend = []
    for k in range(N + 2 * depth):
        if k < N:
            pend.append((k, pipe.submit(fn=frame)))
        if len(pend) > 2 * depth or k >= N:
            if not pend:
                break
            j, p = pend.pop(0)
            out = p.result()
            for key in ('rgb_map', 'acc_map'):
                if out[key].shape != want[j % NA][key].shape or not torch.equal(out[key], want[j % NA][key]):
                    bad += 1
            del out, p
    torch.cuda.synchronize()
print('animated relight (DeviceFrameLoader, poses / rays issued a turn ahead)', f'{2 * N} frames, differing from sequential rendering:', bad, flush=True)
assert bad == 0
