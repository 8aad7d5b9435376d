"""Adaptive precision of the shadow rays (cfg.shadow_requery_tol, csrc/ra_trace.hip requery_select / requery_apply): BASELINE's 512 x 512
relight frame with trace_precision 1 at several tolerances against the same frame with EVERY distance query compensated
(trace_precision 2) — pixels over 1e-2 / 5e-3, the share of fine queries that is re-answered in compensated arithmetic, and the
sequential frame time.  Usage (GPU box): python3 tools/ab_requery.py [H] [skin_noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
from relightableavatar_amd.renderer import make_renderer

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sn = float(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device('cuda:0')


def run(tp, tol, frames=6):
    cfg = make_cfg('relight', trace_precision=tp, shadow_requery_tol=tol)
    net = make_network(cfg)
    net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg))
    net = net.to(dev).eval()
    kw = {} if sn is None else dict(skin_noise=sn)
    batch = synthetic.to_device(synthetic.make_batch(H, H, seed=0, posed=True, **kw), dev)
    r = make_renderer(cfg, net)
    out = r.render(batch)
    torch.cuda.synchronize()
    net.engine().reset_counters()
    t0 = time.perf_counter()
    for _ in range(frames):
        out = r.render(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
    c = net.engine().counters()
    return {k: out[k].clone() for k in ('rgb_map', 'acc_map', 'surf_map')}, ms, c.n_fine_sdf / frames, c.n_fine_sdf_comp / frames


ref, ms2, nf2, nc2 = run(2, 0.0, frames=2)
hit = ref['acc_map'][0] > 0
print(f'{H} x {H}, skin_noise {sn}: {int(hit.sum())} hit pixels; trace_precision 2: {ms2:.1f} ms/frame sequential, {nf2 / 1e6:.2f} M fine queries')
base_comp = None
for tol in (0.0, 1e-2, 5e-3, 2e-3, 1e-3, 5e-4):
    a, ms, nf, nc = run(1, tol)
    assert torch.equal(a['acc_map'], ref['acc_map']) and torch.equal(a['surf_map'], ref['surf_map'])
    e = (a['rgb_map'] - ref['rgb_map']).abs()[0]
    pe = e.amax(-1)
    ph = float(-10 * torch.log10((e[hit] ** 2).mean()))
    if base_comp is None:
        base_comp = nc
    print(f'tol {tol:7.1e}: {ph:5.1f} dB over hit pixels, max {float(e.max()):.2e}, pixels > 1e-2: {int((pe > 1e-2).sum())}, > 5e-3: {int((pe > 5e-3).sum())}, > 2e-3: {int((pe > 2e-3).sum())}; '
          f're-queried {(nc - base_comp) / 1e3:8.1f} k of {(nf - nc) / 1e6:.2f} M plain fine queries ({100 * (nc - base_comp) / max(nf - nc, 1):.2f} %); {ms:.2f} ms/frame sequential', flush=True)
