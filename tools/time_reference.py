#!/usr/bin/env python3
"""Reference-on-CPU anchor: the GENUINE reference hot path, imported from /root/reference in
the build container with the third-party stubs of tests/golden/make_golden.py, timed on the same synthetic 64x64 inputs the
build's CPU twin is timed on.  The KNN stub (brute-force squared-L2 top-3 on the CPU, in place of pytorch3d's CUDA kernel) is
timed separately, as the survey asked.  One mode per process (the reference binds cfg at import).

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py {anisdf|sphere|relight} [--size 64]
"""
import argparse
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch  # noqa: E402
import make_golden as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('mode', choices=['anisdf', 'sphere', 'relight'])
    ap.add_argument('--size', type=int, default=64)
    args = ap.parse_args()
    from relightableavatar_amd import synthetic
    from relightableavatar_amd.config import make_cfg
    cfg = G.install_reference()
    G.set_cfg(cfg, args.mode)
    relight = args.mode == 'relight'
    sd = synthetic.make_state_dict(0, relight=relight, cfg=make_cfg({'anisdf': 'anisdf', 'sphere': 'sphere_tracing', 'relight': 'relight'}[args.mode]))
    import pytorch3d.ops
    knn = pytorch3d.ops.knn_points
    t_knn = [0.0, 0]

    def timed_knn(*a, **k):
        t0 = time.perf_counter()
        r = knn(*a, **k)
        t_knn[0] += time.perf_counter() - t0
        t_knn[1] += a[0].shape[1]
        return r
    pytorch3d.ops.knn_points = timed_knn
    import lib.utils.sample_utils as su
    su.knn_points = timed_knn
    if relight:
        from lib.networks.relight.relight_network import Network
    else:
        from lib.networks.deform.base_network import Network
    net = Network()
    net.load_state_dict(sd, strict=False)
    net.eval()
    from lib.networks.renderer import base_renderer, sphere_tracing_renderer
    H = args.size
    batch = G.to_ref_batch(synthetic.make_batch(H, H, seed=0, posed=True))
    rend = base_renderer.Renderer(net) if args.mode == 'anisdf' else sphere_tracing_renderer.Renderer(net)
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = rend.render(batch)
    dt = time.perf_counter() - t0
    P = batch.ray_o.shape[1]
    hits = int((out.acc_map > 0).sum())
    print(f'{args.mode} {H}x{H}: {P} in-box rays of {H * H}, {hits} hit; reference render {dt:.2f} s on {torch.get_num_threads()} threads '
          f'= {H * H / dt:.0f} rays/s ({P / dt:.0f} in-box rays/s); KNN stub {t_knn[0]:.2f} s over {t_knn[1]} queries; '
          f'without KNN {dt - t_knn[0]:.2f} s = {H * H / max(dt - t_knn[0], 1e-9):.0f} rays/s')


if __name__ == '__main__':
    main()
