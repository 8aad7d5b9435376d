"""N3: time of the per-frame body state on the device (ra_pose_frame) vs the oracle's torch CPU restatement."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ra_oracle as O
from relightableavatar_amd import synthetic
from relightableavatar_amd.config import make_cfg
from relightableavatar_amd.networks import make_network
dev = torch.device('cuda:0')
cfg = make_cfg('relight')
net = make_network(cfg); net.load_state_dict(synthetic.make_state_dict(0, relight=True, cfg=cfg)); net = net.to(dev).eval()
eng = net.set_frame(synthetic.to_device(synthetic.make_body(0, posed=True), dev))
sk = synthetic.make_skeleton(0)
T = torch.from_numpy
big_A, _ = O.rigid_transforms(T(sk.big_poses), T(sk.tjoints), T(sk.parents))
tv, w = T(sk.tverts).to(dev), T(sk.weights).to(dev)
eng.pose_frame(sk.poses, sk.tjoints, sk.parents, tv, w, big_A, sk.faces, sk.Rh, sk.Th)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): eng.pose_frame(sk.poses, sk.tjoints, sk.parents, tv, w, big_A, sk.faces, sk.Rh, sk.Th)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
t0 = time.perf_counter()
O.pose_frame(T(sk.poses), T(sk.tjoints), T(sk.parents), T(sk.tverts), T(sk.weights), big_A, T(sk.faces), T(sk.Rh), T(sk.Th))
dc = time.perf_counter() - t0
print(f'pose_frame: device {dt*1e3:.3f} ms per frame (6890 vertices, 52 bones, 13776 faces), torch CPU restatement {dc*1e3:.1f} ms')
