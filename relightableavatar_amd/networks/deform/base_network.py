"""AniSDF network, host-side mirror of lib/networks/deform/base_network.py.

Same class name, constructor contract (no arguments, everything from cfg), state_dict keys
 and method surface the renderers use:
  forward(x, v, d, batch) -> dotdict(raw)                              base_network.py:496-515
  inference_world_distance_field(x, batch, smooth_transition, **kw)    base_network.py:385-387
  inference_observed_distance_field(x, batch, smooth_transition, filtering, **kw)     base_network.py:447-449
  world_to_bigpose_transform / bigpose_to_world_transform(x, batch)    base_network.py:338-363
  signed_distance_network.beta                                         base_network.py:74-76
The modules below hold parameters only; all arithmetic runs in the HIP library through Engine.
"""
import torch
from torch import nn

from ... import config
from ...base_utils import dotdict
from ...engine import Engine


def _buffer(t):   # make_buffer: nn.Parameter(requires_grad=False) so it lands in state_dict (net_utils.py:815-816)
    return nn.Parameter(t, requires_grad=False)


class _Embedder(nn.Module):
    def __init__(self, multires):
        super().__init__()
        fb = 2.0 ** torch.linspace(0.0, multires - 1, steps=multires)
        self.freq_bands = _buffer(fb[:, None, None].expand(multires, 2, 1).clone())
        self.multires = multires
        self.out_dim = 3 + 6 * multires


class _WNLinear(nn.Module):
    """parameters of nn.utils.weight_norm(nn.Linear): weight_g (O,1), weight_v (O,I), bias (O)."""

    def __init__(self, i, o):
        super().__init__()
        lin = nn.Linear(i, o)
        self.weight_v = nn.Parameter(lin.weight.detach().clone())
        self.weight_g = nn.Parameter(lin.weight.detach().norm(dim=1, keepdim=True))
        self.bias = nn.Parameter(lin.bias.detach().clone())


class _MLP(nn.Module):
    def __init__(self, input_ch, W, D, out_ch, skips=(4,)):
        super().__init__()
        self.linears = nn.ModuleList()
        for i in range(D + 1):
            I = input_ch if i == 0 else (input_ch + W if i in skips else W)
            O = out_ch if i == D else W
            self.linears.append(nn.Linear(I, O))


class ResidualDeformation(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embedder = _Embedder(cfg.xyz_res)
        self.mlp = _MLP(self.embedder.out_dim + cfg.cond_dim, 256, 8, 3)
        self.mlp.linears[-1].bias.data.zero_()


class _SdfMLP(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        dims = [d_in] + [256] * 8 + [d_out]
        for l in range(9):
            o = dims[l + 1] - dims[0] if l + 1 == 4 else dims[l + 1]
            setattr(self, f'lin{l}', _WNLinear(dims[l], o))


class SignedDistanceNetwork(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embedder = _Embedder(cfg.sdf_res)
        self._beta = nn.Parameter(torch.tensor(float(cfg.sdf_beta_init_value)))
        self.mlp = _SdfMLP(self.embedder.out_dim, 1 + cfg.feat_dim)

    @property
    def beta(self):
        return self._beta.clamp(1e-9, 1e6)


class RenderNetwork(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embedder = _Embedder(cfg.view_res)
        i0 = 3 + cfg.feat_dim + self.embedder.out_dim
        self.l0, self.l1, self.l2 = _WNLinear(i0, 256), _WNLinear(256, 256), _WNLinear(256, 256)
        self.l3, self.l4 = _WNLinear(256 + cfg.n_bones * 3, 256), _WNLinear(256, 3)


class Network(nn.Module):
    def __init__(self):
        super().__init__()
        cfg = config.active_cfg()
        self.cfg = cfg
        self.dist_th = cfg.dist_th
        self.residual_deformation_network = ResidualDeformation(cfg)
        self.signed_distance_network = SignedDistanceNetwork(cfg)
        self.render_network = RenderNetwork(cfg)
        self._engine = None
        self._dirty = True

    # ---- engine plumbing: parameters -> packed weights whenever they may have changed
    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._dirty = True
        return super().load_state_dict(*a, **k)

    def refresh(self):
        self._dirty = True

    def engine(self) -> Engine:
        if self._engine is None:
            dev = next(self.parameters()).device
            if dev.type != 'cuda':
                raise RuntimeError('relightableavatar_amd.Network must live on the GPU (call .cuda()); the render path has no CPU fallback')
            self._engine = Engine(self.cfg, dev, relight=hasattr(self, 'albedo_network'))
        if self._dirty:
            self._engine.load_state_dict(self.state_dict())
            self._dirty = False
        return self._engine

    def set_frame(self, batch):
        eng = self.engine()
        eng.set_frame(batch)
        return eng

    # ---- the reference's method surface
    @staticmethod
    def condition_vector(batch):
        return batch.poses.view(batch.poses.shape[0], -1)

    def inference_world_distance_field(self, x: torch.Tensor, batch, smooth_transition=False, **kwargs) -> torch.Tensor:
        eng = self.set_frame(batch)
        dist_th = kwargs.get('dist_th', self.dist_th)
        return eng.hdq_sdf(x, dist_th, smooth_transition).view(*x.shape[:-1], 1)

    @staticmethod
    def _template_frame(batch, identity_bones: bool):
        """a frame whose posed body IS the template (pverts = tverts, pnorm = tnorm, R = I, Th = 0): what geodesic_knn sees when
        the reference searches in space 't' (base_network.py:260-263) or filters big-pose points (:407-409)."""
        tb = dotdict(batch)
        tb.pverts, tb.pnorm = batch.tverts, batch.tnorm
        tb.R = torch.eye(3, device=batch.R.device, dtype=batch.R.dtype)[None]
        tb.Th = torch.zeros_like(batch.Th)
        if identity_bones:        # big-pose points are queried as they are: no inverse skinning
            eye = torch.eye(4, device=batch.A.device, dtype=batch.A.dtype).expand_as(batch.A).contiguous()
            tb.A, tb.big_A = eye, eye
        return tb

    def inference_observed_distance_field(self, x: torch.Tensor, batch, smooth_transition=False, filtering=False, **kwargs) -> torch.Tensor:
        """x: big-pose ("observed") points.  filtering=False: SDF(x + resd(x)).  filtering=True: the hierarchical query with the
        template as the body (the reference's ablation mode; HIP kernels are the same as for the world-space query)."""
        if not filtering:
            return self.set_frame(batch).observed_sdf(x).view(*x.shape[:-1], 1)
        tb = self._template_frame(batch, identity_bones=True)
        eng = self.engine()
        eng.set_frame(tb, force=True)
        try:
            return eng.hdq_sdf(x, kwargs.get('dist_th', self.dist_th), smooth_transition).view(*x.shape[:-1], 1)
        finally:
            eng.set_frame(batch, force=True)

    def world_to_bigpose_transform(self, x: torch.Tensor, batch, backward=False, **kwargs) -> torch.Tensor:
        """(B,P,4,4) = big_A_bw @ affine_inverse(A_bw) @ affine_inverse([R|Th]) per point, blended over the 3 nearest posed
        vertices of x (backward: template vertices of x).  Th may be (B,3) or the dataset's (B,1,3) — the reference only
        accepts the former (it concatenates Th[..., None] to R)."""
        return self._transform(x, batch, backward, invert=False)

    def bigpose_to_world_transform(self, x: torch.Tensor, batch, **kwargs) -> torch.Tensor:
        return self._transform(x, batch, True, invert=True)

    def _transform(self, x, batch, backward, invert):
        eng = self.engine()
        if backward:
            eng.set_frame(self._template_frame(batch, identity_bones=False), force=True)
        else:
            eng.set_frame(batch)
        try:
            out = eng.bigpose_transform(x, batch.R[0], batch.Th.reshape(-1, 3)[0], invert)
        finally:
            if backward:
                eng.set_frame(batch, force=True)
        return out.view(*x.shape[:-1], 4, 4)

    def forward(self, x: torch.Tensor, v: torch.Tensor, d, batch, **kwargs):
        eng = self.set_frame(batch)
        dist_th = kwargs.get('dist_th', None) or self.dist_th
        raw = eng.forward(x, v, dist_th)
        return dotdict(raw=raw.view(*x.shape[:-1], raw.shape[-1]))
