"""TransField / learned 'gradient' (reference nn/trans_field.py).  On a ROCm device the aggregation, the zonal
contractions and their gradients are HIP kernels (csrc/fc_trans_field.hip through fc_trans_field_forward / _backward);
the torch composite below is the host-side restatement the CPU tests check against the reference fixtures."""
import torch
import torch.nn as nn

from ..utils.field import softAbs, softAbsolute, softAngle


class TransField(nn.Module):
    """Scalar features -> equivariant tangent-vector features (supplement section C, eqs (2)-(3))."""

    def __init__(self, in_channels, out_channels, n_rings=6, ftype=1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.R = n_rings
        self.ftype = ftype
        self.zonalAng = nn.Parameter(torch.empty(out_channels, in_channels, n_rings))
        self.zonalMag = nn.Parameter(torch.empty(out_channels, in_channels, n_rings))
        if ftype == 0:
            self.register_buffer('phase', torch.zeros(out_channels, in_channels))
        else:
            self.phase = nn.Parameter(torch.empty(out_channels, in_channels))
            torch.nn.init.xavier_uniform_(self.phase)
        torch.nn.init.xavier_uniform_(self.zonalAng)
        torch.nn.init.xavier_uniform_(self.zonalMag)

    def forward(self, x, supp_edges, lift_sten):
        """x (N,in) real; lift_sten (E,R,2) cfloat = stencil columns m=0,1 -> (N,out) cfloat."""
        if x.is_cuda:       # device tensors always take the HIP kernels (unsupported shapes raise, no torch fallback there)
            from ..functional import trans_field
            return trans_field(x, supp_edges, lift_sten, self.zonalAng, self.zonalMag, self.phase, self.ftype)
        N = x.shape[0]
        src, dst = supp_edges[:, 0], supp_edges[:, 1]
        s0 = lift_sten[:, :, 0]
        s1 = lift_sten[:, :, 1]
        diff = x[src] - x[dst]                                                     # (E,in)
        ang = torch.zeros((N, x.shape[1], self.R), dtype=lift_sten.dtype, device=x.device)
        ang = -ang.index_add(0, dst, diff[..., None] * s1[:, None, :])             # trans_field.py:106
        mag = torch.zeros((N, x.shape[1], self.R), dtype=x.dtype, device=x.device)
        mag = mag.index_add(0, dst, x[src][..., None] * softAbs(s0)[:, None, :])   # trans_field.py:110
        phi = softAngle(torch.einsum('nir,oir->noi', ang, self.zonalAng.to(ang.dtype)))
        if self.ftype != 0:
            phi = phi + self.phase[None]
        rho = softAbsolute(torch.einsum('nir,oir->noi', mag, self.zonalMag))
        return torch.polar(rho, phi).sum(dim=-1)
