"""TransField / learned 'gradient' (reference nn/trans_field.py): the aggregation, the zonal contractions and their
gradients are HIP kernels (csrc/fc_trans_field.hip through fc_trans_field_forward / _backward).  Device tensors only."""
import torch
import torch.nn as nn


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
        from ..functional import trans_field       # raises for CPU tensors: there is no torch fallback
        return trans_field(x, supp_edges, lift_sten, self.zonalAng, self.zonalMag, self.phase, self.ftype)
