"""FieldConv: same constructor, parameters, state_dict and forward signature as the reference
module (reference nn/field_conv.py:36-137), with the gather-rotate-filter-reduce loop running in
the HIP kernels of libfieldconv_hip.so instead of torch broadcast ops + torch_scatter."""
import torch
import torch.nn as nn
from torch.nn import Parameter

from ..functional import field_conv_act, field_conv_params
from ..graph import get_graph


def effective_filter(zonal, spherical, phase, ftype, B):
    """W_eff (O,I,R,2B+1) complex64 with y = <contrib, W_eff> / (2B+1).

    ftype 0 / 1: Hermitian-symmetric coefficients [conj(sph) reversed | zonal | sph] (reference
    nn/field_conv.py:12,18); ftype 1 additionally carries exp(i*phase[o,i,|m|]) (:23-25), which
    commutes with the ring sum and is therefore folded into the filter here.  ftype 2: the complex
    coefficients as stored (:31).  Differentiable: torch autograd carries the (tiny) chain from the
    kernels' dL/dW_eff back to (zonal, spherical, phase).
    """
    sph = torch.view_as_complex(spherical)
    if ftype == 2:
        zc = torch.view_as_complex(zonal)
        return torch.cat((sph[..., :B], zc[..., None], sph[..., B:]), dim=3)
    coeff = torch.cat((torch.conj(sph).flip(3), zonal[..., None].to(sph.dtype), sph), dim=3)
    if ftype == 1:
        ph = torch.cat((phase[:, :, 1:].flip(2), phase), dim=-1)
        coeff = coeff * torch.polar(torch.ones_like(ph), ph)[:, :, None, :]
    return coeff


def _contract(contrib, w_eff, B):
    N, O = contrib.shape[0], w_eff.shape[0]
    return (contrib.reshape(N, -1) @ w_eff.reshape(O, -1).transpose(0, 1)) / (2 * B + 1)


def weightContribReal(contrib, zonal, spherical, phase, B):
    """(N,I,R,F) response -> (N,O); kept for API compatibility (reference nn/field_conv.py:10-14)."""
    return _contract(contrib, effective_filter(zonal, spherical, phase, 0, B), B)


def weightContribOffset(contrib, zonal, spherical, phase, B):
    """reference nn/field_conv.py:16-25"""
    return _contract(contrib, effective_filter(zonal, spherical, phase, 1, B), B)


def weightContribComplex(contrib, zonal, spherical, phase, B):
    """reference nn/field_conv.py:28-33"""
    return _contract(contrib, effective_filter(zonal, spherical, phase, 2, B), B)


class FieldConv(nn.Module):
    """Field convolution (Equations (4) and (7) of the paper).

    in_channels, out_channels: complex tangent-vector feature channels
    band_limit: angular band limit B (2B+1 frequencies);  n_rings: radial bins R
    ftype: 0 real filters, 1 real filters + per-channel phase offsets, 2 complex filters
    """

    def __init__(self, in_channels, out_channels, band_limit=1, n_rings=6, ftype=1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.R = n_rings
        self.B = band_limit
        self.ftype = ftype
        O, I, R, B = out_channels, in_channels, n_rings, band_limit
        if ftype == 0:
            self.zonal = Parameter(torch.empty(O, I, R))
            self.spherical = Parameter(torch.empty(O, I, R, B, 2))
            self.register_buffer('phase', torch.zeros(O, I, B + 1))
            self.WR = weightContribReal
        elif ftype == 1:
            self.zonal = Parameter(torch.empty(O, I, R))
            self.spherical = Parameter(torch.empty(O, I, R, B, 2))
            self.phase = Parameter(torch.empty(O, I, B + 1))
            torch.nn.init.xavier_uniform_(self.phase)
            self.WR = weightContribOffset
        else:
            self.zonal = Parameter(torch.empty(O, I, R, 2))
            self.spherical = Parameter(torch.empty(O, I, R, 2 * B, 2))
            self.register_buffer('phase', torch.zeros(O, I, B + 1))
            self.WR = weightContribComplex
        torch.nn.init.xavier_uniform_(self.zonal)
        torch.nn.init.xavier_uniform_(self.spherical)

    def effective_filter(self):
        return effective_filter(self.zonal, self.spherical, self.phase, self.ftype, self.B)

    def forward(self, x, supp_edges, supp_sten):
        """x (N,in) cfloat; supp_edges (E,2) long, (j, i): j -> i; supp_sten (E,R,2B+1) cfloat -> (N,out) cfloat."""
        if supp_sten.dim() != 3 or supp_sten.shape[1] != self.R or supp_sten.shape[2] != 2 * self.B + 1:
            raise ValueError(f'supp_sten must be (E, {self.R}, {2 * self.B + 1}), got {tuple(supp_sten.shape)}')
        graph = get_graph(supp_edges, supp_sten, x.shape[0])
        return field_conv_params(x, self.zonal, self.spherical, self.phase, self.ftype, self.B, graph)

    def forward_act(self, x, supp_edges, supp_sten, bias, addend=None):
        """modReLU(self(x) + addend) with the residual add and the modReLU in the convolution kernel's epilogue: what
        FCResNetBlock / ECHOBlock compute around their convolutions (reference nn/fc_resnet_block.py:84-88)."""
        if supp_sten.dim() != 3 or supp_sten.shape[1] != self.R or supp_sten.shape[2] != 2 * self.B + 1:
            raise ValueError(f'supp_sten must be (E, {self.R}, {2 * self.B + 1}), got {tuple(supp_sten.shape)}')
        graph = get_graph(supp_edges, supp_sten, x.shape[0])
        return field_conv_act(x, self.zonal, self.spherical, self.phase, self.ftype, self.B, graph, bias, addend)
