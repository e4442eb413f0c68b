"""CPU port of the reference FieldConv algorithm in stock torch ops -- TEST / BASELINE INFRASTRUCTURE.

Used only as (1) the `cpu_baseline` leg of bench.py and (2) a cross-check in tests/.  It keeps
the reference's *structure* (reference nn/field_conv.py:128-137 and :10-33): per-vertex phase
rotation with trigonometry, an edge-sized materialised product (E,C,R,F), an index-add to the
targets (what torch_scatter.scatter_add does), then a broadcast-multiply-and-sum
against the filter coefficients, with the backward pass left to torch autograd.  So its cost is
the reference CPU path's cost.  Pinned against the reference-generated fixtures in
tests/test_oracle_golden.py.  Never imported by fieldconv_amd.
"""
import torch

EPS = 1e-7


def soft_angle(z):
    box = (z.real.abs() < EPS) & (z.imag.abs() < EPS)
    return torch.where(box, torch.zeros_like(z.real), torch.angle(torch.where(box, torch.ones_like(z), z)))


def field_conv(x, supp_edges, supp_sten, zonal, spherical, phase, ftype, B, n_out=None):
    """n_out: number of output rows when the targets (supp_edges[:, 1]) are numbered 0..n_out-1 independently of the
    sources -- a slab of targets of a larger mesh (bench.py's cpu_baseline processes 20k-vertex meshes in slabs)."""
    N = x.shape[0] if n_out is None else n_out
    F = 2 * B + 1
    phi = soft_angle(x)
    m = torch.arange(-B, B + 1, device=x.device, dtype=phi.dtype)
    ang = -m[None, None, :] * phi[..., None]
    rotated = x[..., None] * torch.polar(torch.ones_like(ang), ang)                    # (N,C,F)
    per_edge = rotated[supp_edges[:, 0]][:, :, None, :] * supp_sten[:, None, :, :]    # (E,C,R,F) materialised
    contrib = torch.zeros((N,) + tuple(per_edge.shape[1:]), dtype=per_edge.dtype, device=x.device)
    contrib = contrib.index_add(0, supp_edges[:, 1], per_edge)
    sph = torch.view_as_complex(spherical)
    if ftype == 2:
        coeff = torch.cat((sph[..., :B], torch.view_as_complex(zonal)[..., None], sph[..., B:]), dim=3)
        return (contrib[:, None] * coeff[None]).sum(dim=(2, 3, 4)) / F
    coeff = torch.cat((torch.conj(sph).flip(3), zonal[..., None], sph), dim=3)
    if ftype == 0:
        return (contrib[:, None] * coeff[None]).sum(dim=(2, 3, 4)) / F
    ringsum = (contrib[:, None] * coeff[None]).sum(dim=3)                              # (N,O,I,F)
    ph = torch.cat((phase[:, :, 1:].flip(2), phase), dim=-1)
    return (ringsum * torch.polar(torch.ones_like(ph), ph)[None]).sum(dim=(2, 3)) / F
