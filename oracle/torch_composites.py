"""Torch restatements of the components around the hot path -- TEST INFRASTRUCTURE ONLY.

ECHO descriptors, TransField and FCPrecomp written with stock torch ops (device-agnostic, differentiable through torch
autograd, any real dtype).  The shipped package runs these components as HIP kernels and refuses CPU tensors; the
functions here are the checker.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.

Parity pinning: every function is checked against fixtures produced by importing the reference modules themselves
(``tests/golden/make_golden.py``): FCPrecomp in ``tests/test_host_logic.py::test_fc_precomp_matches_reference``, ECHO in
``::test_echo_descriptor_matches_reference``, TransField in ``::test_trans_field_matches_reference``.
Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import torch

EPS = 1e-7      # utils/field.py:8


# ---- zero-safe polar helpers, utils/field.py:10-48 (branch-free: masks instead of `nonzero` compaction) ----
def is_zero(x, eps=EPS):
    return (x < eps) & (x > -eps)


def is_origin(z, eps=EPS):
    return is_zero(z.real, eps) & is_zero(z.imag, eps)


def _safe(z, mask):
    return torch.where(mask, torch.ones_like(z), z)     # keeps autograd finite at masked entries


def soft_abs(z, eps=EPS):
    mask = is_origin(z, eps)
    return torch.where(mask, torch.zeros_like(z.real), torch.abs(_safe(z, mask)))


def soft_angle(z, eps=EPS):
    mask = is_origin(z, eps)
    return torch.where(mask, torch.zeros_like(z.real), torch.angle(_safe(z, mask)))


def soft_absolute(x):
    return torch.where(x < 0, -x, x)                    # utils/field.py:18-26, out of place


# ---- FCPrecomp, transforms/fc_precomp.py:10-27,53-97 ----
def radial_interpolant(r, n_rings):
    """(E,R) linear-interpolation weights on the equal-area knots sqrt(q/(R-1)); exactly two non-zeros per row
    (transforms/fc_precomp.py:10-27).  The upper knot is the first knot >= r, never knot 0."""
    knots = torch.sqrt(torch.arange(n_rings, device=r.device) / (n_rings - 1))
    gap = knots[None, :] - r[:, None]
    gap = torch.where(gap < 0, torch.full_like(gap, 1e8), gap)
    hi = torch.argmin(gap, dim=1).clamp_min(1)
    lo = hi - 1
    w_hi = (r - knots[lo]) / (knots[hi] - knots[lo])
    w = torch.zeros(r.shape[0], n_rings, device=r.device, dtype=torch.float32)
    w.scatter_(1, hi[:, None], w_hi[:, None].float())
    w.scatter_(1, lo[:, None], (1 - w_hi)[:, None].float())
    return w


def fc_precomp(r, theta, w, supp_edges, xp, band_limit, n_rings, epsilon):
    """(supp_edges', supp_sten (E',R,2B+1), ln (E'), wxp (E')) restricted to r <= epsilon (transforms/fc_precomp.py:53-97)."""
    B, R = band_limit, n_rings
    r = r / epsilon
    keep = torch.nonzero(r <= 1.0).squeeze(-1)
    r, theta, supp_edges, xp = r[keep], theta[keep], supp_edges[keep, :], xp[keep]
    ln = torch.polar(r, theta)
    ring = radial_interpolant(r, R)
    m = torch.arange(-B, B + 1, device=theta.device)
    ang = m[None, :] * theta[:, None]
    freq = torch.polar(torch.ones_like(ang), ang)
    src, dst = supp_edges[:, 0], supp_edges[:, 1]
    ws = w[src, 0]
    total = torch.zeros(w.shape[0], dtype=ws.dtype, device=ws.device).index_add(0, dst, ws)
    wxp = (ws / (1e-12 + total[dst])) * xp
    supp_sten = ring[:, :, None] * freq[:, None, :] * wxp[:, None, None]
    return supp_edges, supp_sten, ln, wxp


class FCPrecomp:
    """The reference transform's call contract (`FCPrecomp(band_limit, n_rings, epsilon)(data)`) on top of fc_precomp:
    how the tests build their stencils on the CPU."""

    def __init__(self, band_limit, n_rings, epsilon):
        self.B, self.R, self.max_r = band_limit, n_rings, epsilon

    def __call__(self, data):
        return fc_precomp(data.logMag, data.logAng, data.w, data.supp_edges, data.xp, self.B, self.R, self.max_r)


# ---- ECHO descriptors, nn/echo.py:11-27,30-61,94-148 ----
def disk_map(n_bins):
    """Rasterised disk: flat (2n+1)^2 grid cell -> bin id; cells outside the disk alias bin 0 (nn/echo.py:11-27)."""
    w = 2 * n_bins + 1
    ii, jj = torch.meshgrid(torch.arange(w), torch.arange(w), indexing='ij')
    inside = ((ii - n_bins) ** 2 + (jj - n_bins) ** 2).double() <= (n_bins + 0.25) ** 2
    flat = inside.reshape(-1)
    dmap = torch.zeros(w * w, dtype=torch.long)
    dmap[flat] = torch.arange(int(flat.sum()))
    return dmap, int(flat.sum())


def rasterize(p, d_map, n_bins):
    """Bilinear vote weights and bins of points p (complex, unit disk) (nn/echo.py:30-61): rast (...,4), ind (...,4)."""
    w = 2 * n_bins + 1
    q = torch.view_as_real(p * n_bins)
    qc = torch.clamp(torch.ceil(q), -n_bins, n_bins)
    qf = torch.clamp(torch.floor(q), -n_bins, n_bins)
    up = qc - q
    dn = q - qf
    rast = torch.stack((up[..., 0] * up[..., 1], dn[..., 0] * dn[..., 1],
                        dn[..., 0] * up[..., 1], up[..., 0] * dn[..., 1]), dim=-1)
    c0, c1 = qc[..., 0].long() + n_bins, qc[..., 1].long() + n_bins
    f0, f1 = qf[..., 0].long() + n_bins, qf[..., 1].long() + n_bins
    ind = torch.stack((d_map[w * f0 + f1], d_map[w * c0 + c1], d_map[w * c0 + f1], d_map[w * f0 + c1]), dim=-1)
    return rast, ind


def echo_descriptors(x, supp_edges, ln, wxp, n_bins):
    """|hist| (N, C, dS): per-channel ECHO descriptors of a tangent vector field (nn/echo.py:94-148).  Written without
    `nonzero` compaction: zero features are masked instead of filtered, which gives the same sums."""
    d_map, dS = disk_map(n_bins)
    d_map = d_map.to(x.device)
    N, C = x.shape
    src, dst = supp_edges[:, 0], supp_edges[:, 1]
    live = torch.logical_not(is_origin(x))                                   # (N,C)
    frame = torch.conj(torch.polar(torch.ones_like(x.real), soft_angle(x)))  # exp(-i angle)
    aligned = ln[:, None] * frame[src]                                       # (E,C)
    rast, ind = rasterize(aligned, d_map, n_bins)                            # (E,C,4)
    xw = torch.where(live[src], x[src] * wxp[:, None], torch.zeros_like(x[src]))
    base = (dst[:, None] * C + torch.arange(C, device=x.device)[None, :]) * dS
    votes = (xw[..., None] * rast).reshape(-1)
    slots = (base[..., None] + ind).reshape(-1)
    hist = torch.zeros(N * C * dS, dtype=x.dtype, device=x.device).index_add(0, slots, votes)
    return soft_abs(hist.reshape(N, C, dS))


# ---- TransField, nn/trans_field.py:9-24,78-113 ----
def trans_field(x, supp_edges, lift_sten, zonal_ang, zonal_mag, phase, ftype):
    """x (N,in) real; lift_sten (E,R,>=2) complex (columns 0 and 1 are used) -> (N,out) complex (nn/trans_field.py:78-113)."""
    N, R = x.shape[0], zonal_ang.shape[2]
    src, dst = supp_edges[:, 0], supp_edges[:, 1]
    s0 = lift_sten[:, :, 0]
    s1 = lift_sten[:, :, 1]
    diff = x[src] - x[dst]                                                     # (E,in)
    ang = torch.zeros((N, x.shape[1], R), dtype=lift_sten.dtype, device=x.device)
    ang = -ang.index_add(0, dst, diff[..., None] * s1[:, None, :])             # trans_field.py:106
    mag = torch.zeros((N, x.shape[1], R), dtype=x.dtype, device=x.device)
    mag = mag.index_add(0, dst, x[src][..., None] * soft_abs(s0)[:, None, :])  # trans_field.py:110
    phi = soft_angle(torch.einsum('nir,oir->noi', ang, zonal_ang.to(ang.dtype)))
    if ftype != 0:
        phi = phi + phase[None]
    rho = soft_absolute(torch.einsum('nir,oir->noi', mag, zonal_mag))
    return torch.polar(rho, phi).sum(dim=-1)


# ---- pointwise tangent ops, nn/tangent_lin.py:27-29 and nn/tangent_nonlin.py:24-35 ----
def tangent_lin(x, re_w, im_w):
    """y[n,o] = sum_i x[n,i] (Re + i Im)[o,i]"""
    return x @ torch.complex(re_w, im_w).to(x.dtype).t()


def tangent_nonlin(x, bias):
    """modReLU: non-origin x -> relu(|x| + b_c) x / |x|; entries inside the origin box pass through unchanged."""
    mask = is_origin(x)
    mag = torch.abs(_safe(x, mask))
    out = torch.relu(mag + bias.to(mag.dtype)) / mag * _safe(x, mask)
    return torch.where(mask, x, out)
