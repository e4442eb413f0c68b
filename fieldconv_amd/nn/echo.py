"""ECHO descriptors (reference nn/echo.py).  On a ROCm device the splat and its gradient are HIP kernels
(csrc/fc_echo.hip through fc_echo_forward / fc_echo_backward); the torch composite below is the host-side
restatement the CPU tests check against the reference fixtures.  It is written without `nonzero` compaction:
zero features are masked instead of filtered, which gives the same sums."""
import torch
import torch.nn as nn

from ..utils.field import isOrigin, softAbs, softAngle


def diskMap(n_bins):
    """Rasterised disk: flat (2n+1)^2 grid cell -> bin id; cells outside the disk alias bin 0
    (reference nn/echo.py:11-27)."""
    w = 2 * n_bins + 1
    ii, jj = torch.meshgrid(torch.arange(w), torch.arange(w), indexing='ij')
    inside = ((ii - n_bins) ** 2 + (jj - n_bins) ** 2).double() <= (n_bins + 0.25) ** 2
    flat = inside.reshape(-1)
    dmap = torch.zeros(w * w, dtype=torch.long)
    dmap[flat] = torch.arange(int(flat.sum()))
    return dmap, int(flat.sum())


def rasterize(p, dMap, n_bins):
    """Bilinear vote weights and bins of points p (complex, unit disk) (reference nn/echo.py:30-61).
    Returns rast (..., 4) float and ind (..., 4) long."""
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
    ind = torch.stack((dMap[w * f0 + f1], dMap[w * c0 + c1], dMap[w * c0 + f1], dMap[w * f0 + c1]), dim=-1)
    return rast, ind


class ECHO(nn.Module):
    """Per-channel ECHO descriptors of a tangent vector field (reference nn/echo.py:65-148)."""

    def __init__(self, channels, n_bins=2):
        super().__init__()
        self.channels = channels
        self.n_bins = n_bins
        dmap, dim = diskMap(n_bins)
        self.register_buffer('dMap', dmap)
        self.hdim = dim

    def forward(self, x, supp_edges, ln, wxp):
        if x.is_cuda:       # device tensors always take the HIP kernels (unsupported shapes raise, no torch fallback there)
            from ..functional import echo_descriptors
            return echo_descriptors(x, supp_edges, ln, wxp, self.n_bins)
        N, C, dS = x.shape[0], self.channels, self.hdim
        src, dst = supp_edges[:, 0], supp_edges[:, 1]
        live = torch.logical_not(isOrigin(x))                                   # (N,C)
        frame = torch.conj(torch.polar(torch.ones_like(x.real), softAngle(x)))  # exp(-i angle)
        aligned = ln[:, None] * frame[src]                                      # (E,C)
        rast, ind = rasterize(aligned, self.dMap, self.n_bins)                  # (E,C,4)
        xw = torch.where(live[src], x[src] * wxp[:, None], torch.zeros_like(x[src]))
        base = (dst[:, None] * C + torch.arange(C, device=x.device)[None, :]) * dS
        votes = (xw[..., None] * rast).reshape(-1)
        slots = (base[..., None] + ind).reshape(-1)
        hist = torch.zeros(N * C * dS, dtype=x.dtype, device=x.device).index_add(0, slots, votes)
        return softAbs(hist.reshape(N, C, dS))
