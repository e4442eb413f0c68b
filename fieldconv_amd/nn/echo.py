"""ECHO descriptors (reference nn/echo.py): the splat and its gradient are HIP kernels (csrc/fc_echo.hip through
fc_echo_forward / fc_echo_backward).  Device tensors only, like every operator of this package."""
import torch
import torch.nn as nn


def diskMap(n_bins):
    """Rasterised disk: flat (2n+1)^2 grid cell -> bin id; cells outside the disk alias bin 0
    (reference nn/echo.py:11-27).  Only the `dMap` buffer of the module (state_dict layout); the kernels rebuild the
    same table on the device."""
    w = 2 * n_bins + 1
    ii, jj = torch.meshgrid(torch.arange(w), torch.arange(w), indexing='ij')
    inside = ((ii - n_bins) ** 2 + (jj - n_bins) ** 2).double() <= (n_bins + 0.25) ** 2
    flat = inside.reshape(-1)
    dmap = torch.zeros(w * w, dtype=torch.long)
    dmap[flat] = torch.arange(int(flat.sum()))
    return dmap, int(flat.sum())


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
        from ..functional import echo_descriptors      # raises for CPU tensors: there is no torch fallback
        return echo_descriptors(x, supp_edges, ln, wxp, self.n_bins)
