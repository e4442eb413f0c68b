import torch
import torch.nn as nn

from ..functional import tangent_lin


class TangentLin(nn.Module):
    """Bias-free complex linear layer on tangent-vector features (reference nn/tangent_lin.py:4-29):
    y[n,o] = sum_i x[n,i] (Re + i Im)[o,i], evaluated on MFMA by fc_tangent_lin_forward."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.Re = nn.Parameter(torch.empty(out_channels, in_channels))
        self.Im = nn.Parameter(torch.empty(out_channels, in_channels))
        torch.nn.init.xavier_uniform_(self.Re)
        torch.nn.init.xavier_uniform_(self.Im, gain=0.1)

    def forward(self, x):
        return tangent_lin(x, self.Re, self.Im)
