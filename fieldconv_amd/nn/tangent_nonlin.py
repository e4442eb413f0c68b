import torch
import torch.nn as nn

from ..functional import tangent_nonlin


class TangentNonLin(nn.Module):
    """modReLU on the magnitude of complex features, Equation (8) of the paper (reference
    nn/tangent_nonlin.py:8-35).  Entries inside the origin box pass through unchanged."""

    def __init__(self, in_channels):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, in_channels))

    def forward(self, x):
        return tangent_nonlin(x, self.bias)
