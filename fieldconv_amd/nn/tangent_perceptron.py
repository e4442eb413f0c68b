import torch.nn as nn

from .tangent_lin import TangentLin
from .tangent_nonlin import TangentNonLin


class TangentPerceptron(nn.Module):
    """TangentLin followed by TangentNonLin (reference nn/tangent_perceptron.py:7-23)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = TangentLin(in_channels, out_channels)
        self.nonlin = TangentNonLin(out_channels)

    def forward(self, x):
        return self.nonlin(self.lin(x))
