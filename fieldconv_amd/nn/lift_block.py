import torch.nn as nn

from .tangent_nonlin import TangentNonLin
from .trans_field import TransField


class LiftBlock(nn.Module):
    """TransField followed by modReLU (reference nn/lift_block.py:6-55)."""

    def __init__(self, in_channels, out_channels, n_rings=6, ftype=1):
        super().__init__()
        self.field = TransField(in_channels, out_channels, n_rings=n_rings, ftype=ftype)
        self.nonlin = TangentNonLin(out_channels)

    def forward(self, x, supp_edges, lift_sten):
        from ..blocks import lift_block            # one autograd node, one native call per pass (csrc/fc_blocks.hip)
        out = lift_block(self, x, supp_edges, lift_sten)
        if out is not None:
            return out
        return self.nonlin(self.field(x, supp_edges, lift_sten))
